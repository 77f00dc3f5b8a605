"""Pipeline parallelism (`parallel.pipeline = dict(size=pp)`): the 1F1B schedule, non-interleaved and interleaved.

Reference behaviour being matched (core/scheduler/pipeline_scheduler.py:111-709 `PipelineScheduler`, core/communication/p2p.py,
solver/pipeline_utils.py:9-34 `partition_uniform`, core/context/process_group_initializer.py `Initializer_Pipeline`):
  * the L transformer layers are cut into pp contiguous ranges (L // pp each, the LAST L % pp stages one more); the first stage also
    owns the embedding, the last one the final norm, the head and the loss;
  * ranks of one stage are consecutive: stage = rank // (world / pp); the ranks with the same position in their stage form a
    pipeline (rank, rank + world / pp, ...) and read the same micro-batches; data parallelism + ZeRO run inside a stage;
  * per optimizer step a stage runs min(pp - stage - 1, M) warm-up forwards, then one-forward-one-backward, then the remaining
    backwards; activations [T, hidden] travel forward, their gradients backward, between neighbouring stages only;
  * gradients accumulate over the M micro-batches exactly as without pipeline (loss / M each); the gradient norm (and with it the
    overflow decision of the loss scaler) is summed over the stages of a pipeline.
MI355X notes: one process per GPU, point-to-point over xGMI (every GPU pair has its own link); the paired send + receive of the
steady state is ONE batch_isend_irecv so both directions of a link are busy at once.  A stage keeps the saved activations of its
in-flight micro-batches (at most pp - stage) as whole sets in HBM -- 288 GB make activation recomputation unnecessary here.

Interleaved 1F1B (`model.num_chunks = C > 1`, InterleavedPipelineScheduler, pipeline_scheduler.py:711-1430): every stage holds C model chunks;
chunk c of stage s is virtual stage c * pp + s, its layers the s-th part of the c-th C-th of the model (pipeline_utils.py:9-34).  Micro-batches
travel stage 0 -> pp - 1 through chunk 0, wrap around to stage 0 for chunk 1, and so on: the neighbours form a RING.  A stage works through
micro_num * C forward and as many backward micro-steps in the reference's order -- (pp - stage - 1) * 2 + (C - 1) * pp warm-up forwards (all
of them when micro_num == pp), then one-forward-one-backward, then the remaining backwards; forward micro-step k is chunk (k mod pp C) div pp of
micro-batch (k div pp C) pp + k mod pp, backward micro-step k the mirrored chunk (:925-945, :1327-1373) -- micro_num must be a multiple of pp.
What this module adds to that order is WHEN messages move: `interleaved_plan` replays the whole pipeline on a common clock (every stage runs
its next micro-step in the first tick in which its input has arrived) and gives every stage, per tick, its micro-step and the messages it sends
and receives behind it.  All stages then run the same number of ticks with ONE paired exchange per tick, a send and its receive always in the
same tick's exchange: no ordering of individual sends and receives between two ranks can deadlock (RCCL point-to-point operations of a pair
share a stream), and a message that arrives before its consumer runs waits in that micro-batch's own buffer.
"""
import torch.distributed as dist

from .comm import backend_for


def partition_uniform(num_layers, pp):
    """[(lo, hi)] per stage: solver/pipeline_utils.py:9-34 with num_chunks = 1."""
    base, left = num_layers // pp, pp - num_layers % pp
    if base == 0:
        raise ValueError("Some nodes in Pipeline have no requests")
    out, lo = [], 0
    for p in range(pp):
        hi = lo + base + (1 if p >= left else 0)
        out.append((lo, hi))
        lo = hi
    assert lo == num_layers
    return out


def partition_chunks(num_layers, pp, chunks):
    """[[(lo, hi) per chunk] per stage]: solver/pipeline_utils.py:9-34."""
    if num_layers % chunks:
        raise ValueError("Layer length should be divided by the number of chunks, otherwise parameter method is recomended")
    per = num_layers // chunks
    parts = [[] for _ in range(pp)]
    for c in range(chunks):
        for p, (lo, hi) in enumerate(partition_uniform(per, pp)):
            parts[p].append((c * per + lo, c * per + hi))
    return parts


def interleaved_order(stage, pp, chunks, micro_num):
    """The micro-steps of one stage in the reference's order: [("F" | "B", micro-batch, chunk)] (pipeline_scheduler.py:925-945, 1327-1373)."""
    if micro_num % pp:
        raise ValueError(f"num_microbatches: {micro_num} must be an integer multiple of pipeline parallel world size")
    n = micro_num * chunks
    warm = n if micro_num == pp else min((pp - stage - 1) * 2 + (chunks - 1) * pp, n)

    def step(k, backward):
        g, r = divmod(k, pp * chunks)
        c = r // pp
        return ("B" if backward else "F", g * pp + r % pp, chunks - 1 - c if backward else c)

    order = [step(k, False) for k in range(warm)]
    for i in range(n - warm):
        order += [step(warm + i, False), step(i, True)]
    order += [step(i, True) for i in range(n - warm, n)]
    return order


def interleaved_plan(pp, chunks, micro_num):
    """plan[stage] = [tick] with tick = dict(op = ("F" | "B", m, c) or None, sends = [(kind, m, c, to_stage)], recvs = [(kind, m, c, from_stage)]):
    kind "F" = the input of forward micro-step (m, c) of the RECEIVING stage, "B" = the output gradient of its backward micro-step (m, c).  A
    message is sent in the exchange behind the tick that produced it and may be consumed from the next tick on."""
    orders = [interleaved_order(s, pp, chunks, micro_num) for s in range(pp)]
    pos = [0] * pp
    have = [set() for _ in range(pp)]          # messages that have arrived: (kind, m, c)
    plan = [[] for _ in range(pp)]
    last_v = pp * chunks - 1
    while any(pos[s] < len(orders[s]) for s in range(pp)):
        ran = []
        for s in range(pp):
            op = None
            if pos[s] < len(orders[s]):
                kind, m, c = orders[s][pos[s]]
                v = c * pp + s
                needs_msg = (kind == "F" and v > 0) or (kind == "B" and v < last_v)
                if not needs_msg or (kind, m, c) in have[s]:
                    op = orders[s][pos[s]]
            ran.append(op)
        if not any(ran):
            raise RuntimeError("interleaved pipeline order cannot make progress")   # (cannot happen with the reference's order)
        arrivals = [[] for _ in range(pp)]
        for s, op in enumerate(ran):
            tick = dict(op=op, sends=[], recvs=[])
            if op is not None:
                pos[s] += 1
                kind, m, c = op
                v = c * pp + s
                if kind == "F" and v < last_v:     # output -> next virtual stage
                    to, cc = (s + 1) % pp, c + (s == pp - 1)
                    tick["sends"].append(("F", m, cc, to))
                    arrivals[to].append(("F", m, cc, s))
                if kind == "B" and v > 0:          # input gradient -> previous virtual stage
                    to, cc = (s - 1) % pp, c - (s == 0)
                    tick["sends"].append(("B", m, cc, to))
                    arrivals[to].append(("B", m, cc, s))
            plan[s].append(tick)
        for s in range(pp):
            for kind, m, c, frm in arrivals[s]:
                plan[s][-1]["recvs"].append((kind, m, c, frm))
                have[s].add((kind, m, c))
    return plan


class PipeParallel:
    def __init__(self, pp_size, rank, world_size):
        if world_size % pp_size:
            raise ValueError(f"world size {world_size} is not a multiple of the pipeline size {pp_size}")
        self.pp = pp_size
        self.dp_world = world_size // pp_size
        self.stage = rank // self.dp_world
        self.dp_rank = rank % self.dp_world
        self.first, self.last = self.stage == 0, self.stage == pp_size - 1
        self.prev = rank - self.dp_world if not self.first else None      # global ranks of the neighbouring stages of this pipeline
        self.next = rank + self.dp_world if not self.last else None
        self.stage_rank = [s * self.dp_world + self.dp_rank for s in range(pp_size)]   # global rank of every stage of this pipeline (the ring of the interleaved schedule)
        self.dp_group = self.group = None
        self.backend = None
        self.be = None
        if pp_size > 1:
            if not dist.is_initialized():
                raise RuntimeError("torch.distributed must be initialised for pipeline parallelism")
            for s in range(pp_size):           # collective: every rank creates every group in the same order
                ranks = list(range(s * self.dp_world, (s + 1) * self.dp_world))
                grp = dist.new_group(ranks)
                if rank in ranks:
                    self.dp_group = grp
            for d in range(self.dp_world):
                ranks = list(range(d, world_size, self.dp_world))
                grp = dist.new_group(ranks)
                if rank in ranks:
                    self.group = grp
            self.be = backend_for(self.group)
            self.backend = self.be.name

    # ---- point to point: one call = the sends and receives that may proceed together ----------------------------------------
    def exchange(self, sends=(), recvs=()):
        """sends / recvs: [(device tensor, global peer rank)].  Returns after all of them are ordered in front of the current stream
        (the receive buffers are valid, the send buffers reusable): ONE batch_isend_irecv on RCCL."""
        if not sends and not recvs:
            return
        self.be.exchange(list(sends), list(recvs)).wait()

    def all_reduce_sum(self, t):
        """In-place sum over the stages of this pipeline (squared gradient norm)."""
        if self.pp > 1:
            self.be.all_reduce(t, self.group).wait()
        return t

    def broadcast_from_last(self, t):
        """The last stage's value of `t` (the loss) on every stage of the pipeline."""
        if self.pp > 1:
            self.be.broadcast(t, (self.pp - 1) * self.dp_world + self.dp_rank, self.group).wait()
        return t

    def barrier(self):
        if self.pp > 1:
            dist.barrier(group=self.group)


def schedule_1f1b(stage, pp, micro_num):
    """The order of work of one stage: [("F", i) | ("B", i)] (pipeline_scheduler.py:430-560: warm-up, 1F1B, cool-down)."""
    warm = min(pp - stage - 1, micro_num)
    order = [("F", i) for i in range(warm)]
    for i in range(micro_num - warm):
        order += [("F", warm + i), ("B", i)]
    order += [("B", i) for i in range(micro_num - warm, micro_num)]
    return order
