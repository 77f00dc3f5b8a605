"""Pipeline parallelism (`parallel.pipeline = dict(size=pp)`): the non-interleaved 1F1B schedule.

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
The interleaved schedule (model.num_chunks > 1) is not implemented.
"""
import torch
import torch.distributed as dist


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
        self.dp_group = self.group = None
        self.backend = None
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
            self.backend = dist.get_backend(self.group)

    # ---- point to point: one call = the sends and receives that may proceed together ----------------------------------------
    def exchange(self, sends=(), recvs=()):
        """sends / recvs: [(device tensor, global peer rank)].  Returns after all of them completed (the receive buffers are valid,
        the send buffers reusable).  RCCL: device buffers directly; gloo (tests): staged through the host."""
        if not sends and not recvs:
            return
        if self.backend == "nccl":
            ops = [dist.P2POp(dist.isend, t, peer) for t, peer in sends] + [dist.P2POp(dist.irecv, t, peer) for t, peer in recvs]
            for w in dist.batch_isend_irecv(ops):
                w.wait()
            return
        host_s = [(t.detach().to("cpu", copy=True).contiguous(), peer) for t, peer in sends]
        host_r = [(torch.empty(t.shape, dtype=t.dtype), peer) for t, peer in recvs]
        work = [dist.isend(c, peer) for c, peer in host_s] + [dist.irecv(c, peer) for c, peer in host_r]
        for w in work:
            w.wait()
        for (t, _), (c, _) in zip(recvs, host_r):
            t.copy_(c)

    def all_reduce_sum(self, t):
        """In-place sum over the stages of this pipeline (squared gradient norm)."""
        if self.pp == 1:
            return t
        if self.backend == "nccl":
            dist.all_reduce(t, op=dist.ReduceOp.SUM, group=self.group)
        else:
            c = t.detach().to("cpu", copy=True)
            dist.all_reduce(c, op=dist.ReduceOp.SUM, group=self.group)
            t.copy_(c)
        return t

    def broadcast_from_last(self, t):
        """The last stage's value of `t` (the loss) on every stage of the pipeline."""
        if self.pp == 1:
            return t
        src = (self.pp - 1) * self.dp_world + self.dp_rank
        if self.backend == "nccl":
            dist.broadcast(t, src=src, group=self.group)
        else:
            c = t.detach().to("cpu", copy=True)
            dist.broadcast(c, src=src, group=self.group)
            t.copy_(c)
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
