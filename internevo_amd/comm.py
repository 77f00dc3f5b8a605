"""The collective / point-to-point primitives every parallel mode of the engine is written against.

ONE code shape for the product and for the tests.  zero.py, tensorpar.py, seqpar.py, pipeline.py, moe.py and metrics.py never ask which
backend is running: they call a `Backend` object from here and get `Work` handles back, and they place `wait()` where the RCCL path needs
it.  Two implementations of the same contract:

  * `RcclBackend` ("nccl" = RCCL over xGMI, the product): the c10d call on the device tensors, always `async_op=True`; the collective
    runs on c10d's own HIP stream behind the work already queued on the caller's current stream, `Work.wait()` orders the caller's
    current stream behind it.  In-place forms are used wherever a destination aliases its source (reduce-scatter into the owner's
    slice of the bucket, all-gather from the owner's slice into the bucket).

  * `StagedGlooBackend` ("gloo", the multi-process tests on a box with ONE GPU -- RCCL refuses two ranks on one device -- and on CPU):
    reads the source at the call (a device -> host copy, ordered behind the current stream exactly like c10d's stream wait), runs the
    collective over gloo on the host, and writes the destination ONLY in `wait()`, on the stream that is current there.  A missing
    or misplaced `wait()` therefore shows up as stale data in a test instead of passing by accident, destinations may alias their
    sources as on RCCL, and the order of calls per group is the order the product issues them in.

Reference calls being replaced: dist.all_reduce / reduce_scatter / all_gather / broadcast / all_to_all_single / batch_isend_irecv at
hybrid_zero_optim.py:489-527,809-837, solver/optimizer/utils.py:80-134,352-357, model/utils.py:228-346, multi_head_attention.py:28-53,
core/communication/p2p.py, moe/sharded_moe.py (_AllToAll).
"""
import torch
import torch.distributed as dist


class WaitTimer:
    """Exposed communication time (bench.py): while `comm.TIMER` holds one of these, every Work.wait() is bracketed by two HIP events on the caller's
    current stream -- the first completes when the stream reaches the wait, the second when the stream may continue behind the collective -- so their
    distance is the time the COMPUTE stream stood still for that collective (0 when the collective had finished under earlier kernels).  No host
    synchronisation; read with `summary()` after the timed region."""

    def __init__(self):
        self.pairs = []

    def summary(self):
        """{kind: (milliseconds the waiting stream was held, number of waits)} -- synchronises."""
        out = {}
        for kind, e0, e1 in self.pairs:
            e1.synchronize()
            ms, n = out.get(kind, (0.0, 0))
            out[kind] = (ms + e0.elapsed_time(e1), n + 1)
        return out


TIMER = None   # a WaitTimer while bench.py times its steps


class Work:
    """A collective in flight.  wait(): order the caller's current stream behind it (and, staged backend, land the result)."""

    def __init__(self, handle=None, land=None, kind=None):
        self._handle, self._land, self.kind = handle, land, kind

    def wait(self):
        timer = TIMER if (self._handle is not None or self._land is not None) and torch.cuda.is_available() else None
        if timer is not None:
            e0 = torch.cuda.Event(enable_timing=True)
            e0.record()
        if self._handle is not None:
            self._handle.wait()
            self._handle = None
        if self._land is not None:
            land, self._land = self._land, None
            land()
        if timer is not None:
            e1 = torch.cuda.Event(enable_timing=True)
            e1.record()
            timer.pairs.append((self.kind or "other", e0, e1))
        return True


DONE = Work()


class RcclBackend:
    name = "nccl"

    def reduce_scatter(self, shard, full, group, avg=True):
        """full -> this rank's 1/world part, reduced over the group; `shard` may be (and in the engine is) the rank's own slice of `full`."""
        return Work(dist.reduce_scatter_tensor(shard, full, op=dist.ReduceOp.AVG if avg else dist.ReduceOp.SUM, group=group, async_op=True), kind="reduce_scatter")

    def all_reduce(self, t, group, avg=False):
        return Work(dist.all_reduce(t, op=dist.ReduceOp.AVG if avg else dist.ReduceOp.SUM, group=group, async_op=True), kind="all_reduce")

    def all_gather(self, full, shard, group):
        """every rank's `shard` -> `full` (rank order); `shard` may be the rank's own slice of `full`."""
        return Work(dist.all_gather_into_tensor(full, shard, group=group, async_op=True), kind="all_gather")

    def all_to_all(self, recv, send, group):
        """chunk r of the flat `send` -> rank r; chunk s of the flat `recv` <- rank s."""
        return Work(dist.all_to_all_single(recv.view(-1), send.view(-1), group=group, async_op=True), kind="all_to_all")

    def broadcast(self, t, src, group):
        return Work(dist.broadcast(t, src=src, group=group, async_op=True), kind="broadcast")

    def exchange(self, sends, recvs):
        """[(tensor, global peer rank)] each: the sends and receives that may proceed together, as ONE batch (RCCL runs the point-to-point
        operations of a pair of ranks on one stream: a send and the matching receive must be in the same batch on both sides)."""
        ops = [dist.P2POp(dist.isend, t, peer) for t, peer in sends] + [dist.P2POp(dist.irecv, t, peer) for t, peer in recvs]
        if not ops:
            return DONE
        works = dist.batch_isend_irecv(ops)

        class _All:
            def wait(self):
                for w in works:
                    w.wait()

        return Work(_All(), kind="send_recv")


def _host(t):
    return t.detach().to("cpu", copy=True).contiguous()


class StagedGlooBackend:
    name = "gloo"

    def reduce_scatter(self, shard, full, group, avg=True):
        src = _host(full)
        n = shard.numel()
        world = dist.get_world_size(group)
        # gloo has neither reduce_scatter_tensor for every dtype nor AVG: all-reduce(SUM) in fp32, cut, scale -- the value RCCL's
        # AVG produces up to the summation order (tests compare against a one-rank run with tolerances that cover it)
        acc = src.float()
        dist.all_reduce(acc, op=dist.ReduceOp.SUM, group=group)
        r = dist.get_rank(group)
        part = acc.view(-1)[r * n : (r + 1) * n]
        if avg:
            part = part / world
        return Work(land=lambda: shard.view(-1).copy_(part.to(shard.dtype)), kind="reduce_scatter")

    def all_reduce(self, t, group, avg=False):
        c = _host(t)
        wide = c.float() if c.dtype in (torch.bfloat16, torch.float16) else c
        dist.all_reduce(wide, op=dist.ReduceOp.SUM, group=group)
        if avg:
            wide = wide / dist.get_world_size(group)
        return Work(land=lambda: t.copy_(wide.to(t.dtype)), kind="all_reduce")

    def all_gather(self, full, shard, group):
        c = _host(shard)
        parts = [torch.empty_like(c) for _ in range(dist.get_world_size(group))]
        dist.all_gather(parts, c, group=group)
        return Work(land=lambda: full.view(-1).copy_(torch.cat([p.view(-1) for p in parts])), kind="all_gather")

    def all_to_all(self, recv, send, group):
        s = _host(send).view(-1)
        r = torch.empty_like(s)
        dist.all_to_all_single(r, s, group=group)
        return Work(land=lambda: recv.view(-1).copy_(r), kind="all_to_all")

    def broadcast(self, t, src, group):
        c = _host(t)
        dist.broadcast(c, src=src, group=group)
        return Work(land=lambda: t.copy_(c), kind="broadcast")

    def exchange(self, sends, recvs):
        if not sends and not recvs:
            return DONE
        host_s = [(_host(t), peer) for t, peer in sends]
        host_r = [(torch.empty(t.shape, dtype=t.dtype), peer) for t, peer in recvs]
        works = [dist.isend(c, peer) for c, peer in host_s] + [dist.irecv(c, peer) for c, peer in host_r]

        def land():
            for w in works:
                w.wait()
            for (t, _), (c, _) in zip(recvs, host_r):
                t.copy_(c)

        return Work(land=land, kind="send_recv")


def backend_for(group=None):
    """The Backend of a process group (default group when None)."""
    if not dist.is_initialized():
        raise RuntimeError("torch.distributed is not initialised")
    name = dist.get_backend(group)
    if name == "nccl":
        return RcclBackend()
    if name == "gloo":
        return StagedGlooBackend()
    raise NotImplementedError(f"process-group backend {name!r}: 'nccl' (RCCL, product) or 'gloo' (tests)")
