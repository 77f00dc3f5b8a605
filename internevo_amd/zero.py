"""ZeRO-1 data-parallel exchange over RCCL/xGMI for the flat layout (one process per GPU).

Reference behaviour being matched:
  C1  all_reduce(AVG) of flattened gradient buckets from the AccumulateGrad hooks of the LAST
      micro-batch, overlapped with the rest of backward        hybrid_zero_optim.py:290-367,489-527
  C2  broadcast of every rank's updated bf16 partition          hybrid_zero_optim.py:809-837
  C3  all_reduce(SUM) of the squared-norm scalar over ZERO1     solver/optimizer/utils.py:352-357
MI355X redesign with identical values: xGMI is point-to-point (7 links per GPU), so instead of
all-reduce + broadcast (2x the necessary bytes, the owner needs only its shard) each bucket is
  reduce_scatter_tensor(AVG)  -> the owner rank's contiguous 1/world shard, in place in the flat grads
  all_gather_into_tensor      <- updated bf16 shards, in place in the flat params
launched asynchronously per bucket (c10d runs them on its own HIP stream and orders them against
the compute stream), waited only where the values are consumed.
Backends: every collective goes through a comm.Backend ("nccl" = RCCL, the product; "gloo" = the staged
test backend that lands results only in wait()): there is ONE code path here, the tests drive the very
calls, aliasing and wait() placement the RCCL run executes (comm.py).
"""
import torch
import torch.distributed as dist

from .comm import backend_for


def job_dp_groups(world_size, tp=1, pp=1):
    """Every data-parallel group of a job as lists of global ranks, in a job-wide fixed order: the ranks that hold the same
    model-parallel shard.  Tensor (or sequence-as-tensor) groups are consecutive ranks, so their data-parallel groups are the
    residue classes modulo tp (tensorpar.py); pipeline stages are blocks of consecutive ranks (pipeline.py), so a stage IS a
    data-parallel group (parallel_context.py:499-520, process_group_initializer.py); with both, the residue classes inside every stage."""
    per = world_size // pp   # ranks of a stage; inside it the tensor groups are the innermost dimension
    return [[s * per + r for r in range(t, per, tp)] for s in range(pp) for t in range(tp)]


class ZeroComm:
    def __init__(self, layout, group=None, world_size=1, rank=0, force_collectives=False, zero_size=None, dp_groups=None):
        """group / world_size / rank: the data-parallel group (ranks that hold the same model-parallel shard).
        zero_size = z < world_size: hybrid ZeRO ("ZeRO-1.5", parallel.zero1.size; parallel_context.py:499-520,
        process_group_initializer.py:249-329): the fp32 state is sharded over groups of z CONSECUTIVE data-parallel ranks and
        replicated across the world_size / z groups.  Gradients then take two hops with the same result as one average over the
        whole data-parallel group: reduce-scatter(AVG) inside the zero group, all-reduce(AVG) of the 1/z shard across the
        replicas (ranks with the same position in their zero group); parameters are all-gathered inside the zero group.
        `layout` must have been built for z shards.
        dp_groups: EVERY data-parallel group of the job (job_dp_groups), needed for hybrid ZeRO when the job has more than one
        (tensor or pipeline parallelism): new_group is collective over the whole job, so every rank creates every zero / replica
        group of every data-parallel group in the same order.  None = this group is the only one.
        force_collectives: issue the collectives even on a 1-rank group (they are identities there) -- lets a single-GPU
        test drive the exact RCCL call sequence of the multi-GPU path."""
        self.layout = layout
        self.dp_group = group
        self.dp_world = world_size
        self.dp_rank = rank
        z = world_size if not zero_size or zero_size < 0 or zero_size >= world_size else int(zero_size)
        if world_size % z:
            raise ValueError(f"parallel.zero1.size = {z} does not divide the data-parallel size {world_size}")
        self.world = z                      # shards per bucket
        self.rank = rank % z                # this rank's shard
        self.replica = rank // z            # which copy of the sharded state this rank's zero group holds
        self.n_replica = world_size // z
        self.group = group                  # the group the reduce-scatter / all-gather run over (the zero group)
        self.replica_group = None
        self.pending = []
        self.gathers = {}
        self.active = world_size > 1 or force_collectives
        self.side = None
        self.be = None
        if self.active:
            if not dist.is_initialized():
                raise RuntimeError("torch.distributed must be initialised for world_size > 1")
            self.be = backend_for(group)
            self.backend = self.be.name
            if z < world_size:
                members = dist.get_process_group_ranks(group) if group is not None else list(range(dist.get_world_size()))
                if len(members) != world_size:
                    raise ValueError(f"the data-parallel group has {len(members)} ranks, world_size says {world_size}")
                if dp_groups is None and len(members) != dist.get_world_size():
                    raise ValueError("hybrid ZeRO (zero_size < data-parallel size) over a SUB-group of the job needs dp_groups = every data-parallel group of the "
                                     "job (zero.job_dp_groups): dist.new_group is collective over the default group, all ranks must create the same groups")
                groups = [list(g) for g in dp_groups] if dp_groups is not None else [members]
                if members not in groups:
                    raise ValueError(f"this rank's data-parallel group {members} is not among the job's {groups}")
                me = members[rank]
                # every rank of the job takes part in the creation of every group, in the same order (new_group is collective
                # over the default group): callers construct ZeroComm on all ranks at the same point
                for others in groups:
                    for g0 in range(0, world_size, z):
                        ranks = others[g0 : g0 + z]
                        grp = dist.new_group(ranks)
                        if me in ranks:
                            self.group = grp
                    for j in range(z):
                        ranks = others[j::z]
                        grp = dist.new_group(ranks)
                        if me in ranks:
                            self.replica_group = grp
                assert self.group is not group and self.replica_group is not None
        else:
            self.backend = None

    def _side_stream(self, device):
        """The stream the second hop of hybrid ZeRO is ordered on (never the compute stream); None on a CPU test."""
        if device.type != "cuda":
            return None
        if self.side is None:
            self.side = torch.cuda.Stream(device=device)
        return self.side

    # ---- gradients: bucket -> averaged shard on its owner ----------------------------------------
    def reduce_scatter_async(self, full, shard):
        """full (a whole bucket of gradients) -> `shard` = this rank's 1/world part averaged over the WHOLE data-parallel group: one
        reduce-scatter(AVG), or under hybrid ZeRO reduce-scatter(AVG) inside the zero group + all-reduce(AVG) across the replicas, the
        second hop ordered on a side stream.  `shard` may be the rank's own slice of `full` (in place) or any other tensor of that size
        (weight parallelism: from a pool slot into the resident gradient shard).  Returns the Work to wait for."""
        if self.replica_group is None:
            return self.be.reduce_scatter(shard, full, self.group, avg=True)
        side = self._side_stream(full.device)
        if side is None:
            first = self.be.reduce_scatter(shard, full, self.group, avg=True)
            first.wait()
            return self.be.all_reduce(shard, self.replica_group, avg=True)
        cur = torch.cuda.current_stream(full.device)
        ready = torch.cuda.Event()
        ready.record(cur)                       # the bucket's last weight gradient is queued
        with torch.cuda.stream(side):
            side.wait_event(ready)
            first = self.be.reduce_scatter(shard, full, self.group, avg=True)
            first.wait()                        # orders the SIDE stream (not the compute stream) behind the first hop
            return self.be.all_reduce(shard, self.replica_group, avg=True)

    def reduce_bucket_async(self, grads_flat, bucket_index):
        if not self.active or self.layout.buckets[bucket_index].size == 0:   # (size 0: a bucket this pipeline stage does not own)
            return
        b = self.layout.buckets[bucket_index]
        full = grads_flat[b.start : b.start + b.size]
        s, n = b.shard(self.rank, self.world)
        self.pending.append(self.reduce_scatter_async(full, grads_flat[s : s + n]))   # the owner's slice OF `full`: in place

    # ---- parameters: updated shard -> every rank ---------------------------------------------------
    def all_gather_async(self, full, shard):
        """every zero-group rank's `shard` -> `full` (rank order); `shard` may be the rank's own slice of `full`."""
        return self.be.all_gather(full, shard, self.group)

    def gather_bucket_async(self, params_flat, bucket_index):
        if not self.active or self.layout.buckets[bucket_index].size == 0:
            return
        b = self.layout.buckets[bucket_index]
        full = params_flat[b.start : b.start + b.size]
        s, n = b.shard(self.rank, self.world)
        self.gathers[bucket_index] = self.all_gather_async(full, params_flat[s : s + n])   # in place

    def gather_full_bucket(self, local_flat, local_offset, bucket_index):
        """Checkpointing only (synchronous): this rank's slice of a bucket of optimizer state -> the whole bucket on every rank."""
        b = self.layout.buckets[bucket_index]
        n = b.size // self.world
        shard = local_flat[local_offset : local_offset + n]
        if not self.active or self.world == 1:
            return shard
        full = torch.empty(b.size, dtype=shard.dtype, device=shard.device)
        self.be.all_gather(full, shard.contiguous(), self.group).wait()
        return full

    def barrier(self):
        if self.active and self.dp_world > 1:
            dist.barrier(group=self.dp_group)

    def wait_gather(self, bucket_index):
        """Block the compute stream until the all-gather of this bucket's parameters (issued by the previous step) is
        done -- called by the forward right before the first kernel that reads the bucket, so the parameter exchange
        of step n overlaps the forward of step n+1 (the reference's overlap_sync_param idea, hybrid_zero_optim.py:831-834)."""
        work = self.gathers.pop(bucket_index, None)
        if work is not None:
            work.wait()

    def wait_all(self):
        for work in self.pending:
            work.wait()
        self.pending = []

    def wait_all_gathers(self):
        for b in list(self.gathers):
            self.wait_gather(b)

    def all_reduce_sum(self, t):
        """In-place sum over the ZERO group (the squared-norm scalar: every replica holds the same shards, so the sum over one
        zero group is the global value -- solver/optimizer/utils.py:352-357 reduces over ZERO1 too)."""
        if self.active:
            self.be.all_reduce(t, self.group).wait()
        return t

    def broadcast_params(self, params_flat, src=0):
        """sync_model_param at init (internlm/utils/parallel.py:71-107): every DP rank starts from rank 0's weights."""
        if self.active:
            if self.dp_group is not None:
                src = dist.get_global_rank(self.dp_group, src)  # `src` counts inside the data-parallel group
            self.be.broadcast(params_flat, src, self.dp_group).wait()
