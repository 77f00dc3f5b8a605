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
Backends: "nccl" (= RCCL on ROCm) is the product path.  "gloo" exists for the multi-process tests
(CPU tensors, or device tensors staged through the host): same bucket/shard arithmetic, synchronous.
"""
import torch
import torch.distributed as dist


class ZeroComm:
    def __init__(self, layout, group=None, world_size=1, rank=0, force_collectives=False, zero_size=None):
        """group / world_size / rank: the data-parallel group (ranks that hold the same model-parallel shard).
        zero_size = z < world_size: hybrid ZeRO ("ZeRO-1.5", parallel.zero1.size; parallel_context.py:499-520,
        process_group_initializer.py:249-329): the fp32 state is sharded over groups of z CONSECUTIVE data-parallel ranks and
        replicated across the world_size / z groups.  Gradients then take two hops with the same result as one average over the
        whole data-parallel group: reduce-scatter(AVG) inside the zero group, all-reduce(AVG) of the 1/z shard across the
        replicas (ranks with the same position in their zero group); parameters are all-gathered inside the zero group.
        `layout` must have been built for z shards.
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
        if self.active:
            if not dist.is_initialized():
                raise RuntimeError("torch.distributed must be initialised for world_size > 1")
            self.backend = dist.get_backend(group)
            if z < world_size:
                members = dist.get_process_group_ranks(group) if group is not None else list(range(dist.get_world_size()))
                # every rank of the job takes part in the creation of every group, in the same order (new_group is collective
                # over the default group): callers construct ZeroComm on all ranks at the same point
                for others in self._all_dp_groups(members):
                    for g0 in range(0, world_size, z):
                        ranks = others[g0 : g0 + z]
                        grp = dist.new_group(ranks)
                        if members[rank] in ranks and others == members:
                            self.group = grp
                    for j in range(z):
                        ranks = others[j::z]
                        grp = dist.new_group(ranks)
                        if members[rank] in ranks and others == members:
                            self.replica_group = grp
                if self.backend == "nccl":
                    # the second hop is ordered behind the first on a side stream, never on the compute stream
                    self.side = torch.cuda.Stream()
        else:
            self.backend = None

    @staticmethod
    def _all_dp_groups(members):
        """All data-parallel groups of the job, as lists of global ranks, in a job-wide fixed order.  Data-parallel groups are the
        residue classes of the global rank modulo the model-parallel size (tensorpar.py: rank % tp), i.e. strided by
        stride = members[1] - members[0]."""
        if len(members) < 2:
            return [members]
        stride = members[1] - members[0]
        return [[m - members[0] + o for m in members] for o in range(stride)]

    # ---- gradients: bucket -> averaged shard on its owner ----------------------------------------
    def reduce_bucket_async(self, grads_flat, bucket_index):
        if not self.active or self.layout.buckets[bucket_index].size == 0:   # (size 0: a bucket this pipeline stage does not own)
            return
        b = self.layout.buckets[bucket_index]
        full = grads_flat[b.start : b.start + b.size]
        s, n = b.shard(self.rank, self.world)
        shard = grads_flat[s : s + n]
        if self.backend == "nccl":
            if self.replica_group is None:
                work = dist.reduce_scatter_tensor(shard, full, op=dist.ReduceOp.AVG, group=self.group, async_op=True)
            else:
                cur = torch.cuda.current_stream(grads_flat.device)
                ready = torch.cuda.Event()
                ready.record(cur)                       # the bucket's last weight gradient is queued
                with torch.cuda.stream(self.side):
                    self.side.wait_event(ready)
                    first = dist.reduce_scatter_tensor(shard, full, op=dist.ReduceOp.AVG, group=self.group, async_op=True)
                    first.wait()                        # orders the SIDE stream (not the compute stream) behind the first hop
                    work = dist.all_reduce(shard, op=dist.ReduceOp.AVG, group=self.replica_group, async_op=True)
            self.pending.append((work, None))
        else:
            # test path: gloo has no AVG / in-place reduce-scatter and no device tensors -> host staging, SUM, scale
            src = full.detach().to("cpu", copy=True)
            tmp = torch.empty(n, dtype=src.dtype)
            dist.reduce_scatter_tensor(tmp, src, op=dist.ReduceOp.SUM, group=self.group)
            acc = tmp.float() / self.world
            if self.replica_group is not None:
                dist.all_reduce(acc, op=dist.ReduceOp.SUM, group=self.replica_group)
                acc /= self.n_replica
            shard.copy_(acc.to(shard.dtype))

    # ---- parameters: updated shard -> every rank ---------------------------------------------------
    def gather_bucket_async(self, params_flat, bucket_index):
        if not self.active or self.layout.buckets[bucket_index].size == 0:
            return
        b = self.layout.buckets[bucket_index]
        full = params_flat[b.start : b.start + b.size]
        s, n = b.shard(self.rank, self.world)
        shard = params_flat[s : s + n]
        if self.backend == "nccl":
            work = dist.all_gather_into_tensor(full, shard, group=self.group, async_op=True)
            self.gathers[bucket_index] = work
        else:
            src = shard.detach().to("cpu", copy=True)
            tmp = torch.empty(b.size, dtype=src.dtype)
            dist.all_gather_into_tensor(tmp, src, group=self.group)
            full.copy_(tmp)

    def gather_full_bucket(self, local_flat, local_offset, bucket_index):
        """Checkpointing only (synchronous): this rank's slice of a bucket of optimizer state -> the whole bucket on every rank.
        Returns a device tensor on the RCCL path, a host tensor on the gloo test path."""
        b = self.layout.buckets[bucket_index]
        n = b.size // self.world
        shard = local_flat[local_offset : local_offset + n]
        if not self.active or self.world == 1:
            return shard
        if self.backend == "nccl":
            full = torch.empty(b.size, dtype=shard.dtype, device=shard.device)
            dist.all_gather_into_tensor(full, shard.contiguous(), group=self.group)
            return full
        src = shard.detach().to("cpu", copy=True)
        tmp = torch.empty(b.size, dtype=src.dtype)
        dist.all_gather_into_tensor(tmp, src, group=self.group)
        return tmp

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
        for work, fin in self.pending:
            work.wait()
            if fin is not None:
                fin()
        self.pending = []

    def wait_all_gathers(self):
        for b in list(self.gathers):
            self.wait_gather(b)

    def all_reduce_sum(self, t):
        if self.active:
            if self.backend == "nccl":
                dist.all_reduce(t, op=dist.ReduceOp.SUM, group=self.group)
            else:
                c = t.detach().to("cpu", copy=True)
                dist.all_reduce(c, op=dist.ReduceOp.SUM, group=self.group)
                t.copy_(c)
        return t

    def broadcast_params(self, params_flat, src=0):
        """sync_model_param at init (internlm/utils/parallel.py:71-107): every DP rank starts from rank 0's weights."""
        if self.active:
            if self.dp_group is not None:
                src = dist.get_global_rank(self.dp_group, src)  # `src` counts inside the data-parallel group
            if self.backend == "nccl":
                dist.broadcast(params_flat, src=src, group=self.dp_group)
            else:
                c = params_flat.detach().to("cpu", copy=True)
                dist.broadcast(c, src=src, group=self.dp_group)
                params_flat.copy_(c)
