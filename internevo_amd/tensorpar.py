"""Megatron-style tensor parallelism of the transformer layers (parallel.tensor = dict(size=tp, mode="mtp" | "msp" | "fsp")).

Reference behaviour being matched (model/ops/linear.py:205-337, model/utils.py:228-346, modeling_internlm2.py:86-189,
modules/mlp.py:100-140): wqkv / w1 / w3 are column-parallel (each rank owns the output rows of its 1/tp of the kv groups / FFN
units), wo / w2 are row-parallel (each rank owns the matching input columns); the row-parallel outputs are partial sums over
the tensor group and are all-reduced (4 x 33.5 MB per layer and micro-batch: two in forward, two in
backward for the input gradients of the column-parallel layers).  Norm weights are replicated.

Every rank of a tensor group reads the same micro-batches; data parallelism (and the ZeRO-1 sharding) runs over the ranks that
hold the same shard (rank % tp).

Output head and loss (`vocab_parallel`, default; the reference's `parallel_output=True`: ScaleColumnParallelLinear without
gather_output, ops/linear.py:124-153, + flash-attn's vocabulary-parallel CrossEntropyLoss, losses/ce_loss.py:26-36): every rank holds
V/tp rows of the head and computes [T, V/tp] logits; the cross-entropy runs on the local columns with the EXISTING fused kernels
(labels another rank owns become "valid, not here"), and ONE all-gather of two floats per token (local log-sum-exp, local target
logit) replaces flash-attn's three all-reduces (max, sum-exp, target); the backward uses the global log-sum-exp, and the head's
input gradient is summed over the group under its weight gradient.  `vocab_parallel=False` keeps the whole head on every rank
(logits computed redundantly).  The embedding: whole on every rank (0.76 GB of 288, no exchange) unless `model.embed_split_hidden` asks for the reference's
layout (modules/embedding.py:24-60: every rank holds h / tp columns, the looked-up rows are all-gathered along the hidden dimension, the
backward keeps its own columns of the gradient); losses and gradients are the same numbers.
"""
import torch
import torch.distributed as dist

from .comm import DONE, backend_for


class TensorParallel:
    def __init__(self, tp_size, rank, world_size, vocab_parallel=True, embed_split=False, stages=1, stage=0):
        """rank / world_size: this rank inside ONE pipeline stage and the stage's size (the whole job without pipeline parallelism); stages / stage:
        the pipeline size and this rank's stage -- stage s occupies the global ranks s * world_size ... (parallel_context.py: tensor is the innermost
        dimension, then data, then pipeline).  Every rank of the job creates every group of every stage, in the same order."""
        if world_size % tp_size != 0:
            raise ValueError(f"world size {world_size} is not a multiple of the tensor-parallel size {tp_size}")
        self.tp = tp_size
        self.vocab_parallel = bool(vocab_parallel) and tp_size > 1
        self.embed_split = bool(embed_split) and tp_size > 1   # model.embed_split_hidden: the embedding cut along the hidden dim
        self.tp_rank = rank % tp_size
        self.dp_rank = rank // tp_size
        self.dp_world = world_size // tp_size
        self.group = None      # the tensor group this rank belongs to
        self.dp_group = None   # ranks holding the same shard (gradient averaging / ZeRO-1)
        self.backend = None
        self.be = None
        if tp_size > 1:
            if not dist.is_initialized():
                raise RuntimeError("torch.distributed must be initialised for tensor parallelism")
            me = stage * world_size + rank
            for s_ in range(stages):
                base = s_ * world_size
                for g in range(world_size // tp_size):  # consecutive ranks form a tensor group (parallel_context.py: innermost dimension)
                    ranks = [base + r for r in range(g * tp_size, (g + 1) * tp_size)]
                    grp = dist.new_group(ranks)
                    if me in ranks:
                        self.group = grp
                for t in range(tp_size):                # the ranks of a stage that hold the same shard
                    ranks = [base + r for r in range(t, world_size, tp_size)]
                    grp = dist.new_group(ranks)
                    if me in ranks:
                        self.dp_group = grp
            self.be = backend_for(self.group)
            self.backend = self.be.name

    def all_reduce_sum_async(self, t):
        """Start the in-place sum over the tensor group (the `g` operator of a row-parallel output / a column-parallel input gradient)
        and return a handle; `.wait()` orders the CURRENT stream behind it.  On RCCL the collective runs on c10d's own HIP stream
        (which first waits for the work already queued on the current stream, i.e. for the GEMM that produced `t`), so kernels
        launched between this call and `.wait()` overlap it: the engine puts the weight-gradient GEMM of the same layer there, as
        the reference does with its column-parallel backward (model/utils.py:329-345: all_reduce(grad_input, async_op=True) ->
        wgrad -> handle.wait()).  `t` must not be touched before `.wait()` (the staged test backend lands the sum only there)."""
        if self.tp == 1:
            return DONE
        return self.be.all_reduce(t, self.group)

    def all_reduce_sum(self, t):
        self.all_reduce_sum_async(t).wait()
        return t

    # ---- sequence-sharded activations ("msp" / "fsp": model/utils.py:228-463, ops/linear.py:260-354) -------------------------------------
    def rows(self, T):
        """This rank's token rows of a [T, ...] activation (split_forward_gather_backward along the sequence, modules/embedding.py:57-58)."""
        if T % self.tp:
            raise ValueError(f"{T} token rows do not split over {self.tp} tensor ranks")
        n = T // self.tp
        return slice(self.tp_rank * n, (self.tp_rank + 1) * n)

    def reduce_scatter_rows_async(self, t):
        """The row-parallel output / column-parallel input gradient `t` [T, C] (a partial sum on every rank) -> summed over the tensor group
        into THIS rank's rows, in place (the other rows are dead afterwards).  Handle as all_reduce_sum_async."""
        if self.tp == 1:
            return DONE
        return self.be.reduce_scatter(t[self.rows(t.shape[0])], t, self.group, avg=False)

    def all_gather_rows_async(self, t):
        """Every rank's rows of `t` [T, C] -> the whole tensor on every rank, in place (all-gather in front of a column-parallel product,
        and of a row-parallel product's backward)."""
        if self.tp == 1:
            return DONE
        return self.be.all_gather(t, t[self.rows(t.shape[0])], self.group)

    def all_reduce_avg(self, t):
        """reduce_tensor(..., ParallelMode.TENSOR) of a norm weight's gradient under sequence-sharded activations: ReduceOp.AVG
        (solver/optimizer/utils.py:120) -- every rank's gradient covers its own rows, the reference averages (it does not sum) them."""
        if self.tp > 1:
            self.be.all_reduce(t, self.group, avg=True).wait()
        return t

    def all_gather(self, t):
        """[tp, *t.shape] tensor with every rank's `t` (rank order).  Small per-token statistics only (vocabulary-parallel loss)."""
        if self.tp == 1:
            return t.unsqueeze(0)
        out = torch.empty((self.tp,) + tuple(t.shape), dtype=t.dtype, device=t.device)
        self.be.all_gather(out, t.contiguous(), self.group).wait()
        return out

    def barrier(self):
        if self.tp > 1:
            dist.barrier(group=self.group)

    # ---- shard <-> full parameter ------------------------------------------------------------------------------------
    def shard(self, kind, full):
        """This rank's part of a full parameter tensor (kind as in layout.ParamSpec.kind)."""
        if self.tp == 1 or kind in ("norm", "bo") or (kind == "embed" and not getattr(self, "embed_split", False)) or (kind == "head" and not self.vocab_parallel):
            return full
        r, tp = self.tp_rank, self.tp
        if kind in ("wqkv", "bqkv", "w1", "w3", "head"):  # column-parallel: output rows (wqkv rows / bias elements are grouped by kv head: whole groups; head: vocabulary rows)
            n = full.shape[0] // tp
            return full[r * n : (r + 1) * n]
        n = full.shape[1] // tp               # row-parallel (wo, w2): input columns; the hidden-split embedding: hidden columns
        return full[:, r * n : (r + 1) * n]

    @staticmethod
    def unshard(kind, parts, vocab_parallel=True, embed_split=False):
        """Inverse of shard() given every rank's part in rank order."""
        if kind in ("norm", "bo") or (kind == "embed" and not embed_split) or (kind == "head" and not vocab_parallel) or len(parts) == 1:
            return parts[0]
        return torch.cat(parts, dim=0 if kind in ("wqkv", "bqkv", "w1", "w3", "head") else 1)
