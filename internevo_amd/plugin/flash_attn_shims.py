"""flash-attn v2.2.1 call surface used by InternEvo, on the gfx950 flash kernels (K1) and fused CE (K4)."""
import torch
import torch.nn as nn

from .. import kernels as K


def _check_no_dropout(p):
    if p:
        raise NotImplementedError("attention dropout is not implemented (the path trains with attn_drop_rate = 0)")


class FlashAttnVarlenKVPackedFunc(torch.autograd.Function):
    """flash_attn.flash_attn_interface.FlashAttnVarlenKVPackedFunc (modeling_llama.py:297; the functional form is
    called at modeling_internlm2.py:446-468).  q [T, H, D], kv [T, 2, Hk, D]."""

    @staticmethod
    def forward(ctx, q, kv, cu_seqlens_q, cu_seqlens_k, max_seqlen_q, max_seqlen_k, dropout_p, softmax_scale, causal, return_softmax=False):
        _check_no_dropout(dropout_p)
        if cu_seqlens_q.data_ptr() != cu_seqlens_k.data_ptr():
            # self-attention only.  What the host knows is checked here; the element-wise comparison stays on the device (an asynchronous
            # assert: no device -> host synchronisation on the step's path, a mismatch surfaces as a device-side assertion)
            if cu_seqlens_q.shape != cu_seqlens_k.shape or q.shape[0] != kv.shape[0] or int(max_seqlen_q) != int(max_seqlen_k):
                raise NotImplementedError("self-attention only: cu_seqlens_q must equal cu_seqlens_k")
            torch._assert_async((cu_seqlens_q == cu_seqlens_k).all())
        if softmax_scale is None:
            softmax_scale = q.shape[-1] ** (-0.5)
        cu = cu_seqlens_q.to(torch.int32)
        out, lse = K.flash_attn_fwd(q, kv[:, 0], kv[:, 1], cu, int(max_seqlen_q), softmax_scale, causal)
        ctx.save_for_backward(q, kv, out, lse, cu)
        ctx.max_seqlen, ctx.scale, ctx.causal = int(max_seqlen_q), softmax_scale, causal
        return out

    @staticmethod
    def backward(ctx, dout, *args):
        q, kv, out, lse, cu = ctx.saved_tensors
        dkv = torch.empty_like(kv)
        dq, _, _ = K.flash_attn_bwd(dout.contiguous(), q, kv[:, 0], kv[:, 1], out, lse, cu, ctx.max_seqlen, ctx.scale, ctx.causal,
                                    None, dkv[:, 0], dkv[:, 1])
        return dq, dkv, None, None, None, None, None, None, None, None


def flash_attn_varlen_kvpacked_func(q, kv, cu_seqlens_q, cu_seqlens_k, max_seqlen_q, max_seqlen_k, dropout_p=0.0, softmax_scale=None,
                                    causal=False, return_attn_probs=False):
    if return_attn_probs:
        raise NotImplementedError("return_attn_probs")
    return FlashAttnVarlenKVPackedFunc.apply(q, kv, cu_seqlens_q, cu_seqlens_k, max_seqlen_q, max_seqlen_k, dropout_p, softmax_scale, causal)


class _QKVPackedFunc(torch.autograd.Function):
    """qkv [T, 3, H, D] (v1 model, multi_head_attention.py:381-392,646-659): same kernels, different strides."""

    @staticmethod
    def forward(ctx, qkv, cu_seqlens, max_seqlen, dropout_p, softmax_scale, causal):
        _check_no_dropout(dropout_p)
        if softmax_scale is None:
            softmax_scale = qkv.shape[-1] ** (-0.5)
        cu = cu_seqlens.to(torch.int32)
        out, lse = K.flash_attn_fwd(qkv[:, 0], qkv[:, 1], qkv[:, 2], cu, int(max_seqlen), softmax_scale, causal)
        ctx.save_for_backward(qkv, out, lse, cu)
        ctx.max_seqlen, ctx.scale, ctx.causal = int(max_seqlen), softmax_scale, causal
        return out

    @staticmethod
    def backward(ctx, dout):
        qkv, out, lse, cu = ctx.saved_tensors
        dqkv = torch.empty_like(qkv)
        K.flash_attn_bwd(dout.contiguous(), qkv[:, 0], qkv[:, 1], qkv[:, 2], out, lse, cu, ctx.max_seqlen, ctx.scale, ctx.causal,
                         dqkv[:, 0], dqkv[:, 1], dqkv[:, 2])
        return dqkv, None, None, None, None, None


def flash_attn_varlen_qkvpacked_func(qkv, cu_seqlens, max_seqlen, dropout_p=0.0, softmax_scale=None, causal=False, return_attn_probs=False):
    if return_attn_probs:
        raise NotImplementedError("return_attn_probs")
    return _QKVPackedFunc.apply(qkv, cu_seqlens, max_seqlen, dropout_p, softmax_scale, causal)


def _uniform_cu(batch, seqlen, device):
    return torch.arange(0, (batch + 1) * seqlen, seqlen, dtype=torch.int32, device=device)


class FlashSelfAttention(nn.Module):
    """flash_attn.modules.mha.FlashSelfAttention(causal, softmax_scale, attention_dropout)."""

    def __init__(self, causal=False, softmax_scale=None, attention_dropout=0.0):
        super().__init__()
        self.causal, self.softmax_scale = causal, softmax_scale
        self.drop = nn.Dropout(attention_dropout)

    def forward(self, qkv, causal=None, cu_seqlens=None, max_seqlen=None):
        causal = self.causal if causal is None else causal
        p = self.drop.p if self.training else 0.0
        if cu_seqlens is not None:  # packed [T, 3, H, D]
            return flash_attn_varlen_qkvpacked_func(qkv, cu_seqlens, max_seqlen, p, self.softmax_scale, causal)
        B, S = qkv.shape[:2]
        out = flash_attn_varlen_qkvpacked_func(qkv.reshape(B * S, *qkv.shape[2:]), _uniform_cu(B, S, qkv.device), S, p, self.softmax_scale, causal)
        return out.reshape(B, S, *out.shape[1:])


class FlashCrossAttention(nn.Module):
    """flash_attn.modules.mha.FlashCrossAttention; q [B, Sq, H, D], kv [B, Sk, 2, Hk, D] (Sq == Sk)."""

    def __init__(self, causal=False, softmax_scale=None, attention_dropout=0.0):
        super().__init__()
        self.causal, self.softmax_scale = causal, softmax_scale
        self.drop = nn.Dropout(attention_dropout)

    def forward(self, q, kv, causal=None, cu_seqlens=None, max_seqlen=None, cu_seqlens_k=None, max_seqlen_k=None):
        causal = self.causal if causal is None else causal
        p = self.drop.p if self.training else 0.0
        if cu_seqlens is not None:
            return flash_attn_varlen_kvpacked_func(q, kv, cu_seqlens, cu_seqlens if cu_seqlens_k is None else cu_seqlens_k, max_seqlen,
                                                   max_seqlen if max_seqlen_k is None else max_seqlen_k, p, self.softmax_scale, causal)
        B, Sq = q.shape[:2]
        if kv.shape[1] != Sq:
            raise NotImplementedError("FlashCrossAttention shim: seqlen_q must equal seqlen_k")
        cu = _uniform_cu(B, Sq, q.device)
        out = flash_attn_varlen_kvpacked_func(q.reshape(B * Sq, *q.shape[2:]), kv.reshape(B * Sq, *kv.shape[2:]), cu, cu, Sq, Sq, p,
                                              self.softmax_scale, causal)
        return out.reshape(B, Sq, *out.shape[1:])


def _group_all_gather(t, group):
    """[world, *t.shape]: every rank's `t` over `group` (per-token statistics of the vocabulary-parallel loss: small)."""
    import torch.distributed as dist

    world = dist.get_world_size(group)
    out = torch.empty((world,) + tuple(t.shape), dtype=t.dtype, device=t.device)
    from ..comm import backend_for

    backend_for(group).all_gather(out, t.contiguous(), group).wait()
    return out


class _CEFunc(torch.autograd.Function):
    @staticmethod
    def forward(ctx, logits, labels, smoothing, ignore_index, inplace_backward, group=None):
        if group is None:
            loss_rows, lse, _, _ = K.ce_fwd(logits, labels, ignore_index, smoothing)
            own = labels
        else:
            # vocabulary-parallel (flash-attn's SoftmaxCrossEntropyLossFn with a process group, reached from losses/ce_loss.py:26-36 when
            # parallel_output=True): `logits` are this rank's [rows, V / world] columns.  The fused kernel runs on them with the
            # labels mapped into the local range (-1: valid, owned by another rank), then ONE all-gather of (local log-sum-exp, local
            # target logit) per row gives the global log-sum-exp and the loss
            import torch.distributed as dist

            Vl, r = logits.shape[-1], dist.get_rank(group)
            here = (labels >= r * Vl) & (labels < (r + 1) * Vl)
            own = torch.where(labels == ignore_index, labels, torch.where(here, labels - r * Vl, torch.full_like(labels, -1)))
            if ignore_index == -1:
                raise NotImplementedError("vocabulary-parallel cross entropy uses -1 internally: ignore_index must differ")
            loss_rows, lse, _, _ = K.ce_fwd(logits, own, ignore_index, 0.0)
            stats = _group_all_gather(torch.stack([lse, torch.where(here, lse - loss_rows, torch.zeros_like(loss_rows))]), group)
            lse = torch.logsumexp(stats[:, 0], dim=0)
            loss_rows = torch.where(labels != ignore_index, lse - stats[:, 1].sum(dim=0), torch.zeros_like(lse))
        ctx.save_for_backward(logits, own, lse)
        ctx.smoothing, ctx.ignore_index, ctx.inplace = smoothing, ignore_index, inplace_backward
        ctx.mark_non_differentiable(lse)
        return loss_rows

    @staticmethod
    def backward(ctx, dloss_rows):
        # per-row upstream grads (ie_ce_bwd per-row mode): dlogits[r] = (softmax - onehot) * dloss_rows[r]; vocabulary-parallel: the
        # softmax uses the GLOBAL log-sum-exp, the one-hot term exists on the owning rank only
        logits, labels, lse = ctx.saved_tensors
        dlogits = logits if ctx.inplace else torch.empty_like(logits)
        K.ce_bwd(logits, labels, lse, dloss_rows.contiguous().float(), None, 1.0, ctx.ignore_index, ctx.smoothing, dlogits)
        return dlogits, None, None, None, None, None


class CrossEntropyLoss(nn.Module):
    """flash_attn.losses.cross_entropy.CrossEntropyLoss(ignore_index, reduction, label_smoothing, inplace_backward,
    process_group) as constructed at internlm/model/losses/ce_loss.py:31-36.  process_group of more than one rank: `input` holds this
    rank's vocabulary columns (rank r of the group owns [r V/world, (r+1) V/world)), `target` the global labels."""

    def __init__(self, ignore_index=-100, reduction="mean", label_smoothing=0.0, inplace_backward=False, process_group=None):
        super().__init__()
        if reduction not in ("mean", "none"):
            raise NotImplementedError("Only support reduction = 'mean' or 'none'")
        self.group = None
        if process_group is not None and torch.distributed.is_initialized() and torch.distributed.get_world_size(process_group) > 1:
            if label_smoothing > 0:
                raise NotImplementedError("label smoothing with the vocabulary-parallel cross entropy (its uniform term needs one more reduction)")
            self.group = process_group
        self.ignore_index, self.reduction, self.label_smoothing, self.inplace_backward = ignore_index, reduction, label_smoothing, inplace_backward

    def forward(self, input, target):
        assert input.is_cuda and target.is_cuda
        loss = _CEFunc.apply(input, target, self.label_smoothing, self.ignore_index, self.inplace_backward, self.group)
        if self.reduction == "mean":
            return loss.sum() / (target != self.ignore_index).sum()
        return loss


def dropout_add_layer_norm(*args, **kwargs):
    raise NotImplementedError("dropout_add_layer_norm is imported but asserted unused by the reference (modeling_internlm2.py:552)")


class ParallelFusedMLP(nn.Module):  # only referenced in isinstance checks / use_swiglu=False configs
    def __init__(self, *a, **k):
        raise NotImplementedError("ParallelFusedMLP (use_swiglu=False) is outside the hot path")


class ParallelGPT2Embeddings(nn.Module):  # only referenced in isinstance checks / embed_split_hidden=False configs
    def __init__(self, *a, **k):
        raise NotImplementedError("ParallelGPT2Embeddings (embed_split_hidden=False) is outside the hot path")


class VocabParallelEmbedding(nn.Module):
    def __init__(self, *a, **k):
        raise NotImplementedError("VocabParallelEmbedding is outside the hot path")
