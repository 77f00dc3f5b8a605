"""Thin torch-tensor wrappers over the C ABI (include/internevo_hip.h).

torch is plumbing here: it owns device memory and the current HIP stream.  Every function passes raw
device pointers + sizes + ``torch.cuda.current_stream().cuda_stream`` to libinternevo_hip.so and raises
``InternEvoHipError`` on a non-zero return code.  There is no CPU / eager fallback: a tensor that is not
on a HIP device is a ``ValueError``.
"""
import ctypes
import math

import torch

from . import _lib
from ._lib import IE_BF16, IE_F32, IeScalerConfig, IeStepState, InternEvoHipError, check  # noqa: F401  (InternEvoHipError re-exported for callers)

_DT = {torch.bfloat16: IE_BF16, torch.float32: IE_F32}


def _L():
    return _lib.load()


def _stream():
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


_EMPTY_ANCHOR = {}


def _p(t):
    if t is None:
        return ctypes.c_void_p(0)
    if not t.is_cuda:
        raise ValueError("internevo_amd kernels need HIP device tensors (no CPU fallback)")
    if t.numel() == 0:
        # torch hands out a NULL data pointer for empty tensors; the C ABI wants non-NULL pointers whatever the extent, so an
        # empty tensor is represented by a small live allocation of its device (never dereferenced: the extent is zero)
        a = _EMPTY_ANCHOR.get(t.device)
        if a is None:
            a = _EMPTY_ANCHOR[t.device] = torch.zeros(64, dtype=torch.uint8, device=t.device)
        return ctypes.c_void_p(a.data_ptr())
    return ctypes.c_void_p(t.data_ptr())


def _dt(t):
    try:
        return _DT[t.dtype]
    except KeyError:
        raise ValueError(f"unsupported dtype {t.dtype}") from None


def _contig(t, name):
    if not t.is_contiguous():
        raise ValueError(f"{name} must be contiguous")
    return t


# ------------------------------------------------------------------------------------------ RMSNorm
def rmsnorm_fwd(x, w, eps, y=None, rstd=None):
    """x [..., C] (bf16|fp32), w [C] -> (y [..., C] in w.dtype, rstd [rows] fp32)."""
    _contig(x, "x"); _contig(w, "w")
    C = x.shape[-1]
    rows = x.numel() // C
    if y is None:
        y = torch.empty(x.shape, dtype=w.dtype, device=x.device)
    if rstd is None:
        rstd = torch.empty(rows, dtype=torch.float32, device=x.device)
    check(_L().ie_rmsnorm_fwd(_p(x), _dt(x), _p(w), _dt(w), _p(y), _p(rstd), rows, C, eps, _stream()), "ie_rmsnorm_fwd")
    return y, rstd


def add_rmsnorm_fwd(a, b, w, eps, r_out=None, y=None, rstd=None):
    """r = bf16(a+b); y = RMSNorm(r).  Returns (r, y, rstd)."""
    _contig(a, "a"); _contig(b, "b")
    C = a.shape[-1]
    rows = a.numel() // C
    r = torch.empty_like(a) if r_out is None else r_out
    if y is None:
        y = torch.empty_like(a)
    if rstd is None:
        rstd = torch.empty(rows, dtype=torch.float32, device=a.device)
    check(_L().ie_add_rmsnorm_fwd(_p(a), _p(b), _p(r), _p(w), _p(y), _p(rstd), rows, C, eps, _stream()), "ie_add_rmsnorm_fwd")
    return r, y, rstd


def rmsnorm_bwd(dy, x, w, rstd, dres=None, dw_out=None, accumulate=False, partial_ws=None, dx=None):
    """Returns (dx [x.dtype], dw [w.dtype]).  dres (optional, x.dtype) is added to dx."""
    _contig(dy, "dy"); _contig(x, "x")
    C = x.shape[-1]
    rows = x.numel() // C
    L = _L()
    nparts = L.ie_rmsnorm_bwd_partials(rows)
    if partial_ws is None or partial_ws.numel() < nparts * C:
        partial_ws = torch.empty(nparts * C, dtype=torch.float32, device=x.device)
    if dx is None:
        dx = torch.empty_like(x)
    if dw_out is None:
        dw_out = torch.empty_like(w)
        accumulate = False
    check(L.ie_rmsnorm_bwd(_p(dy), _p(x), _dt(x), _p(w), _dt(w), _p(rstd), _p(dres), _p(dx), _p(partial_ws), _p(dw_out),
                           int(accumulate), rows, C, _stream()), "ie_rmsnorm_bwd")
    return dx, dw_out


# ------------------------------------------------------------------------------------------ rotary
def apply_rotary(x1, x2, cos, sin, out1, out2, conj):
    """rotary_emb.apply_rotary: x1/x2/out1/out2 [B, S, H, half] strided views (last dim stride 1),
    cos/sin [S, 1, half] or [S, half]."""
    if x1.dim() == 3:  # packed (total, heads, half)
        x1, x2, out1, out2 = (t.unsqueeze(0) for t in (x1, x2, out1, out2))
    B, S, H, half = x1.shape
    cos2 = cos.reshape(cos.shape[0], cos.shape[-1])
    sin2 = sin.reshape(sin.shape[0], sin.shape[-1])
    for t in (x1, x2, out1, out2, cos2, sin2):
        if t.stride(-1) != 1:
            raise ValueError("apply_rotary: last dim must have stride 1")
    if x1.stride() != x2.stride() or out1.stride() != out2.stride():
        raise ValueError("apply_rotary: x1/x2 (and out1/out2) must share strides")
    if cos2.stride(0) != sin2.stride(0):
        raise ValueError("apply_rotary: cos/sin must share strides")
    if cos2.shape[0] < S:
        raise ValueError("apply_rotary: cos/sin shorter than seqlen")
    check(_L().ie_apply_rotary(_p(x1), _p(x2), _p(cos2), _p(sin2), _p(out1), _p(out2), _dt(x1), B, S, H, half,
                               x1.stride(0), x1.stride(1), x1.stride(2), out1.stride(0), out1.stride(1), out1.stride(2),
                               cos2.stride(0), int(bool(conj)), _stream()), "ie_apply_rotary")


def qkv_rotary_fwd(qkv, cos, sin, pos, hkv, q_per_kv, d, interleaved=True, q_out=None, kv_out=None, q_scale=1.0):
    """qkv [T, hkv*(q_per_kv+2)*d] -> q [T, hkv*q_per_kv, d], kv [T, 2, hkv, d].  q_scale: factor applied to the rotated q in fp32 before its
    rounding to bf16 (the engine puts softmax_scale * log2 e there and calls attention with softmax_scale = ln 2)."""
    _contig(qkv, "qkv")
    T = qkv.numel() // (hkv * (q_per_kv + 2) * d)
    if q_out is None:
        q_out = torch.empty((T, hkv * q_per_kv, d), dtype=qkv.dtype, device=qkv.device)
    if kv_out is None:
        kv_out = torch.empty((T, 2, hkv, d), dtype=qkv.dtype, device=qkv.device)
    check(_L().ie_qkv_rotary_fwd_scaled(_p(qkv), _p(cos), _p(sin), _p(pos), _p(q_out), _p(kv_out), T, hkv, q_per_kv, d, int(interleaved),
                                        float(q_scale), _stream()), "ie_qkv_rotary_fwd_scaled")
    return q_out, kv_out


def linear_qkv_rotary_fwd(x, wqkv, cos, sin, pos, hkv, q_per_kv, d, interleaved, q_out, kv_out, qkv_scratch, q_scale=1.0):
    """q_out [T, hkv q_per_kv, d], kv_out [T, 2, hkv, d] = GQA split + rotary of x [T, K] @ wqkv [N, K]^T in ONE launch where the persistent GEMM frame takes the
    product and d = 128 (ie_gemm_qkv_rotary_fwd: the split / rotation sits in the product's epilogue, the [T, N] product is never written), else the product into
    qkv_scratch [T, N] + qkv_rotary_fwd; bit-identical either way."""
    M, Kd = x.shape
    N = hkv * (q_per_kv + 2) * d
    if wqkv.shape != (N, Kd) or qkv_scratch.shape != (M, N) or any(t.stride(-1) != 1 for t in (x, wqkv, qkv_scratch)) or not q_out.is_contiguous() or not kv_out.is_contiguous():
        raise ValueError("linear_qkv_rotary_fwd: bad shapes")
    prof = GEMM_PROFILER
    if prof is not None:
        prof.begin()
    check(_L().ie_gemm_qkv_rotary_fwd(_p(x), x.stride(0), _p(wqkv), wqkv.stride(0), _p(cos), _p(sin), _p(pos), _p(q_out), _p(kv_out), _p(qkv_scratch),
                                      qkv_scratch.stride(0), M, hkv, q_per_kv, d, Kd, int(interleaved), float(q_scale), _stream()), "ie_gemm_qkv_rotary_fwd")
    if prof is not None:
        prof.end(2.0 * M * N * Kd, 2.0 * (M * Kd + N * Kd + M * N))
    return q_out, kv_out


def qkv_rotary_bwd(dq, dkv, cos, sin, pos, hkv, q_per_kv, d, interleaved=True, dqkv_out=None, dq_scale=1.0):
    _contig(dq, "dq"); _contig(dkv, "dkv")
    T = dq.numel() // (hkv * q_per_kv * d)
    if dqkv_out is None:
        dqkv_out = torch.empty((T, hkv * (q_per_kv + 2) * d), dtype=dq.dtype, device=dq.device)
    check(_L().ie_qkv_rotary_bwd_scaled(_p(dq), _p(dkv), _p(cos), _p(sin), _p(pos), _p(dqkv_out), T, hkv, q_per_kv, d, int(interleaved),
                                        float(dq_scale), _stream()), "ie_qkv_rotary_bwd_scaled")
    return dqkv_out


# ------------------------------------------------------------------------------------------ SwiGLU
def _rows_ld(t):
    if t.dim() != 2 or t.stride(1) != 1:
        raise ValueError("expected a 2-D tensor with unit column stride")
    return t.shape[0], t.shape[1], t.stride(0)


def swiglu_fwd(a, b, out=None):
    rows, cols, lda = _rows_ld(a)
    _, _, ldb = _rows_ld(b)
    if out is None:
        out = torch.empty((rows, cols), dtype=a.dtype, device=a.device)
    check(_L().ie_swiglu_fwd(_p(a), lda, _p(b), ldb, _p(out), out.stride(0), rows, cols, _stream()), "ie_swiglu_fwd")
    return out


def swiglu_bwd(dout, a, b, da=None, db=None, act_out=None):
    rows, cols, lda = _rows_ld(a)
    _, _, ldb = _rows_ld(b)
    _, _, lddo = _rows_ld(dout)
    if da is None:
        da = torch.empty((rows, cols), dtype=a.dtype, device=a.device)
    if db is None:
        db = torch.empty((rows, cols), dtype=a.dtype, device=a.device)
    check(_L().ie_swiglu_bwd(_p(dout), lddo, _p(a), lda, _p(b), ldb, _p(da), da.stride(0), _p(db), db.stride(0), _p(act_out),
                             act_out.stride(0) if act_out is not None else 0, rows, cols, _stream()), "ie_swiglu_bwd")
    return da, db


# ------------------------------------------------------------------------------------------ Ulysses exchange layout
def seq_head_permute(x, out, A, B, S, C, inverse=False):
    """bf16 [A][B][S][C] -> [S][A][B][C] (or back): the send-side copy of the sequence<->head all-to-all."""
    _contig(x, "x"); _contig(out, "out")
    assert x.numel() == A * B * S * C == out.numel()
    check(_L().ie_seq_head_permute(_p(x), _p(out), A, B, S, C, int(inverse), _stream()), "ie_seq_head_permute")
    return out


def scale_bf16(x, factor):
    _contig(x, "x")
    check(_L().ie_scale_bf16(_p(x), x.numel(), float(factor), _stream()), "ie_scale_bf16")
    return x


def grad_scale_mix(x, scale):
    """x <- scale * x + (1 - scale) * x.detach() in place: the value of the reference's bf16 expression (modeling_internlm2.py:970-973)."""
    _contig(x, "x")
    check(_L().ie_grad_scale_mix(_p(x), x.numel(), float(scale), _stream()), "ie_grad_scale_mix")
    return x


def head_weight_fwd(w, scale, norm_head, out, inv_norm=None):
    """The weight ScaleColumnParallelLinearWithNormHead multiplies by (ops/linear.py:124-153): out = F.normalize(scale w + (1 - scale) w.detach())."""
    rows, cols = w.shape
    assert out.shape == w.shape and w.stride(1) == 1 and out.stride(1) == 1 and (inv_norm is not None or not norm_head)
    check(_L().ie_head_weight_fwd(_p(w), w.stride(0), _p(out), out.stride(0), _p(inv_norm) if inv_norm is not None else None, rows, cols, float(scale),
                                  int(bool(norm_head)), _stream()), "ie_head_weight_fwd")
    return out


def head_weight_bwd(dy, y, inv_norm, scale, norm_head, dw, accumulate):
    """dw (= or +=) the gradient of head_weight_fwd's input, from dy = the gradient w.r.t. its output y."""
    rows, cols = y.shape
    assert dy.shape == y.shape == dw.shape and dy.stride(1) == 1 and y.stride(1) == 1 and dw.stride(1) == 1
    check(_L().ie_head_weight_bwd(_p(dy), dy.stride(0), _p(y), y.stride(0), _p(inv_norm) if inv_norm is not None else None, _p(dw), dw.stride(0), rows, cols,
                                  float(scale), int(bool(norm_head)), int(bool(accumulate)), _stream()), "ie_head_weight_bwd")
    return dw


# ------------------------------------------------------------------------------------------ CE
def ce_fwd(logits, labels, ignore_index=-100, label_smoothing=0.0, loss_rows=None, lse=None, out=None, argmax_rows=None, nll_rows=None):
    """logits [rows, V] (bf16|fp32, row stride arbitrary), labels int64 [rows] ->
    (loss_rows fp32, lse fp32, loss_mean fp32[1], count fp32[1]).
    With argmax_rows (int32 [rows]) and nll_rows (fp32 [rows]) the metric variant of the kernel fills them too."""
    rows, V, ld = _rows_ld(logits)
    _contig(labels, "labels")
    dev = logits.device
    if loss_rows is None:
        loss_rows = torch.empty(rows, dtype=torch.float32, device=dev)
    if lse is None:
        lse = torch.empty(rows, dtype=torch.float32, device=dev)
    if out is None:
        out = torch.empty(2, dtype=torch.float32, device=dev)
    L = _L()
    if argmax_rows is not None:
        assert argmax_rows.dtype == torch.int32 and nll_rows is not None and nll_rows.dtype == torch.float32
        check(L.ie_ce_fwd_metric(_p(logits), _dt(logits), ld, _p(labels), _p(loss_rows), _p(lse), _p(argmax_rows), _p(nll_rows), rows, V,
                                 ignore_index, label_smoothing, _stream()), "ie_ce_fwd_metric")
    else:
        check(L.ie_ce_fwd(_p(logits), _dt(logits), ld, _p(labels), _p(loss_rows), _p(lse), rows, V, ignore_index, label_smoothing, _stream()),
              "ie_ce_fwd")
    check(L.ie_ce_mean(_p(loss_rows), _p(labels), rows, ignore_index, _p(out[0:1]), _p(out[1:2]), _stream()), "ie_ce_mean")
    return loss_rows, lse, out[0:1], out[1:2]


def ce_mean(loss_rows, labels, ignore_index, out):
    """out[0] = mean of loss_rows over the rows whose label is not ignore_index, out[1] = their count (fixed-order reduction)."""
    check(_L().ie_ce_mean(_p(loss_rows), _p(labels), labels.numel(), ignore_index, _p(out[0:1]), _p(out[1:2]), _stream()), "ie_ce_mean")
    return out


def metric_accumulate(nll_rows, argmax_rows, labels, type_ids, facc, ds_right=None, ds_tokens=None, ds_loss=None, ds_token_num=None,
                      ignore_index=-100):
    """One micro-batch into the AccPerplex / LossWithTypeId accumulators (see include/internevo_hip.h)."""
    ntypes = 0 if ds_right is None else ds_right.numel()
    assert facc.dtype == torch.float32 and facc.numel() >= 5
    if ntypes:
        assert type_ids is not None and type_ids.dtype == torch.int64 and type_ids.numel() == labels.numel()
        assert ds_right.dtype == torch.int64 and ds_tokens.dtype == torch.int64
        _contig(type_ids, "type_ids")
    check(_L().ie_metric_accumulate(_p(nll_rows), _p(argmax_rows), _p(labels), _p(type_ids) if ntypes else None, labels.numel(), ignore_index, ntypes,
                                    _p(facc), _p(ds_right) if ntypes else None, _p(ds_tokens) if ntypes else None,
                                    _p(ds_loss) if ntypes else None, _p(ds_token_num) if ntypes else None, _stream()), "ie_metric_accumulate")


def ce_bwd(logits, labels, lse, dloss, count, dloss_mul=1.0, ignore_index=-100, label_smoothing=0.0, dlogits=None):
    """dlogits (in place over logits when dlogits is None).  dloss/count are device fp32 scalars."""
    rows, V, ld = _rows_ld(logits)
    if dlogits is None:
        dlogits = logits
    check(_L().ie_ce_bwd(_p(logits), _p(dlogits), _dt(logits), ld, _p(labels), _p(lse), _p(dloss), dloss_mul, _p(count), rows, V,
                         ignore_index, label_smoothing, _stream()), "ie_ce_bwd")
    return dlogits


# ------------------------------------------------------------------------------------------ L2 norm
def sumsq(tensors, out=None, accumulate=False, partial_ws=None):
    """Squared L2 norm of a tensor or list of tensors (fp32 accumulate) -> fp32[1] on device."""
    if torch.is_tensor(tensors):
        tensors = [tensors]
    L = _L()
    maxp = L.ie_sumsq_max_partials()
    dev = tensors[0].device
    if partial_ws is None or partial_ws.numel() < maxp * len(tensors):
        partial_ws = torch.empty(maxp * len(tensors), dtype=torch.float32, device=dev)
    if out is None:
        out = torch.zeros(1, dtype=torch.float32, device=dev)
        accumulate = False
    off = 0
    n_out = ctypes.c_int64(0)
    for t in tensors:
        t = _contig(t, "tensor")
        check(L.ie_sumsq_partial(_p(t), _dt(t), t.numel(), _p(partial_ws), off, ctypes.byref(n_out), _stream()), "ie_sumsq_partial")
        off += n_out.value
    check(L.ie_sumsq_finish(_p(partial_ws), off, _p(out), int(accumulate), _stream()), "ie_sumsq_finish")
    return out


# ------------------------------------------------------------------------------------------ step control + AdamW
STATE_BYTES = ctypes.sizeof(IeStepState)


def step_state_new(device, initial_scale):
    st = torch.zeros(STATE_BYTES, dtype=torch.uint8, device=device)
    check(_L().ie_step_state_init(_p(st), float(initial_scale), _stream()), "ie_step_state_init")
    return st


def step_state_read(st):
    """Host copy of the device step state (this DOES synchronise; use for logging/tests only)."""
    raw = bytes(st.cpu().numpy().tobytes())
    return IeStepState.from_buffer_copy(raw)


def step_control(st, sumsq_dev, cfg: IeScalerConfig):
    check(_L().ie_step_control(_p(st), _p(sumsq_dev), ctypes.byref(cfg), _stream()), "ie_step_control")


def adamw_step(g, p32, m, v, p16, st, lr, beta1, beta2, eps, weight_decay):
    n = p32.numel()
    if g.numel() != n or m.numel() != n or v.numel() != n or (p16 is not None and p16.numel() != n):
        raise ValueError("adamw_step: size mismatch")
    check(_L().ie_adamw_step(_p(g), _dt(g), _p(p32), _p(m), _p(v), _p(p16), n, _p(st), lr, beta1, beta2, eps, weight_decay, _stream()),
          "ie_adamw_step")


def tune_adamw_cus(cus):
    """How many CUs the following adamw_step launches of this thread's process may occupy (0 = the whole chip); include/internevo_hip.h."""
    check(_L().ie_tune_adamw_cus(int(cus)), "ie_tune_adamw_cus")


def step_control_groups(st, sumsq_dev, cfg: IeScalerConfig, group_inv, group_norm):
    """HybridZeroOptimizer._step with several parameter groups: one overflow decision / scaler update over all of them, every group unscaled and
    clipped by its OWN norm (hybrid_zero_optim.py:760-779,863-876).  sumsq_dev [n]; group_inv / group_norm [n] fp32 outputs on the device."""
    n = sumsq_dev.numel()
    if group_inv.numel() != n or group_norm.numel() != n:
        raise ValueError("step_control_groups: size mismatch")
    check(_L().ie_step_control_groups(_p(st), _p(sumsq_dev), n, ctypes.byref(cfg), _p(group_inv), _p(group_norm), _stream()), "ie_step_control_groups")


def adamw_step_group(g, p32, m, v, p16, st, group_inv, lr, beta1, beta2, eps, weight_decay):
    """adamw_step with the gradient factor of a parameter group (group_inv: ONE float on the device, an element of step_control_groups' output)."""
    n = p32.numel()
    if g.numel() != n or m.numel() != n or v.numel() != n or (p16 is not None and p16.numel() != n) or group_inv.numel() != 1:
        raise ValueError("adamw_step_group: size mismatch")
    check(_L().ie_adamw_step_group(_p(g), _dt(g), _p(p32), _p(m), _p(v), _p(p16), n, _p(st), _p(group_inv), lr, beta1, beta2, eps, weight_decay, _stream()),
          "ie_adamw_step_group")


# ------------------------------------------------------------------------------------------ embedding / elementwise
def embedding_fwd(weight, ids, out=None):
    V, dim = weight.shape
    T = ids.numel()
    if out is None:
        out = torch.empty((T, dim), dtype=weight.dtype, device=weight.device)
    check(_L().ie_embedding_fwd(_p(weight), _p(ids), _p(out), T, V, dim, _stream()), "ie_embedding_fwd")
    return out


def embedding_bwd(dout, ids, dweight, accumulate, present_ws=None):
    V, dim = dweight.shape
    T = ids.numel()
    if present_ws is None or present_ws.numel() < V + 1 + T:
        present_ws = torch.empty(V + 1 + T, dtype=torch.int32, device=dweight.device)
    check(_L().ie_embedding_bwd(_p(dout), _p(ids), _p(dweight), _p(present_ws), T, V, dim, int(accumulate), _stream()), "ie_embedding_bwd")
    return dweight


def add_bf16(a, b, out=None):
    if out is None:
        out = torch.empty_like(a)
    check(_L().ie_add_bf16(_p(a), _p(b), _p(out), a.numel(), _stream()), "ie_add_bf16")
    return out


def cast(src, dtype, out=None):
    if out is None:
        out = torch.empty(src.shape, dtype=dtype, device=src.device)
    check(_L().ie_cast(_p(src), _dt(src), _p(out), _dt(out), src.numel(), _stream()), "ie_cast")
    return out


# ------------------------------------------------------------------------------------------ GEMM
def gemm(A, B, a_kmajor=False, b_kmajor=False, out=None, accumulate=False, variant=-1):
    """C[M,N] = op(A) @ op(B)  (bf16 in, fp32 accumulate, bf16 out).
    a_kmajor=False: A is [M,K];  True: A is [K,M].   b_kmajor=False: B is [N,K];  True: B is [K,N]."""
    if A.dim() != 2 or B.dim() != 2 or A.stride(1) != 1 or B.stride(1) != 1:
        raise ValueError("gemm: 2-D operands with unit column stride expected")
    if a_kmajor:
        K, M = A.shape
    else:
        M, K = A.shape
    if b_kmajor:
        Kb, N = B.shape
    else:
        N, Kb = B.shape
    if K != Kb:
        raise ValueError(f"gemm: contraction mismatch {K} vs {Kb}")
    if out is None:
        out = torch.empty((M, N), dtype=torch.bfloat16, device=A.device)
        accumulate = False
    if out.shape != (M, N) or out.stride(1) != 1:
        raise ValueError("gemm: bad output")
    if K == 0 or M == 0 or N == 0:  # empty contraction: the product is zero
        if not accumulate:
            out.zero_()
        return out
    prof = GEMM_PROFILER
    if prof is not None:
        prof.begin()
    if variant < 0:
        check(_L().ie_gemm_bf16(_p(A), A.stride(0), int(a_kmajor), _p(B), B.stride(0), int(b_kmajor), _p(out), out.stride(0), M, N, K,
                                int(accumulate), _stream()), "ie_gemm_bf16")
    else:
        check(_L().ie_gemm_bf16_tile(int(variant), _p(A), A.stride(0), int(a_kmajor), _p(B), B.stride(0), int(b_kmajor), _p(out),
                                     out.stride(0), M, N, K, int(accumulate), _stream()), "ie_gemm_bf16_tile")
    if prof is not None:
        prof.end(2.0 * M * N * K, 2.0 * (M * K + N * K + M * N))
    return out


def fp8_quantize(x, per_slice=False, out=None):
    """A contiguous bf16 tensor -> (q: uint8 tensor of the same shape holding OCP e4m3 bits, dequant: fp32 [count] on the device) with per-tensor dynamic scaling:
    q = e4m3(x * 448 / max|x|), dequant = max|x| / 448 -- two launches (ie_fp8_amax, ie_fp8_quantize), the scale never leaves the device.
    per_slice: x[z] for z in range(x.shape[0]) are `count` tensors with a scale each (the experts' blocks / weights).  out = (q, amax, dequant): buffers to use."""
    if x.dtype != torch.bfloat16 or not x.is_contiguous():
        raise ValueError("fp8_quantize: a contiguous bf16 tensor expected")
    count = x.shape[0] if per_slice else 1
    n = x.numel() // max(count, 1)
    if out is None:
        q = torch.empty(x.shape, dtype=torch.uint8, device=x.device)
        amax = torch.zeros(count, dtype=torch.float32, device=x.device)
        dequant = torch.empty(count, dtype=torch.float32, device=x.device)
    else:
        q, amax, dequant = out
        amax.zero_()
    check(_L().ie_fp8_amax(_p(x), n, count, _p(amax), _stream()), "ie_fp8_amax")
    check(_L().ie_fp8_quantize(_p(x), n, count, _p(amax), _p(q), _p(dequant), _stream()), "ie_fp8_quantize")
    return q, dequant


def gemm_fp8(Aq, a_dequant, Bq, b_dequant, out=None, accumulate=False):
    """C[M, N] bf16 = (Aq[M, K] e4m3)(Bq[N, K] e4m3)^T * a_dequant * b_dequant (fp32 accumulation; ie_gemm_fp8).  K % 128 == 0."""
    if Aq.dtype != torch.uint8 or Bq.dtype != torch.uint8 or Aq.dim() != 2 or Bq.dim() != 2 or Aq.stride(1) != 1 or Bq.stride(1) != 1:
        raise ValueError("gemm_fp8: 2-D uint8 (e4m3) operands with unit column stride expected")
    (M, K), (N, Kb) = Aq.shape, Bq.shape
    if K != Kb:
        raise ValueError(f"gemm_fp8: contraction mismatch {K} vs {Kb}")
    if out is None:
        out = torch.empty((M, N), dtype=torch.bfloat16, device=Aq.device)
        accumulate = False
    if out.shape != (M, N) or out.stride(1) != 1 or out.dtype != torch.bfloat16:
        raise ValueError("gemm_fp8: bad output")
    check(_L().ie_gemm_fp8(_p(Aq), Aq.stride(0), _p(Bq), Bq.stride(0), _p(out), out.stride(0), M, N, K, _p(a_dequant), _p(b_dequant), int(accumulate), _stream()),
          "ie_gemm_fp8")
    return out


def gemm_fp8_batched(Aq, a_dequant, Bq, b_dequant, out, accumulate=False):
    """out[z] = (Aq[z] e4m3)(Bq[z] e4m3)^T * a_dequant[z] * b_dequant[z] for the Z products of a strided batch in ONE launch (ie_gemm_fp8_batched):
    Aq [Z, M, K], Bq [Z, N, K] uint8, out [Z, M, N] bf16, the scales fp32 [Z]."""
    if Aq.dim() != 3 or Bq.dim() != 3 or out.dim() != 3 or Aq.dtype != torch.uint8 or Bq.dtype != torch.uint8 or out.dtype != torch.bfloat16:
        raise ValueError("gemm_fp8_batched: Aq [Z, M, K], Bq [Z, N, K] uint8 and out [Z, M, N] bf16 expected")
    (Z, M, K), (Zb, N, Kb) = Aq.shape, Bq.shape
    if Z != Zb or K != Kb or out.shape != (Z, M, N) or Aq.stride(2) != 1 or Bq.stride(2) != 1 or out.stride(2) != 1 or a_dequant.numel() < Z or b_dequant.numel() < Z:
        raise ValueError("gemm_fp8_batched: shapes / strides do not match")
    check(_L().ie_gemm_fp8_batched(_p(Aq), Aq.stride(1), Aq.stride(0), _p(Bq), Bq.stride(1), Bq.stride(0), _p(out), out.stride(1), out.stride(0), Z, M, N, K,
                                   _p(a_dequant), _p(b_dequant), int(accumulate), _stream()), "ie_gemm_fp8_batched")
    return out


class KernelProfiler:
    """Times every launch of one kernel class with HIP events recorded on the launch stream (torch's
    current stream) and sums algorithmic flops / bytes.  Used by bench.py for the `roofline` object."""

    def __init__(self):
        self.pairs = []
        self.flops = 0.0
        self.bytes = 0.0
        self._s = None
        self.kernels = set()          # the kernel instantiations the timed launches went to (ie_gemm_last_kernel)
        self._buf = ctypes.create_string_buffer(128)

    def begin(self):
        self._s = torch.cuda.Event(enable_timing=True)
        self._s.record()

    def end(self, flops, nbytes):
        e = torch.cuda.Event(enable_timing=True)
        e.record()
        self.pairs.append((self._s, e))
        self.flops += flops
        self.bytes += nbytes
        if _L().ie_gemm_last_kernel(self._buf, 128) == 0:
            self.kernels.add(self._buf.value.decode())

    def summary(self):
        torch.cuda.synchronize()
        sec = sum(s.elapsed_time(e) for s, e in self.pairs) * 1e-3
        n = len(self.pairs)
        return {"launches": n, "seconds": sec, "avg_us": sec / max(n, 1) * 1e6, "flops": self.flops, "bytes": self.bytes, "kernels": sorted(self.kernels)}


GEMM_PROFILER = None


LINEAR_FWD_VARIANT = -1  # tile variant forced on the forward product (A/B runs of bench.py --fwd-variant); -1 = the library's choice


def linear_fwd(x, w, out=None):
    """y[T,N] = x[T,K] @ w[N,K]^T"""
    if LINEAR_FWD_VARIANT == -100:
        # YARDSTICK ONLY (bench.py --fwd-variant -100, never a default): the forward products through torch.mm = hipBLASLt, to price this repo's forward
        # kernel against the library's INSIDE the training step (tools/hipblaslt_probe.py compares them in isolation).  Not a product path, not a fallback.
        return torch.mm(x, w.t(), out=out) if out is not None else torch.mm(x, w.t())
    return gemm(x, w, False, False, out, False, LINEAR_FWD_VARIANT)


def linear_fwd_add(x, w, addend, out):
    """out[T, N] = bf16(bf16(x[T, K] @ w[N, K]^T) + addend[T, N]) in ONE launch where the persistent GEMM frame takes the product (ie_linear_fwd_add: the block's
    residual add in the epilogue of wo / w2).  Returns False -- and touches nothing -- elsewhere (the caller then adds in the norm kernel: the same bits)."""
    M, Kd = x.shape
    N = w.shape[0]
    if w.shape != (N, Kd) or addend.shape != (M, N) or out.shape != (M, N) or any(t.stride(1) != 1 for t in (x, w, addend, out)) or addend.stride(0) != out.stride(0):
        raise ValueError("linear_fwd_add: bad shapes")
    L = _L()
    if LINEAR_FWD_VARIANT != -1 or not L.ie_gemm_dma_persistent_takes(M, N, Kd) or any(t.stride(0) % 8 for t in (x, w, out)):
        return False
    prof = GEMM_PROFILER
    if prof is not None:
        prof.begin()
    check(L.ie_linear_fwd_add(_p(x), x.stride(0), _p(w), w.stride(0), _p(addend), _p(out), out.stride(0), M, N, Kd, _stream()), "ie_linear_fwd_add")
    if prof is not None:   # (the product's algorithmic work; the addend's 2 M N bytes ride along)
        prof.end(2.0 * M * N * Kd, 2.0 * (M * Kd + N * Kd + M * N))
    return True


def linear_dgrad(dy, w, out=None):
    """dx[T,K] = dy[T,N] @ w[N,K]"""
    return gemm(dy, w, False, True, out)


def linear_wgrad(dy, x, out=None, accumulate=False):
    """dw[N,K] = dy[T,N]^T @ x[T,K]"""
    return gemm(dy, x, True, True, out, accumulate)


def linear_swiglu_fwd(x, w13, h13, act):
    """h13[T, 2F] = x[T, K] @ w13[2F, K]^T (w1 rows, then w3 rows) and act[T, F] = silu(h13[:, :F]) * h13[:, F:] -- one launch when the shape
    rides on the refill-schedule GEMM (ie_gemm_swiglu_fwd), the product + swiglu_fwd otherwise; bit-identical either way."""
    M, K = x.shape
    F = w13.shape[0] // 2
    if w13.shape != (2 * F, K) or h13.shape != (M, 2 * F) or act.shape != (M, F) or any(t.stride(1) != 1 for t in (x, w13, h13, act)):
        raise ValueError("linear_swiglu_fwd: bad shapes")
    prof = GEMM_PROFILER
    if prof is not None:
        prof.begin()
    check(_L().ie_gemm_swiglu_fwd(_p(x), x.stride(0), _p(w13), w13.stride(0), _p(h13), h13.stride(0), _p(act), act.stride(0), M, F, K, _stream()),
          "ie_gemm_swiglu_fwd")
    if prof is not None:   # (the product's algorithmic work; the gate's 6 F bytes per row ride along)
        prof.end(2.0 * M * 2 * F * K, 2.0 * (M * K + 2 * F * K + M * 2 * F))
    return h13, act


def linear_dgrad_swiglu_bwd(dy, w2, h13, dh13, dact_scratch):
    """dh13[T, 2F] = SwiGLU backward at h13 of d(act) = dy[T, K] @ w2[K, F] (w2 = the [h, F] weight of the down projection)."""
    M, K = dy.shape
    F = w2.shape[1]
    if w2.shape != (K, F) or h13.shape != (M, 2 * F) or dh13.shape != (M, 2 * F) or dact_scratch.shape != (M, F) or any(
            t.stride(1) != 1 for t in (dy, w2, h13, dh13, dact_scratch)):
        raise ValueError("linear_dgrad_swiglu_bwd: bad shapes")
    prof = GEMM_PROFILER
    if prof is not None:
        prof.begin()
    check(_L().ie_gemm_swiglu_bwd(_p(dy), dy.stride(0), _p(w2), w2.stride(0), _p(h13), h13.stride(0), _p(dh13), dh13.stride(0), _p(dact_scratch),
                                  dact_scratch.stride(0), M, F, K, _stream()), "ie_gemm_swiglu_bwd")
    if prof is not None:
        prof.end(2.0 * M * F * K, 2.0 * (M * K + F * K + M * F))
    return dh13


def gemm_batched(A, B, out, a_kmajor=False, b_kmajor=False, accumulate=False):
    """out[z] (+)= op(A[z]) @ op(B[z]) for the leading batch dimension z of three 3-D bf16 tensors (unit inner stride, equal batch strides
    per tensor): ONE launch for all products (the experts of a MoE layer)."""
    if not (A.dim() == B.dim() == out.dim() == 3 and A.shape[0] == B.shape[0] == out.shape[0]):
        raise ValueError("gemm_batched: three 3-D tensors with the same batch size expected")
    if A.stride(2) != 1 or B.stride(2) != 1 or out.stride(2) != 1:
        raise ValueError("gemm_batched: unit inner strides expected")
    Z = A.shape[0]
    K_, M = (A.shape[1], A.shape[2]) if a_kmajor else (A.shape[2], A.shape[1])
    Kb, N = (B.shape[1], B.shape[2]) if b_kmajor else (B.shape[2], B.shape[1])
    if K_ != Kb or out.shape[1:] != (M, N):
        raise ValueError(f"gemm_batched: shapes {tuple(A.shape)} x {tuple(B.shape)} -> {tuple(out.shape)}")
    if M == 0 or N == 0 or Z == 0:
        return out
    if K_ == 0:
        if not accumulate:
            out.zero_()
        return out
    prof = GEMM_PROFILER
    if prof is not None:
        prof.begin()
    check(_L().ie_gemm_bf16_batched(_p(A), A.stride(1), A.stride(0), int(a_kmajor), _p(B), B.stride(1), B.stride(0), int(b_kmajor), _p(out),
                                    out.stride(1), out.stride(0), M, N, K_, Z, int(accumulate), _stream()), "ie_gemm_bf16_batched")
    if prof is not None:
        prof.end(2.0 * Z * M * N * K_, 2.0 * Z * (M * K_ + N * K_ + M * N))
    return out


_ksplit_ws = {}   # device index -> the 32-MiB workspace of the weight-gradient tail k-split (kept for the life of the process: the library holds its address)


def enable_wgrad_ksplit(device, on=True):
    """Registers (or takes back) the workspace of the weight-gradient products' tail k-split (ie_gemm_set_wgrad_ksplit_workspace): a remainder of at most 128 tiles
    is then computed as two half-k products in one launch + a fixed-order fix-up (InternLM2-7B: the wqkv and w2 weight gradients, 384 and 896 tiles)."""
    if not on:
        check(_L().ie_gemm_set_wgrad_ksplit_workspace(None, 0), "ie_gemm_set_wgrad_ksplit_workspace")
        return
    idx = device.index if device.index is not None else torch.cuda.current_device()
    if idx not in _ksplit_ws:
        _ksplit_ws[idx] = torch.empty(32 << 20, dtype=torch.uint8, device=device)
    check(_L().ie_gemm_set_wgrad_ksplit_workspace(_ksplit_ws[idx].data_ptr(), _ksplit_ws[idx].numel()), "ie_gemm_set_wgrad_ksplit_workspace")


def hold_cus(blocks, usec):
    """(diagnostic) `blocks` idle workgroups, one CU each, for `usec` microseconds on the current stream (ie_hold_cus)."""
    check(_L().ie_hold_cus(int(blocks), int(usec), _stream()), "ie_hold_cus")


def bias_add(y, bias):
    """y [rows, cols] += bias [cols] in place (bf16): the linear biases of the InternLM-1 block (multi_head_attention.py:371-408)."""
    rows, cols, ld = _rows_ld(y)
    if bias.numel() != cols:
        raise ValueError("bias_add: size mismatch")
    check(_L().ie_bias_add_bf16(_p(y), ld, _p(bias), rows, cols, _stream()), "ie_bias_add_bf16")
    return y


def colsum(x, out=None):
    rows, cols, ld = _rows_ld(x)
    if out is None:
        out = torch.empty(cols, dtype=x.dtype, device=x.device)
    check(_L().ie_colsum_bf16(_p(x), ld, _p(out), rows, cols, _stream()), "ie_colsum_bf16")
    return out


# ------------------------------------------------------------------------------------------ flash attention
def _tok_stride(t, d):
    # t: [T, H, d] view with strides (ts, d, 1)
    if t.dim() != 3 or t.stride(2) != 1 or (t.shape[1] > 1 and t.stride(1) != d):
        raise ValueError("attention operand must be a [T, H, d] view with strides (ts, d, 1)")
    return t.stride(0)


def flash_attn_fwd(q, k, v, cu_seqlens, max_seqlen, softmax_scale=None, causal=True, out=None, lse=None):
    """q [T,hq,d], k/v [T,hkv,d] (views allowed) -> (out [T,hq,d], lse [hq,T] fp32)."""
    T, hq, d = q.shape
    hkv = k.shape[1]
    if softmax_scale is None:
        softmax_scale = 1.0 / math.sqrt(d)
    q_ts = _tok_stride(q, d)
    kv_ts = _tok_stride(k, d)
    if _tok_stride(v, d) != kv_ts:
        raise ValueError("k and v must share the token stride")
    if out is None:
        out = torch.empty((T, hq, d), dtype=q.dtype, device=q.device)
    if lse is None:
        lse = torch.empty((hq, T), dtype=torch.float32, device=q.device)
    if cu_seqlens.dtype != torch.int32:
        raise ValueError("cu_seqlens must be int32")
    nseq = cu_seqlens.numel() - 1
    check(_L().ie_flash_attn_fwd(_p(q), q_ts, _p(k), _p(v), kv_ts, _p(out), _tok_stride(out, d), _p(lse), _p(cu_seqlens), nseq, T,
                                 int(max_seqlen), hq, hkv, d, float(softmax_scale), int(bool(causal)), _stream()), "ie_flash_attn_fwd")
    return out, lse


def flash_attn_bwd(dout, q, k, v, out, lse, cu_seqlens, max_seqlen, softmax_scale=None, causal=True, dq=None, dk=None, dv=None,
                   delta_ws=None):
    T, hq, d = q.shape
    hkv = k.shape[1]
    if softmax_scale is None:
        softmax_scale = 1.0 / math.sqrt(d)
    if dq is None:
        dq = torch.empty((T, hq, d), dtype=q.dtype, device=q.device)
    if dk is None or dv is None:
        dkv = torch.empty((T, 2, hkv, d), dtype=q.dtype, device=q.device)
        dk, dv = dkv[:, 0], dkv[:, 1]
    need = _L().ie_flash_attn_bwd_workspace(T, hq, hkv, d)
    if delta_ws is None or delta_ws.numel() < need:
        delta_ws = torch.empty(need, dtype=torch.float32, device=q.device)
    kv_ts = _tok_stride(k, d)
    if _tok_stride(v, d) != kv_ts:
        raise ValueError("k and v must share the token stride")
    dkv_ts = _tok_stride(dk, d)
    if _tok_stride(dv, d) != dkv_ts:
        raise ValueError("dk and dv must share the token stride")
    nseq = cu_seqlens.numel() - 1
    check(_L().ie_flash_attn_bwd(_p(dout), _tok_stride(dout, d), _p(q), _tok_stride(q, d), _p(k), _p(v), kv_ts, _p(out),
                                 _tok_stride(out, d), _p(lse), _p(delta_ws), _p(dq), _tok_stride(dq, d), _p(dk), _p(dv), dkv_ts,
                                 _p(cu_seqlens), nseq, T, int(max_seqlen), hq, hkv, d, float(softmax_scale), int(bool(causal)),
                                 _stream()), "ie_flash_attn_bwd")
    return dq, dk, dv


def flash_attn_bwd_qkv_rotary(dout, q, k, v, out, lse, cu_seqlens, max_seqlen, cos, sin, positions, dqkv, softmax_scale=None, delta_ws=None):
    """The causal attention backward with the rotary embedding's backward and the GQA rearrange's in its stores (ie_flash_attn_bwd_qkv_rotary): dQ, dK (rotated
    back) and dV go straight into dqkv [T, hkv * (hq / hkv + 2) * d], the wqkv product's output gradient.  Returns False -- and touches nothing -- where the
    library does not fuse the shape (the caller then runs flash_attn_bwd + qkv_rotary_bwd: the same bits)."""
    T, hq, d = q.shape
    hkv = k.shape[1]
    nseq = cu_seqlens.numel() - 1
    L = _L()
    if not L.ie_flash_attn_bwd_qkv_rotary_is_fused(nseq, int(max_seqlen), hq, hkv, d, 1):
        return False
    if softmax_scale is None:
        softmax_scale = 1.0 / math.sqrt(d)
    if dqkv.numel() != T * hkv * (hq // hkv + 2) * d or not dqkv.is_contiguous() or positions.dtype != torch.int64 or positions.numel() != T:
        raise ValueError("flash_attn_bwd_qkv_rotary: dqkv must be a contiguous [T, hkv * (hq / hkv + 2) * d] tensor, positions int64[T]")
    need = L.ie_flash_attn_bwd_workspace(T, hq, hkv, d)
    if delta_ws is None or delta_ws.numel() < need:
        delta_ws = torch.empty(need, dtype=torch.float32, device=q.device)
    kv_ts = _tok_stride(k, d)
    if _tok_stride(v, d) != kv_ts:
        raise ValueError("k and v must share the token stride")
    check(L.ie_flash_attn_bwd_qkv_rotary(_p(dout), _tok_stride(dout, d), _p(q), _tok_stride(q, d), _p(k), _p(v), kv_ts, _p(out), _tok_stride(out, d), _p(lse),
                                         _p(delta_ws), _p(dqkv), _p(cos), _p(sin), _p(positions), _p(cu_seqlens), nseq, T, int(max_seqlen), hq, hkv, d,
                                         float(softmax_scale), 1, _stream()), "ie_flash_attn_bwd_qkv_rotary")
    return True


def flash_attn_fwd_x(q, k, v, cu_q, cu_k, max_seqlen_q, softmax_scale=None, out=None, lse=None):
    """Full attention of a rectangle of scores per sequence (ie_flash_attn_fwd_x): q [Tq, hq, d] rows cu_q[s] .. cu_q[s+1] against k / v [Tk, hkv, d] rows
    cu_k[s] .. cu_k[s+1] -> (out [Tq, hq, d], lse [hq, Tq] fp32; a sequence without keys: 0 and -inf)."""
    Tq, hq, d = q.shape
    Tk, hkv = k.shape[0], k.shape[1]
    if softmax_scale is None:
        softmax_scale = 1.0 / math.sqrt(d)
    kv_ts = _tok_stride(k, d)
    if _tok_stride(v, d) != kv_ts or v.shape[0] != Tk:
        raise ValueError("k and v must share the token stride and the row count")
    if out is None:
        out = torch.empty((Tq, hq, d), dtype=q.dtype, device=q.device)
    if lse is None:
        lse = torch.empty((hq, Tq), dtype=torch.float32, device=q.device)
    if cu_q.dtype != torch.int32 or cu_k.dtype != torch.int32 or cu_q.numel() != cu_k.numel():
        raise ValueError("cu_q / cu_k must be int32 and of one length")
    check(_L().ie_flash_attn_fwd_x(_p(q), _tok_stride(q, d), _p(k), _p(v), kv_ts, _p(out), _tok_stride(out, d), _p(lse), _p(cu_q), _p(cu_k),
                                   cu_q.numel() - 1, Tq, Tk, int(max_seqlen_q), hq, hkv, d, float(softmax_scale), _stream()), "ie_flash_attn_fwd_x")
    return out, lse


def flash_attn_bwd_x(dout, q, k, v, out, lse, cu_q, cu_k, max_seqlen_q, max_seqlen_k, softmax_scale=None, dq=None, dk=None, dv=None, delta_ws=None):
    """Backward of flash_attn_fwd_x with the lse / out the probabilities are normalised with (one block of ring attention: the merged ones) -> this
    block's (dq [Tq, hq, d], dk, dv [Tk, hkv, d]); overwritten."""
    Tq, hq, d = q.shape
    Tk, hkv = k.shape[0], k.shape[1]
    if Tq == 0:
        raise ValueError("flash_attn_bwd_x: no query rows (the caller zeroes dk / dv)")
    if softmax_scale is None:
        softmax_scale = 1.0 / math.sqrt(d)
    if dq is None:
        dq = torch.empty((Tq, hq, d), dtype=q.dtype, device=q.device)
    if dk is None or dv is None:
        dkv = torch.empty((Tk, 2, hkv, d), dtype=q.dtype, device=q.device)
        dk, dv = dkv[:, 0], dkv[:, 1]
    need = _L().ie_flash_attn_bwd_workspace(Tq, hq, hkv, d)
    if delta_ws is None or delta_ws.numel() < need:
        delta_ws = torch.empty(need, dtype=torch.float32, device=q.device)
    kv_ts, dkv_ts = _tok_stride(k, d), _tok_stride(dk, d)
    if _tok_stride(v, d) != kv_ts or _tok_stride(dv, d) != dkv_ts:
        raise ValueError("k / v and dk / dv must share their token strides")
    if tuple(lse.shape) != (hq, Tq) or not lse.is_contiguous():
        raise ValueError("lse must be a contiguous [hq, Tq] tensor")
    check(_L().ie_flash_attn_bwd_x(_p(dout), _tok_stride(dout, d), _p(q), _tok_stride(q, d), _p(k), _p(v), kv_ts, _p(out), _tok_stride(out, d), _p(lse),
                                   _p(delta_ws), _p(dq), _tok_stride(dq, d), _p(dk), _p(dv), dkv_ts, _p(cu_q), _p(cu_k), cu_q.numel() - 1, Tq, Tk,
                                   int(max_seqlen_q), int(max_seqlen_k), hq, hkv, d, float(softmax_scale), _stream()), "ie_flash_attn_bwd_x")
    return dq, dk, dv


def attn_merge(acc, lse_acc, out_p, lse_p, n):
    """Fold a block's partial (out_p bf16 [>= n, hq, d], lse_p [hq, Tp]) into the running fp32 result of rows 0 .. n-1 (acc [Ta, hq, d], lse_acc [hq, Ta])."""
    Ta, hq, d = acc.shape
    if not (acc.is_contiguous() and lse_acc.is_contiguous() and lse_p.is_contiguous() and acc.dtype == torch.float32):
        raise ValueError("attn_merge: contiguous fp32 accumulators expected")
    check(_L().ie_attn_merge(_p(acc), _p(lse_acc), Ta, _p(out_p), _tok_stride(out_p, d), _p(lse_p), lse_p.shape[1], int(n), hq, d, _stream()), "ie_attn_merge")


def acc_bf16(dst, src):
    """dst (fp32, contiguous) += src (bf16, contiguous, same element count)"""
    if dst.numel() != src.numel() or not (dst.is_contiguous() and src.is_contiguous()) or dst.dtype != torch.float32 or src.dtype != torch.bfloat16:
        raise ValueError("acc_bf16: contiguous fp32 += bf16 of one size")
    check(_L().ie_acc_bf16(_p(dst), _p(src), dst.numel(), _stream()), "ie_acc_bf16")


_spill_buf = None


def flash_attn_bwd_spill(enable, nseq=0, max_seqlen=0, hq=0, causal=True, device=None, four_waves=True):
    """Opt-in five-product attention backward (include/internevo_hip.h: ie_flash_attn_bwd_set_spill): hands the library a spill buffer large enough
    for calls of this shape (kept alive here) and selects the variant; ``enable=False`` takes both back.  Calls whose shape needs more than the
    buffer holds run the default path."""
    global _spill_buf
    L = _L()
    if not enable:
        check(L.ie_tune_flash_bwd_variant(0), "ie_tune_flash_bwd_variant")
        check(L.ie_flash_attn_bwd_set_spill(None, 0), "ie_flash_attn_bwd_set_spill")
        _spill_buf = None
        return 0
    need = L.ie_flash_attn_bwd_spill_bytes(int(nseq), int(max_seqlen), int(hq), int(bool(causal)))
    if need < 0:
        raise ValueError("flash_attn_bwd_spill: bad shape")
    if _spill_buf is None or _spill_buf.numel() < need or _spill_buf.device != torch.device(device or "cuda"):
        check(L.ie_flash_attn_bwd_set_spill(None, 0), "ie_flash_attn_bwd_set_spill")
        _spill_buf = torch.empty(need + 1024, dtype=torch.uint8, device=device or "cuda")
    off = (-_spill_buf.data_ptr()) % 1024
    check(L.ie_flash_attn_bwd_set_spill(_spill_buf.data_ptr() + off, need), "ie_flash_attn_bwd_set_spill")
    check(L.ie_tune_flash_bwd_variant(3 if four_waves else 2), "ie_tune_flash_bwd_variant")
    return need


def mfma_probe(a, b):
    c = torch.empty((32, 32), dtype=torch.float32, device=a.device)
    check(_L().ie_mfma_probe(_p(a), _p(b), _p(c), _stream()), "ie_mfma_probe")
    return c
