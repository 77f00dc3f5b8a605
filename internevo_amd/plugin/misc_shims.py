"""rotary_emb / fused_dense_lib / torch_scatter call surfaces used by InternEvo, on the gfx950 kernels."""
import torch

from .. import kernels as K


def apply_rotary(x1, x2, cos, sin, out1, out2, conj):
    """rotary_emb.apply_rotary(x1, x2, cos, sin, out1, out2, conj) -> None (writes out1/out2; they may alias x1/x2).
    Called at internlm/model/modules/embedding.py:115-120,142-153,207-224,239-256 with x* = the two halves of the
    rotary slice ([b, s, h, d/2] or [total, h, d/2] views) and cos/sin = [s, 1, d/2]."""
    K.apply_rotary(x1, x2, cos, sin, out1, out2, conj)


def linear_bias_wgrad(x, dy, has_bias):
    """fused_dense_lib.linear_bias_wgrad(x [M, K], dy [M, N], has_bias) -> (dW [N, K], db [N] | None)
    (internlm/model/utils.py:293-299: `total_x.reshape(batch_dim, -1), grad_output, ctx.needs_input_grad[2]`)."""
    x = x.contiguous() if x.stride(-1) != 1 else x
    dy = dy.contiguous() if dy.stride(-1) != 1 else dy
    dw = K.linear_wgrad(dy, x)
    db = K.colsum(dy) if has_bias else None
    return dw, db


def scatter(src, index, dim=-1, out=None, dim_size=None, reduce="sum"):
    """torch_scatter.scatter as used by internlm/model/metrics.py:93-96,276-279 (1-D sum over <= a few dataset types).
    Index bookkeeping of the metric pass, not part of the loss/grad path."""
    if reduce != "sum" or src.dim() != 1:
        raise NotImplementedError("scatter shim: 1-D sum only")
    n = int(dim_size) if dim_size is not None else (int(index.max()) + 1 if index.numel() else 0)
    res = out if out is not None else torch.zeros(n, dtype=src.dtype, device=src.device)
    return res.scatter_add_(0, index, src)
