"""apex 23.05 call surface used by InternEvo (K5 fused RMSNorm, K6 multi-tensor L2 norm) on the gfx950 kernels."""
import numbers

import torch
import torch.nn as nn

from .. import kernels as K


class _RMSNormFunc(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, weight, eps):
        x = x.contiguous()
        y, rstd = K.rmsnorm_fwd(x, weight, eps)
        ctx.save_for_backward(x, weight, rstd)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, weight, rstd = ctx.saved_tensors
        dx, dw = K.rmsnorm_bwd(dy.contiguous(), x, weight, rstd)
        return dx, dw, None


class MixedFusedRMSNorm(nn.Module):
    """apex.normalization.fused_layer_norm.MixedFusedRMSNorm(normalized_shape, eps) as constructed through
    internlm/model/utils.py:662-675 (`RMSNorm(hidden_size, eps=layer_norm_epsilon)`): fp32 statistics, output in
    the weight's dtype, input bf16 or fp32 (modeling_internlm2.py:700,725,1002)."""

    def __init__(self, normalized_shape, eps=1e-5, elementwise_affine=True, **kwargs):
        super().__init__()
        if isinstance(normalized_shape, numbers.Integral):
            normalized_shape = (normalized_shape,)
        if len(normalized_shape) != 1 or not elementwise_affine:
            raise NotImplementedError("MixedFusedRMSNorm shim: 1-D normalized_shape with affine weight only")
        self.normalized_shape = torch.Size(normalized_shape)
        self.eps = eps
        self.elementwise_affine = True
        self.weight = nn.Parameter(torch.ones(*normalized_shape))

    def reset_parameters(self):
        nn.init.ones_(self.weight)

    def forward(self, x):
        return _RMSNormFunc.apply(x, self.weight, self.eps)

    def extra_repr(self):
        return f"{tuple(self.normalized_shape)}, eps={self.eps}"


def multi_tensor_l2norm(chunk_size, noop_flag, tensor_lists, per_tensor=False):
    """amp_C.multi_tensor_l2norm: returns (norm[1] fp32, per-tensor norms) for tensor_lists[0]
    (internlm/solver/optimizer/utils.py:191-204 passes one list of fp32-cast grads and per_tensor=False)."""
    tensors = tensor_lists[0]
    if len(tensors) == 0:
        dev = noop_flag.device
        return torch.zeros(1, dtype=torch.float32, device=dev), torch.zeros(0, dtype=torch.float32, device=dev)
    total = K.sumsq(tensors)
    if per_tensor:
        per = torch.cat([K.sumsq(t) for t in tensors]).sqrt_()
    else:
        per = torch.zeros(0, dtype=torch.float32, device=total.device)
    return total.sqrt_(), per


class _Applier:
    """apex.multi_tensor_apply.multi_tensor_applier(op, noop_flag_buffer, tensor_lists, *args)."""

    available = True
    chunk_size = 2048 * 32

    def __call__(self, op, noop_flag_buffer, tensor_lists, *args):
        return op(self.chunk_size, noop_flag_buffer, tensor_lists, *args)


multi_tensor_applier = _Applier()
