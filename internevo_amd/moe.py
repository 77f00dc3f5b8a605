"""GShard mixture-of-experts feed-forward layer on the HIP kernels of csrc/moe.hip (BASELINE configs[4]).

Host-side mirror of the reference's `MoE` / `GShardMOELayer` (internlm/model/moe/moe.py:13-100, moe/gshard_layer.py:360-498,
configs/7B_MoE4_sft.py `moe = dict(top_k=2, capacity_factor, min_capacity, ...)`): top-2 gating by an fp32 gate, capacity drop, E SwiGLU
experts, weighted combine, auxiliary load-balancing loss.  MI355X-first differences that do not change the numbers:
  * index form: a token's two choices are two row indices into [E * C, M] expert buffers; dispatch is a gather, combine a 2-term
    weighted sum -- the reference builds [S, E, C] one-hot tensors and runs O(S E C M) einsums over them;
  * no autograd: `forward` keeps what `backward` needs in pre-allocated buffers, `backward` is explicit;
  * the experts of one layer run as ONE strided-batched GEMM per product over contiguous row ranges of the same buffers
    (ie_gemm_bf16_batched; w1 | w3 fused into one [2F, M] operand per expert, as the dense FFN of the engine): four experts of
    2048 rows each fill the 256 CUs together where four separate launches left half-empty rounds.
Expert parallelism (`parallel.expert`, all_to_all of the expert buffers over xGMI, gshard_layer.py:453-474): `ep_group` splits the E
experts over the ranks of the group; the [E, C, M] buffer is exchanged by ONE all_to_all_single each way (rank r keeps the C-row
blocks of its E/ep experts from every rank), mirrored in backward.

The Gumbel noise of the second choice is generated on the device from (seed, layer, call counter) unless the caller supplies it
(the parity tests inject the oracle's noise so that HIP, oracle and the real reference route identically).
"""
import numpy as np
import torch

from . import kernels as K
from ._lib import check

BF16 = torch.bfloat16


def capacity(num_tokens, num_experts, capacity_factor, min_capacity, top_k=2):
    """gshard_layer.py:113-122 with top2gating's doubled factor (:222): a float32 product, ceil, clamped from below."""
    c = int(np.ceil(np.float32(num_tokens / num_experts) * np.float32(capacity_factor * top_k)))
    return max(c, int(min_capacity))


class MoELayer:
    def __init__(self, hidden, ffn, num_experts, tokens, device, capacity_factor=1.0, min_capacity=4, seed=0, layer_index=0, ep_group=None,
                 ep_size=1, ep_rank=0, tpar=None):
        """tpar (tensorpar.TensorParallel, tp > 1): every expert is a FeedForward over the TENSOR group (gshard_layer.py:421-433 -> modules/mlp.py:40-86):
        `ffn` is then this rank's F / tp units (w1 / w3 cut by rows, w2 by columns), the experts' outputs are partial sums that are all-reduced over the
        group before the combine (RowParallelLinearTorch), and so is the gradient of the dispatched tokens behind the w1 | w3 products
        (ColumnParallelLinearTorch's backward); gate, routing, dispatch and combine run replicated on the same tokens with the same noise.
        tokens: tokens per forward call (one micro-batch: the reference gates per call).  Parameters are NOT owned here: forward /
        backward take views (the engine keeps them in its flat buffers): wg fp32 [E, M]; w13 bf16 [E_local, 2F, M]; w2 bf16 [E_local, M, F].
        (The fp8 expert route of round 4 was removed in round 5: it lost 3 % in the step; the e4m3 product itself stays in the library, kernels.gemm_fp8.)"""
        if not 2 <= num_experts <= 16:
            raise NotImplementedError("2 <= num_experts <= 16")
        if num_experts % ep_size:
            raise ValueError(f"Number of experts ({num_experts}) should be divisible by expert parallel size ({ep_size})")  # gshard_layer.py:404
        if tokens < min_capacity:
            raise ValueError("No. of tokens (batch-size) should be greater than min_capacity.")
        self.M, self.F, self.E, self.S = hidden, ffn, num_experts, tokens
        self.cf, self.min_cap = capacity_factor, min_capacity
        self.C = capacity(tokens, num_experts, capacity_factor, min_capacity)
        self.ep_group, self.ep, self.ep_rank = ep_group, ep_size, ep_rank
        self.tpar = tpar if tpar is not None and tpar.tp > 1 else None
        self.El = num_experts // ep_size
        self.seed, self.layer, self.calls = int(seed), int(layer_index), 0
        self.dev = device
        E, S, C, M, F = self.E, self.S, self.C, hidden, ffn
        f32 = dict(dtype=torch.float32, device=device)
        i32 = dict(dtype=torch.int32, device=device)
        self.noise = torch.empty(S, E, **f32)
        self.logits, self.gates = torch.empty(S, E, **f32), torch.empty(S, E, **f32)
        self.expert, self.row = torch.empty(2, S, **i32), torch.empty(2, S, **i32)
        self.weight, self.d_weight = torch.empty(2, S, **f32), torch.empty(2, S, **f32)
        self.token_of = torch.empty(E * C, **i32)
        self.l_aux = torch.empty(1, **f32)
        self.exp_counts = torch.empty(E, **i32)
        self.d_logits = torch.empty(S, E, **f32)
        self.dwg_ws = torch.empty(K._L().ie_moe_dwg_workspace(M, E), **f32)
        # expert buffers: rows [e*C, (e+1)*C) belong to expert e.  Under expert parallelism the local experts see ep * C rows each
        R = self.El * self.ep * C
        self.ein = torch.empty(E * C, M, dtype=BF16, device=device)          # what this rank's tokens send to every expert
        self.eo = torch.empty(E * C, M, dtype=BF16, device=device)           # what comes back
        self.d_eo = torch.empty(E * C, M, dtype=BF16, device=device)
        self.d_ein = torch.empty(E * C, M, dtype=BF16, device=device)
        if self.ep > 1:
            self.xin, self.xout = torch.empty(R, M, dtype=BF16, device=device), torch.empty(R, M, dtype=BF16, device=device)
            self.d_xin, self.d_xout = torch.empty(R, M, dtype=BF16, device=device), torch.empty(R, M, dtype=BF16, device=device)
        self.rows_local = R
        self.h13 = torch.empty(R, 2 * F, dtype=BF16, device=device)
        self.act = torch.empty(R, F, dtype=BF16, device=device)
        self.d_act = torch.empty(R, F, dtype=BF16, device=device)
        self.d_h13 = torch.empty(R, 2 * F, dtype=BF16, device=device)

    # ---- expert parallel exchange: [ep][El*C rows] send blocks <-> [ep][El*C] received (rank-major); the local experts then see,
    # for local expert j, the rows {g*El*C + j*C .. +C} of every source rank g -- processed as ep separate [C, M] GEMM operands
    def _a2a(self, send, recv):
        from .comm import backend_for

        if getattr(self, "_be", None) is None:
            self._be = backend_for(self.ep_group)
        self._be.all_to_all(recv, send, self.ep_group).wait()
        return recv

    def forward(self, x, wg, w13, w2, out, noise=None):
        """x bf16 [S, M] -> out bf16 [S, M]; returns the device scalar l_aux (bf16-rounded fp32).  noise: fp32 [S, E] to inject."""
        S, E, C, M, F = self.S, self.E, self.C, self.M, self.F
        L, st = K._L(), K._stream
        if noise is None:
            check(L.ie_moe_gumbel_noise(K._p(self.noise), S * E, self.seed & 0xFFFFFFFF, (self.layer << 40) + self.calls * S * E, st()), "ie_moe_gumbel_noise")
            noise = self.noise
        self.calls += 1
        self.x = x
        check(L.ie_moe_gate_fwd(K._p(x), x.stride(0), K._p(wg), K._p(noise), S, M, E, K._p(self.logits), K._p(self.gates), K._p(self.expert), st()),
              "ie_moe_gate_fwd")
        check(L.ie_moe_route(K._p(self.gates), K._p(self.expert), S, E, C, K._p(self.row), K._p(self.weight), K._p(self.token_of), K._p(self.l_aux),
                             K._p(self.exp_counts), st()), "ie_moe_route")
        check(L.ie_moe_dispatch(K._p(x), x.stride(0), K._p(self.token_of), E * C, M, K._p(self.ein), st()), "ie_moe_dispatch")
        ein = self._a2a(self.ein, self.xin) if self.ep > 1 else self.ein
        eo = self.xout if self.ep > 1 else self.eo
        # the El local experts run as ONE strided-batched GEMM per product (and per source rank under expert parallelism): expert j's
        # [C, M] block x its own weights; the SwiGLU gate covers all rows at once
        El, ep = self.El, self.ep
        for g in range(ep):
            rows = slice(g * El * C, (g + 1) * El * C)
            self._products(ein[rows].view(El, C, M), w13, self.h13[rows].view(El, C, 2 * F), "w13")
        K.swiglu_fwd(self.h13[:, :F], self.h13[:, F:], self.act)
        for g in range(ep):
            rows = slice(g * El * C, (g + 1) * El * C)
            self._products(self.act[rows].view(El, C, F), w2, eo[rows].view(El, C, M), "w2")
        if self.tpar is not None:
            self.tpar.all_reduce_sum(eo)          # w2 is row-parallel: every tensor rank holds a partial sum of the experts' outputs
        if self.ep > 1:
            self._a2a(self.xout, self.eo)
        check(L.ie_moe_combine_fwd(K._p(self.eo), K._p(self.row), K._p(self.weight), S, M, K._p(out), out.stride(0), st()), "ie_moe_combine_fwd")
        return self.l_aux

    def _products(self, a, w, out, which):
        """out[j] = a[j] @ w[j]^T for the local experts j in ONE strided-batched launch."""
        K.gemm_batched(a, w, out)

    def backward(self, dout, wg, w13, w2, dx, d_wg, d_w13, d_w2, accumulate, loss_scale_dev=None, aux_factor=0.0):
        """dout bf16 [S, M] -> dx bf16 [S, M] (overwritten).  d_wg fp32 [E, M], d_w13 / d_w2 bf16 like the weights: written, or added to
        when `accumulate`.  aux_factor: d(loss) / d(l_aux) up to the loss scale read from loss_scale_dev (device float, or None = 1)."""
        S, E, C, M, F = self.S, self.E, self.C, self.M, self.F
        L, st = K._L(), K._stream
        check(L.ie_moe_combine_bwd(K._p(dout), dout.stride(0), K._p(self.eo), K._p(self.token_of), K._p(self.weight), E * C, S, M, K._p(self.d_eo),
                                   K._p(self.d_weight), st()), "ie_moe_combine_bwd")
        d_eo = self._a2a(self.d_eo, self.d_xout) if self.ep > 1 else self.d_eo
        ein = self.xin if self.ep > 1 else self.ein
        d_ein = self.d_xin if self.ep > 1 else self.d_ein
        El, ep = self.El, self.ep
        blocks = [slice(g * El * C, (g + 1) * El * C) for g in range(ep)]
        for rows in blocks:      # dgrad of w2: d_act[e] = d_eo[e] @ w2[e]
            K.gemm_batched(d_eo[rows].view(El, C, M), w2, self.d_act[rows].view(El, C, F), b_kmajor=True)
        for g, rows in enumerate(blocks):   # the blocks of one expert coming from different source ranks add up
            K.gemm_batched(d_eo[rows].view(El, C, M), self.act[rows].view(El, C, F), d_w2, a_kmajor=True, b_kmajor=True, accumulate=accumulate or g > 0)
        K.swiglu_bwd(self.d_act, self.h13[:, :F], self.h13[:, F:], self.d_h13[:, :F], self.d_h13[:, F:])
        for rows in blocks:
            K.gemm_batched(self.d_h13[rows].view(El, C, 2 * F), w13, d_ein[rows].view(El, C, M), b_kmajor=True)
        if self.tpar is not None:                 # w1 | w3 are column-parallel: the input gradient is summed over the tensor group, under the weight gradient
            h_ = self.tpar.all_reduce_sum_async(d_ein)
        for g, rows in enumerate(blocks):
            K.gemm_batched(self.d_h13[rows].view(El, C, 2 * F), ein[rows].view(El, C, M), d_w13, a_kmajor=True, b_kmajor=True, accumulate=accumulate or g > 0)
        if self.tpar is not None:
            h_.wait()
        if self.ep > 1:
            self._a2a(self.d_xin, self.d_ein)
        check(L.ie_moe_dispatch_bwd(K._p(self.d_ein), K._p(self.row), K._p(self.token_of), S, M, K._p(dx), dx.stride(0), st()), "ie_moe_dispatch_bwd")
        check(L.ie_moe_gate_bwd(K._p(self.x), self.x.stride(0), K._p(wg), K._p(self.gates), K._p(self.expert), K._p(self.row), K._p(self.d_weight),
                                K._p(self.exp_counts), K._p(loss_scale_dev), float(aux_factor), S, M, E, K._p(self.d_logits), K._p(dx), dx.stride(0),
                                K._p(d_wg), 1 if accumulate else 0, K._p(self.dwg_ws), st()), "ie_moe_gate_bwd")
        return dx
