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
import os

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
                 ep_size=1, ep_rank=0, tpar=None, a2a_chunks=None, a2a_overlap=True, expert_fp8=False):
        """tpar (tensorpar.TensorParallel, tp > 1): every expert is a FeedForward over the TENSOR group (gshard_layer.py:421-433 -> modules/mlp.py:40-86):
        `ffn` is then this rank's F / tp units (w1 / w3 cut by rows, w2 by columns), the experts' outputs are partial sums that are all-reduced over the
        group before the combine (RowParallelLinearTorch), and so is the gradient of the dispatched tokens behind the w1 | w3 products
        (ColumnParallelLinearTorch's backward); gate, routing, dispatch and combine run replicated on the same tokens with the same noise.
        a2a_chunks (expert parallelism only; default 2 where the capacity divides, IE_MOE_A2A_CHUNKS overrides): the exchange of the expert buffers runs in that
        many pieces along the capacity, piece k + 1's all_to_all under piece k's expert products and piece k's way back under piece k + 1's -- the reference's
        dispatch / combine exchanges are blocking (gshard_layer.py:465-498, moe/utils.py:8-63).  The buffers are then kept chunk-major (ie_moe_chunk_rows).
        a2a_overlap=False: the same pieces, every exchange waited for at once (what the overlapped form must equal bit for bit).  The pieces' rows are the
        same rows either way: outputs and input gradients do not depend on the chunk count; the weight gradients add their pieces up in bf16 like the blocks
        of the source ranks do, so they depend on it in the last bf16 bit.
        tokens: tokens per forward call (one micro-batch: the reference gates per call).  Parameters are NOT owned here: forward /
        backward take views (the engine keeps them in its flat buffers): wg fp32 [E, M]; w13 bf16 [E_local, 2F, M]; w2 bf16 [E_local, M, F].
        expert_fp8 (OPT-IN, BASELINE configs[4] "fp8 MFMA linear layers"; the reference has no fp8 linear, SURVEY.md section 8f rank 2, so the tolerance is this
        repo's own, tests/test_fp8_gpu.py): the two FORWARD products of every expert run on OCP e4m3 operands (ie_gemm_fp8_batched on v_mfma_f32_32x32x64_f8f6f4:
        per-tensor dynamic scales per expert block and per expert weight, fp32 accumulation, bf16 results); the backward keeps the bf16 weights and the saved bf16
        activations (straight-through).  Needs hidden % 128 == 0 and ffn % 128 == 0.  An expert's quantised weights are kept until invalidate_fp8() (the engine
        calls it after every optimizer step).  Not the default: the separate amax / convert passes cost the MoE step what the products save
        (profiles/r04_fp8_expert_gemm.md: 224 vs 217 ms on 8 layers) -- round 5 had deleted the route for that, round 6 restored it as the opt-in it is."""
        if expert_fp8 and (hidden % 128 or ffn % 128):
            raise ValueError(f"expert_fp8: hidden ({hidden}) and ffn ({ffn}) must be multiples of 128 (one LDS row of e4m3 values)")
        self.fp8 = bool(expert_fp8)
        self._wq, self._qa = {}, {}   # quantised weights (until invalidate_fp8) / the quantised-activation buffers
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
        nch = a2a_chunks if a2a_chunks is not None else int(os.environ.get("IE_MOE_A2A_CHUNKS", "2"))
        if ep_size <= 1 or nch < 1 or self.C % nch or (self.C // nch) % 8:
            nch = 1   # (no exchange to hide, or the capacity does not cut into whole 8-row pieces)
        self.nch, self.overlap = nch, bool(a2a_overlap)
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
        self.token_of_raw = torch.empty(E * C, **i32) if self.nch > 1 else self.token_of   # (expert-major, as ie_moe_route writes it)
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
        """One piece of the expert exchange, started: -> the Work to wait for (waited for here unless the exchanges are overlapped)."""
        from .comm import backend_for

        if getattr(self, "_be", None) is None:
            self._be = backend_for(self.ep_group)
        w = self._be.all_to_all(recv, send, self.ep_group)
        if not self.overlap:
            w.wait()
        return w

    def _pieces(self):
        """Row ranges of the exchange pieces in the chunk-major expert buffers: piece k = rows [k E Cn, (k + 1) E Cn) = [ep][El][Cn] on the sending side,
        [source rank][El][Cn] on the receiving side."""
        n = self.E * (self.C // self.nch)
        return [slice(k * n, (k + 1) * n) for k in range(self.nch)]

    def _expert_blocks(self, piece):
        """(rows, experts-batch shape) of the GEMM operands inside one piece: with ONE local expert all source ranks' rows are one operand of ep Cn rows;
        with several, one strided batch over the local experts per source rank (as the unchunked layout had per source rank)."""
        Cn, El, ep = self.C // self.nch, self.El, self.ep
        if El == 1:
            return [(piece, (1, ep * Cn))]
        return [(slice(piece.start + g * El * Cn, piece.start + (g + 1) * El * Cn), (El, Cn)) for g in range(ep)]

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
        check(L.ie_moe_route(K._p(self.gates), K._p(self.expert), S, E, C, K._p(self.row), K._p(self.weight), K._p(self.token_of_raw), K._p(self.l_aux),
                             K._p(self.exp_counts), st()), "ie_moe_route")
        if self.nch > 1:   # the expert buffers chunk-major from here on: row / token_of are only ever used as indices into them
            check(L.ie_moe_chunk_rows(K._p(self.row), K._p(self.token_of_raw), K._p(self.token_of), S, E, C, self.nch, st()), "ie_moe_chunk_rows")
        check(L.ie_moe_dispatch(K._p(x), x.stride(0), K._p(self.token_of), E * C, M, K._p(self.ein), st()), "ie_moe_dispatch")
        if self.ep == 1:
            self._experts_fwd(self.ein, self.eo, slice(0, E * C), [(slice(0, E * C), (self.El, C))], w13, w2)
        else:
            # piece k + 1 is on its way while the experts work on piece k, piece k returns while they work on piece k + 1
            pieces = self._pieces()
            there = [self._a2a(self.ein[p], self.xin[p]) for p in pieces]
            back = []
            for p, w in zip(pieces, there):
                w.wait()
                self._experts_fwd(self.xin, self.xout, p, self._expert_blocks(p), w13, w2)
                back.append(self._a2a(self.xout[p], self.eo[p]))
            for w in back:
                w.wait()
        check(L.ie_moe_combine_fwd(K._p(self.eo), K._p(self.row), K._p(self.weight), S, M, K._p(out), out.stride(0), st()), "ie_moe_combine_fwd")
        return self.l_aux

    def _experts_fwd(self, ein, eo, rows, blocks, w13, w2):
        """The local experts on the rows `rows` of their buffers: w1 | w3 products, the gate, w2 products (+ the tensor group's sum of the partial outputs)."""
        M, F = self.M, self.F
        for r, (nb, nr) in blocks:
            self._products(ein[r].view(nb, nr, M), w13, self.h13[r].view(nb, nr, 2 * F), "w13")
        K.swiglu_fwd(self.h13[rows, :F], self.h13[rows, F:], self.act[rows])
        for r, (nb, nr) in blocks:
            self._products(self.act[r].view(nb, nr, F), w2, eo[r].view(nb, nr, M), "w2")
        if self.tpar is not None:
            self.tpar.all_reduce_sum(eo[rows])          # w2 is row-parallel: every tensor rank holds a partial sum of the experts' outputs

    def _experts_bwd(self, d_eo, ein, d_ein, rows, blocks, w13, w2, d_w13, d_w2, accumulate):
        """Backward of _experts_fwd on the same rows: input gradients of w2, its weight gradient (the blocks of one expert add up), the gate's backward,
        input gradients of w1 | w3 (summed over the tensor group under the weight gradient), their weight gradient."""
        M, F = self.M, self.F
        for r, (nb, nr) in blocks:      # dgrad of w2: d_act[e] = d_eo[e] @ w2[e]
            K.gemm_batched(d_eo[r].view(nb, nr, M), w2, self.d_act[r].view(nb, nr, F), b_kmajor=True)
        for g, (r, (nb, nr)) in enumerate(blocks):
            K.gemm_batched(d_eo[r].view(nb, nr, M), self.act[r].view(nb, nr, F), d_w2, a_kmajor=True, b_kmajor=True, accumulate=accumulate or g > 0)
        K.swiglu_bwd(self.d_act[rows], self.h13[rows, :F], self.h13[rows, F:], self.d_h13[rows, :F], self.d_h13[rows, F:])
        for r, (nb, nr) in blocks:
            K.gemm_batched(self.d_h13[r].view(nb, nr, 2 * F), w13, d_ein[r].view(nb, nr, M), b_kmajor=True)
        h_ = self.tpar.all_reduce_sum_async(d_ein[rows]) if self.tpar is not None else None   # (w1 | w3 are column-parallel)
        for g, (r, (nb, nr)) in enumerate(blocks):
            K.gemm_batched(self.d_h13[r].view(nb, nr, 2 * F), ein[r].view(nb, nr, M), d_w13, a_kmajor=True, b_kmajor=True, accumulate=accumulate or g > 0)
        if h_ is not None:
            h_.wait()

    def _products(self, a, w, out, which):
        """out[j] = a[j] @ w[j]^T for the local experts j in ONE strided-batched launch: bf16, or (expert_fp8) the e4m3 products of the blocks quantised now (a
        scale per expert block) and the weights quantised since the last invalidate_fp8() (a scale per expert)."""
        if not self.fp8:
            K.gemm_batched(a, w, out)
            return
        nb = a.shape[0]
        key = (which, w.data_ptr())
        if key not in self._wq:
            self._wq[key] = K.fp8_quantize(w, per_slice=True)
        buf = self._qa.get((which, tuple(a.shape)))
        if buf is None:
            f32 = dict(dtype=torch.float32, device=a.device)
            buf = self._qa[(which, tuple(a.shape))] = (torch.empty(a.shape, dtype=torch.uint8, device=a.device), torch.empty(nb, **f32), torch.empty(nb, **f32))
        qa, da = K.fp8_quantize(a.contiguous(), per_slice=True, out=buf)
        K.gemm_fp8_batched(qa, da, *self._wq[key], out)

    def invalidate_fp8(self):
        """The expert weights have changed (optimizer step, checkpoint load): quantise them again at the next forward."""
        self._wq = {}

    def backward(self, dout, wg, w13, w2, dx, d_wg, d_w13, d_w2, accumulate, loss_scale_dev=None, aux_factor=0.0):
        """dout bf16 [S, M] -> dx bf16 [S, M] (overwritten).  d_wg fp32 [E, M], d_w13 / d_w2 bf16 like the weights: written, or added to
        when `accumulate`.  aux_factor: d(loss) / d(l_aux) up to the loss scale read from loss_scale_dev (device float, or None = 1)."""
        S, E, C, M, F = self.S, self.E, self.C, self.M, self.F
        L, st = K._L(), K._stream
        check(L.ie_moe_combine_bwd(K._p(dout), dout.stride(0), K._p(self.eo), K._p(self.token_of), K._p(self.weight), E * C, S, M, K._p(self.d_eo),
                                   K._p(self.d_weight), st()), "ie_moe_combine_bwd")
        if self.ep == 1:
            self._experts_bwd(self.d_eo, self.ein, self.d_ein, slice(0, E * C), [(slice(0, E * C), (self.El, C))], w13, w2, d_w13, d_w2, accumulate)
        else:
            pieces = self._pieces()
            there = [self._a2a(self.d_eo[p], self.d_xout[p]) for p in pieces]
            back = []
            for k, (p, w) in enumerate(zip(pieces, there)):
                w.wait()
                self._experts_bwd(self.d_xout, self.xin, self.d_xin, p, self._expert_blocks(p), w13, w2, d_w13, d_w2, accumulate or k > 0)
                back.append(self._a2a(self.d_xin[p], self.d_ein[p]))
            for w in back:
                w.wait()
        check(L.ie_moe_dispatch_bwd(K._p(self.d_ein), K._p(self.row), K._p(self.token_of), S, M, K._p(dx), dx.stride(0), st()), "ie_moe_dispatch_bwd")
        check(L.ie_moe_gate_bwd(K._p(self.x), self.x.stride(0), K._p(wg), K._p(self.gates), K._p(self.expert), K._p(self.row), K._p(self.d_weight),
                                K._p(self.exp_counts), K._p(loss_scale_dev), float(aux_factor), S, M, E, K._p(self.d_logits), K._p(dx), dx.stride(0),
                                K._p(d_wg), 1 if accumulate else 0, K._p(self.dwg_ws), st()), "ie_moe_gate_bwd")
        return dx
