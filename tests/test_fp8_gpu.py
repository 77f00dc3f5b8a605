"""fp8 (OCP e4m3) operands for the expert products (BASELINE configs[4] "fp8 MFMA linear layers"; include/internevo_hip.h:
ie_fp8_amax, ie_fp8_quantize, ie_gemm_fp8) and the OPT-IN product path on them: MoELayer / MoEEngine(expert_fp8=True) -- built in round 4, deleted in round 5
because it is slower in the step (profiles/r04_fp8_expert_gemm.md), restored in round 6 as the opt-in it is.
The reference has no fp8 linear, so there is no reference arithmetic to pin; what IS pinned:
  * the quantiser against torch's own float8_e4m3fn cast (same scale arithmetic, round-to-nearest-even, saturation): bit for bit;
  * the product against the fp32 product of the DEQUANTISED operands: only the bf16 rounding of the result and the summation order are left (8e-3);
  * and the tolerance this repo defines for the whole fp8 path against the bf16-operand product on the expert shapes' statistics: relative l2 error <= 5e-2
    (measured 2.6e-2 ... 3.7e-2 for unit-normal operands: two e4m3 roundings of 2^-4 relative half-ulp each)."""
import math

import pytest
import torch

pytestmark = pytest.mark.gpu
F8 = getattr(torch, "float8_e4m3fn", None)


def K():
    from internevo_amd import kernels

    return kernels


def g(seed):
    return torch.Generator().manual_seed(seed)


def _deq(q, dq):
    return q.view(F8).float() * dq.float()


@pytest.mark.skipif(F8 is None, reason="torch without float8_e4m3fn")
@pytest.mark.parametrize("shape", [(1024, 512), (1003,), (7,), (3, 5, 64)])
def test_quantiser_equals_torchs_e4m3_cast_bit_for_bit(dev, shape):
    x = (torch.randn(shape, generator=g(1)) * 3.0).to(torch.bfloat16)
    x.view(-1)[0] = 37.5    # an outlier sets the scale
    xd = x.to(dev)
    q, dq = K().fp8_quantize(xd)
    amax = xd.float().abs().max()
    scale = torch.tensor(448.0, device=dev) / amax
    want = (xd.float() * scale).clamp(-448.0, 448.0).to(F8).view(torch.uint8)
    assert q.shape == x.shape and q.dtype == torch.uint8
    assert torch.equal(q, want), f"{int((q != want).sum())} of {q.numel()} codes differ"
    assert float(dq) == float(amax / 448.0)
    assert float(_deq(q, dq).view(-1)[0]) == 37.5   # the largest magnitude is exactly representable after scaling
    z, dz = K().fp8_quantize(torch.zeros(64, dtype=torch.bfloat16, device=dev))
    assert int(z.max()) == 0 and float(dz) == 1.0


@pytest.mark.skipif(F8 is None, reason="torch without float8_e4m3fn")
@pytest.mark.parametrize("M,N,Kd", [(264, 256, 128), (520, 392, 256), (8, 520, 384), (304, 1000, 1024), (1024, 768, 4096)])
def test_fp8_product_equals_the_fp32_product_of_the_dequantised_operands(dev, M, N, Kd):
    """One, two, three and many k-tiles (a tile is 128 e4m3 values of k), ragged M / N edges, a strided A view, accumulate; a contraction length that is not a whole
    LDS row is refused, not mis-read."""
    Abig = (torch.randn(M, Kd + 128, generator=g(2))).to(torch.bfloat16).to(dev)
    B = (torch.randn(N, Kd, generator=g(3)) * 0.05).to(torch.bfloat16).to(dev)
    qa_big, da = K().fp8_quantize(Abig)
    qb, db = K().fp8_quantize(B)
    qa = qa_big[:, :Kd]                      # lda = Kd + 128
    ref = _deq(qa, da) @ _deq(qb, db).t()
    C = K().gemm_fp8(qa, da, qb, db)
    err = float((C.float() - ref).abs().max())
    bound = 8e-3 * float(ref.abs().max()) + 1e-6
    print(f"fp8 product {M}x{N}x{Kd}: max |C - ref| {err:.3e} (bound {bound:.3e})")
    assert torch.isfinite(C).all() and err <= bound
    C0 = torch.randn(M, N, generator=g(4)).to(torch.bfloat16).to(dev)
    Cd = C0.clone()
    K().gemm_fp8(qa, da, qb, db, out=Cd, accumulate=True)
    want = C0.float() + ref.to(torch.bfloat16).float()
    assert float((Cd.float() - want).abs().max()) <= 1.6e-2 * float(want.abs().max()) + 1e-6
    with pytest.raises(Exception):
        K().gemm_fp8(qa[:, :64], da, qb[:, :64], db)


@pytest.mark.skipif(F8 is None, reason="torch without float8_e4m3fn")
def test_batched_fp8_product_and_per_slice_scales_equal_the_single_products(dev):
    """The experts of a layer as ONE launch (ie_gemm_fp8_batched) with one scale per expert block and per expert weight (per_slice quantisation): bit for bit the
    Z single products on the slices quantised one by one."""
    Z, M, N, Kd = 4, 200, 264, 256
    A = torch.randn(Z, M, Kd, generator=g(20)).to(torch.bfloat16)
    A[2] *= 7.0                                  # every slice has its own scale
    B = (torch.randn(Z, N, Kd, generator=g(21)) * 0.05).to(torch.bfloat16)
    A, B = A.to(dev), B.to(dev)
    qa, da = K().fp8_quantize(A, per_slice=True)
    qb, db = K().fp8_quantize(B, per_slice=True)
    assert da.shape == (Z,) and float(da[2]) > 3.0 * float(da[0])
    out = torch.empty(Z, M, N, dtype=torch.bfloat16, device=dev)
    K().gemm_fp8_batched(qa, da, qb, db, out)
    for z in range(Z):
        q1, d1 = K().fp8_quantize(A[z])
        q2, d2 = K().fp8_quantize(B[z])
        assert torch.equal(q1, qa[z]) and float(d1) == float(da[z]) and torch.equal(q2, qb[z])
        assert torch.equal(out[z], K().gemm_fp8(q1, d1, q2, d2)), z


@pytest.mark.skipif(F8 is None, reason="torch without float8_e4m3fn")
def test_fp8_path_against_the_bf16_product_within_the_tolerance_this_repo_defines(dev):
    """The whole path -- dynamic per-tensor scales, two e4m3 roundings, fp32 accumulation -- against the bf16-operand product of the same tensors on an expert-shaped
    product (activations ~ N(0, 1) with a few large rows, weights ~ N(0, 0.02)): relative l2 error <= 5e-2."""
    M, N, Kd = 4096, 2048, 4096
    x = torch.randn(M, Kd, generator=g(5))
    x[::257] *= 8.0
    x = x.to(torch.bfloat16).to(dev)
    w = (torch.randn(N, Kd, generator=g(6)) * 0.02).to(torch.bfloat16).to(dev)
    qx, dx = K().fp8_quantize(x)
    qw, dw = K().fp8_quantize(w)
    C8 = K().gemm_fp8(qx, dx, qw, dw).float()
    C16 = K().gemm(x, w).float()
    rel = float((C8 - C16).norm() / C16.norm())
    print(f"fp8 vs bf16 operands, {M}x{N}x{Kd}: relative l2 error {rel:.3e}")
    assert rel <= 5e-2 and rel >= 1e-3   # (and it IS an fp8 product: the error is not bf16's)


def _rel(a, b):
    return float((a.float() - b.float()).norm() / b.float().norm())


@pytest.mark.skipif(F8 is None, reason="torch without float8_e4m3fn")
def test_moe_layer_with_fp8_expert_products_stays_within_the_tolerance_and_trains_the_bf16_backward(dev):
    """moe.MoELayer(expert_fp8=True): the two forward products of every expert on e4m3 operands, routing untouched (the gate is an fp32 module upstream of the
    experts), the backward on the bf16 weights and the saved activations.  Against the bf16 layer on the same inputs and noise: identical expert choices, output and
    every gradient within 8e-2 relative l2; quantised weights are reused until invalidate_fp8()."""
    from internevo_amd.moe import MoELayer

    S, M, F_, E = 1024, 256, 512, 4
    x = torch.randn(S, M, generator=g(7)).to(torch.bfloat16).to(dev)
    dy = (torch.randn(S, M, generator=g(8)) * 0.1).to(torch.bfloat16).to(dev)
    wg = (torch.randn(E, M, generator=g(9)) * 0.1).to(dev)
    w13 = (torch.randn(E, 2 * F_, M, generator=g(10)) * 0.05).to(torch.bfloat16).to(dev)
    w2 = (torch.randn(E, M, F_, generator=g(11)) * 0.05).to(torch.bfloat16).to(dev)
    noise = torch.randn(S, E, generator=g(12)).to(dev)
    res = {}
    for fp8 in (False, True):
        lay = MoELayer(M, F_, E, S, dev, 1.0, 4, expert_fp8=fp8)
        out, dx = torch.empty(S, M, dtype=torch.bfloat16, device=dev), torch.empty(S, M, dtype=torch.bfloat16, device=dev)
        d_wg, d_w13, d_w2 = torch.empty_like(wg), torch.empty_like(w13), torch.empty_like(w2)
        lay.forward(x, wg, w13, w2, out, noise=noise)
        lay.backward(dy, wg, w13, w2, dx, d_wg, d_w13, d_w2, accumulate=False, loss_scale_dev=None, aux_factor=0.0)
        res[fp8] = (lay, out.clone(), dx.clone(), d_wg.clone(), d_w13.clone(), d_w2.clone(), lay.expert.clone())
    assert torch.equal(res[True][6], res[False][6]), "the routing must not depend on the experts' arithmetic"
    for name, a, b in zip(("out", "dx", "d_wg", "d_w13", "d_w2"), res[True][1:6], res[False][1:6]):
        r = _rel(a, b)
        print(f"fp8 experts vs bf16 experts, {name}: relative l2 {r:.3e}")
        assert torch.isfinite(a.float()).all() and r <= 8e-2, name
    assert _rel(res[True][1], res[False][1]) >= 1e-3    # (the fp8 path did run)
    lay = res[True][0]
    assert len(lay._wq) == 2
    w2b = (w2.float() * 2.0).to(torch.bfloat16)
    w2.copy_(w2b)                                       # the optimizer updates the weights in place ...
    out_stale, out_new = torch.empty_like(res[True][1]), torch.empty_like(res[True][1])
    lay.forward(x, wg, w13, w2, out_stale, noise=noise)
    lay.invalidate_fp8()                                # ... and says so
    lay.forward(x, wg, w13, w2, out_new, noise=noise)
    assert _rel(out_stale, res[True][1]) <= 1e-6 and abs(_rel(out_new, res[True][1]) - 1.0) <= 5e-2   # out scales with w2
    with pytest.raises(ValueError):
        MoELayer(192, 512, E, S, dev, 1.0, 4, expert_fp8=True)


@pytest.mark.skipif(F8 is None, reason="torch without float8_e4m3fn")
def test_moe_engine_with_fp8_expert_products_follows_the_bf16_engine(dev):
    """MoEEngine(expert_fp8=True) on the tiny INTERNLM_MoE configuration of the reference fixtures, same weights, batches and injected noise as the bf16 engine:
    four training steps, every loss within 3e-2 of the bf16 engine's, no skipped step, the weights re-quantised after every optimizer step."""
    import json
    import os

    from internevo_amd.config import ModelConfig, PathConfig, TrainConfig
    from internevo_amd.data import SyntheticLoader
    from internevo_amd.moe_engine import MoEEngine
    from oracle import moe as MO
    from oracle.model import moe_formula_init

    G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
    c = json.load(open(os.path.join(G, "train_moe_bf16.json")))["config"]
    mc = ModelConfig(vocab_size=c["vocab"], hidden_size=c["hidden"], num_layers=c["layers"], num_attention_heads=c["heads"], num_kv_attention_heads=c["heads"],
                     mlp_ratio=4 / 3, model_type="INTERNLM_MoE", num_experts=c["num_experts"], moe_capacity_factor=c["capacity_factor"], moe_loss_coeff=0.1)
    tc = TrainConfig(seq_len=c["seq_len"], micro_bsz=1, micro_num=c["micro_num"], total_steps=c["total_steps"], lr=1e-3, fixed_random_dataset_seqlen=True)
    cfg = PathConfig(mc, tc)
    losses = {}
    for fp8 in (False, True):
        eng = MoEEngine(cfg, dev, init_fn=moe_formula_init, noise_fn=lambda call, S, E: MO.gumbel_noise((S, E), 5000 + call).to(dev), expert_fp8=fp8)
        assert eng.expert_fp8 == fp8 and all(lay.fp8 == fp8 for lay in eng.moe)
        loader = iter(SyntheticLoader(tc.seq_len, 1, tc.micro_num, True, 4000))
        out = []
        for _ in range(4):
            batch, labels = next(loader)
            r = eng.forward_backward(batch, labels)
            loss = r[0] if isinstance(r, (tuple, list)) else r
            eng.step()
            out.append(float(loss))
        st = eng.read_state()
        assert st.skipped_total == 0 and all(math.isfinite(v) for v in out)
        losses[fp8] = out
    print("bf16 experts:", losses[False], "| fp8 experts:", losses[True])
    for a, b in zip(losses[True], losses[False]):
        assert abs(a - b) <= 3e-2 * b
    assert losses[True] != losses[False]
