"""GPU parity tests: every HIP kernel (through the C ABI) against the CPU oracle on seeded inputs.

Tolerances (stated per test): bf16 outputs are compared with the oracle's bf16 result allowing a
couple of bf16 ulps (rtol 1.6e-2) because reduction order / rsqrt / exp differ in the last fp32 bit;
integer/index work is exact.
"""
import math

import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import ops as O  # noqa: E402  (tests may use the oracle; the product never does)


def K():
    from internevo_amd import kernels

    return kernels


def bf(t):
    return t.to(torch.bfloat16)


FLASH_RMS = 5e-3   # ||got - ref|| / ||ref|| bound of every flash-attention output / gradient (bf16 P and dS tiles, fp32 accumulation)


def close(got, ref, rtol, atol, what="", rms=None):
    """Element-wise |got - ref| <= atol + rtol |ref|, and -- `rms` given -- the relative l2 error ||got - ref|| / ||ref|| <= rms: the element-wise
    bound has to admit the bf16 rounding of the largest elements, the l2 bound catches a systematic error of a few per cent in one tile
    product that the absolute term would let through."""
    got = got.detach().float().cpu()
    ref = ref.detach().float().cpu()
    assert got.shape == ref.shape, f"{what}: shape {got.shape} vs {ref.shape}"
    if rms is not None:
        rel = float((got - ref).double().norm() / ref.double().norm().clamp_min(1e-30))
        print(f"[rel-l2] {what}: {rel:.3e} (bound {rms:.1e})")
        assert rel <= rms, f"{what}: relative l2 error {rel:.3e} > {rms:.1e}"
    err = (got - ref).abs()
    tol = atol + rtol * ref.abs()
    bad = err > tol
    if bad.any():
        idx = torch.nonzero(bad)[0].tolist()
        raise AssertionError(
            f"{what}: {int(bad.sum())}/{bad.numel()} mismatches; max abs err {err.max():.4e}; first at {idx}: "
            f"got {got[tuple(idx)]:.6e} ref {ref[tuple(idx)]:.6e}; ref absmax {ref.abs().max():.4e}"
        )


def g(seed):
    return torch.Generator().manual_seed(seed)


# ---------------------------------------------------------------------------------------------- layout pin
def test_mfma_fragment_layout(dev):
    a = torch.randint(-4, 5, (32, 16), generator=g(0)).float()
    b = torch.randint(-4, 5, (16, 32), generator=g(1)).float() + torch.arange(32).float()[None, :] % 3  # asymmetric
    c = K().mfma_probe(bf(a).to(dev), bf(b).to(dev))
    close(c, a @ b, 0, 0, "mfma 32x32x16 bf16 layout")


# ---------------------------------------------------------------------------------------------- K5
@pytest.mark.parametrize("rows,cols", [(8, 512), (4096, 4096), (33, 1024), (5, 4), (7, 200), (16, 8192)])
def test_rmsnorm_fwd_bf16(dev, rows, cols):
    x = bf(torch.randn(rows, cols, generator=g(2)) * 3)
    w = bf(1 + 0.1 * torch.randn(cols, generator=g(3)))
    y, rstd = K().rmsnorm_fwd(x.to(dev), w.to(dev), 1e-5)
    close(y, O.rms_norm(x, w, 1e-5), 1.6e-2, 1e-6, "rmsnorm fwd (<= 2 bf16 ulp: double rounding)")
    close(rstd, torch.rsqrt(x.float().pow(2).mean(-1) + 1e-5), 1e-5, 1e-7, "rstd")


@pytest.mark.parametrize("xdt,wdt", [(torch.float32, torch.bfloat16), (torch.float32, torch.float32), (torch.bfloat16, torch.float32)])
def test_rmsnorm_fwd_mixed(dev, xdt, wdt):
    x = (torch.randn(64, 512, generator=g(4)) * 2).to(xdt)
    w = (1 + 0.1 * torch.randn(512, generator=g(5))).to(wdt)
    y, _ = K().rmsnorm_fwd(x.to(dev), w.to(dev), 1e-5)
    assert y.dtype == wdt
    close(y, O.rms_norm(x, w, 1e-5), 8e-3 if wdt == torch.bfloat16 else 2e-6, 1e-6, "rmsnorm mixed")


def test_rmsnorm_reference_golden_4x4(dev):
    # the reference's own known-answer test, tests/test_model/test_norm.py:30-61
    x = torch.tensor([[8.3726, 1.9245, 5.5101, 1.0000], [3.3474, 2.9582, 1.0000, 1.0000],
                      [8.3726, 1.2875, 5.5101, 1.0000], [8.3726, 1.2875, 5.5101, 1.0000]])
    golden = torch.tensor([[1.6329, 0.3753, 1.0746, 0.1950], [1.4288, 1.2626, 0.4268, 0.4268],
                           [1.6490, 0.2536, 1.0852, 0.1970], [1.6490, 0.2536, 1.0852, 0.1970]])
    y, _ = K().rmsnorm_fwd(x.to(dev), torch.ones(4, device=dev), 1e-5)
    close(y, golden, 1e-3, 5e-3, "test_norm.py golden")


@pytest.mark.parametrize("rows,cols", [(64, 512), (4096, 4096), (9, 1024), (5, 200)])
def test_add_rmsnorm_fwd(dev, rows, cols):
    a = bf(torch.randn(rows, cols, generator=g(6)))
    b = bf(torch.randn(rows, cols, generator=g(7)))
    w = bf(1 + 0.1 * torch.randn(cols, generator=g(8)))
    r, y, _ = K().add_rmsnorm_fwd(a.to(dev), b.to(dev), w.to(dev), 1e-5)
    r_ref = a + b
    close(r, r_ref, 0, 0, "residual add (bit exact)")
    close(y, O.rms_norm(r_ref, w, 1e-5), 1.6e-2, 1e-6, "add+rmsnorm (<= 2 bf16 ulp)")


@pytest.mark.parametrize("rows,cols,res", [(64, 512, False), (4096, 4096, True), (37, 1024, True), (6, 200, True)])
def test_rmsnorm_bwd(dev, rows, cols, res):
    x = bf(torch.randn(rows, cols, generator=g(9)))
    dy = bf(torch.randn(rows, cols, generator=g(10)))
    w = bf(1 + 0.1 * torch.randn(cols, generator=g(11)))
    dres = bf(torch.randn(rows, cols, generator=g(12))) if res else None
    _, rstd = K().rmsnorm_fwd(x.to(dev), w.to(dev), 1e-5)
    dx, dw = K().rmsnorm_bwd(dy.to(dev), x.to(dev), w.to(dev), rstd, dres.to(dev) if res else None)
    dx_ref, dw_ref = O.rms_norm_bwd_fp32(dy, x, w, 1e-5)
    if res:
        dx_ref = dx_ref + dres.float()
    close(dx, dx_ref, 1.6e-2, 2e-2, "rmsnorm dx")
    close(dw, dw_ref, 1.6e-2, 2e-2 * math.sqrt(rows), "rmsnorm dw")


def test_rmsnorm_bwd_accumulate(dev):
    x = bf(torch.randn(128, 512, generator=g(13)))
    dy = bf(torch.randn(128, 512, generator=g(14)))
    w = bf(torch.ones(512))
    _, rstd = K().rmsnorm_fwd(x.to(dev), w.to(dev), 1e-5)
    dw0 = bf(torch.randn(512, generator=g(15))).to(dev)
    dw_acc = dw0.clone()
    _, dw1 = K().rmsnorm_bwd(dy.to(dev), x.to(dev), w.to(dev), rstd)
    K().rmsnorm_bwd(dy.to(dev), x.to(dev), w.to(dev), rstd, dw_out=dw_acc, accumulate=True)
    close(dw_acc, (dw0.float() + dw1.float()).to(torch.bfloat16), 0, 0, "dw accumulate = bf16(old + bf16(new))")


# ---------------------------------------------------------------------------------------------- K2
@pytest.mark.parametrize("conj", [False, True])
@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float32])
def test_apply_rotary_generic(dev, conj, dtype):
    B, S, H, D = 2, 33, 3, 64
    x = torch.randn(B, S, H, D, generator=g(16)).to(dtype)
    cos, sin = O.rotary_cos_sin(S, D, dtype=dtype)
    xd = x.to(dev)
    out = torch.empty_like(xd)
    x1, x2 = xd[..., : D // 2], xd[..., D // 2 :]
    o1, o2 = out[..., : D // 2], out[..., D // 2 :]
    K().apply_rotary(x1, x2, cos.to(dev)[:, None, :], sin.to(dev)[:, None, :], o1, o2, conj)
    r1, r2 = O.apply_rotary(x[..., : D // 2], x[..., D // 2 :], cos[:, None, :], sin[:, None, :], conj)
    close(out, torch.cat([r1, r2], -1), 8e-3 if dtype == torch.bfloat16 else 1e-6, 1e-6, "apply_rotary")


def test_apply_rotary_inplace_interleaved_view(dev):
    # q1 = q[..., ::2] style strided views are NOT unit-stride; the reference only ever passes chunk() halves
    S, H, D = 17, 4, 128
    x = bf(torch.randn(1, S, H, D, generator=g(17)))
    cos, sin = O.rotary_cos_sin(S, D)
    xd = x.to(dev).clone()
    K().apply_rotary(xd[..., :64], xd[..., 64:], cos.to(dev)[:, None], sin.to(dev)[:, None], xd[..., :64], xd[..., 64:], False)
    r1, r2 = O.apply_rotary(x[..., :64], x[..., 64:], cos[:, None, :], sin[:, None, :])
    close(xd, torch.cat([r1, r2], -1), 8e-3, 1e-6, "apply_rotary in place")


@pytest.mark.parametrize("d,hkv,qpk,T,interleaved", [(128, 8, 4, 256, True), (64, 2, 4, 100, True), (128, 2, 1, 64, False), (64, 1, 2, 33, True)])
def test_qkv_rotary_fwd_bwd(dev, d, hkv, qpk, T, interleaved):
    qkv = bf(torch.randn(T, hkv * (qpk + 2) * d, generator=g(18)))
    cos, sin = O.rotary_cos_sin(512, d)
    pos = torch.randint(0, 512, (T,), generator=g(19))
    q, kv = K().qkv_rotary_fwd(qkv.to(dev), cos.to(dev), sin.to(dev), pos.to(dev), hkv, qpk, d, interleaved)
    q_ref, kv_ref = O.qkv_split_rotary(qkv, cos, sin, pos, hkv, qpk, d, interleaved)
    close(q, q_ref, 8e-3, 1e-6, "qkv_rotary q")
    close(kv, kv_ref, 8e-3, 1e-6, "qkv_rotary kv")
    # backward = adjoint: check with fp32 autograd of the oracle
    x32 = qkv.float().requires_grad_(True)
    q32, kv32 = O.qkv_split_rotary(x32, cos.float(), sin.float(), pos, hkv, qpk, d, interleaved)
    dq = bf(torch.randn(q32.shape, generator=g(20)))
    dkv = bf(torch.randn(kv32.shape, generator=g(21)))
    (q32 * dq.float()).sum().backward(retain_graph=True)
    (kv32 * dkv.float()).sum().backward()
    dqkv = K().qkv_rotary_bwd(dq.to(dev), dkv.to(dev), cos.to(dev), sin.to(dev), pos.to(dev), hkv, qpk, d, interleaved)
    close(dqkv, x32.grad, 8e-3, 1e-6, "qkv_rotary bwd")


@pytest.mark.parametrize("T,hkv,qpk,d,Kd,interleaved,scale,fused", [
    (4096, 8, 4, 128, 1024, True, 1.0, 1),      # 16 x 24 tiles: the persistent frame takes it -- the epilogue form (InternLM2-7B's head geometry)
    (4096, 8, 4, 128, 1024, False, 1.0, 1),     # adapt_hf: halves instead of even / odd pairs
    (4096, 8, 4, 128, 1024, True, 0.1275, 1),   # the scale_on_q arrangement: the factor on the rotated q before its one rounding
    (2048, 4, 1, 128, 512, True, 1.0, 0),       # 8 x 6 = 48 tiles: one round, the frame does not take it -> two launches
    (1000, 8, 4, 128, 1024, True, 1.0, 0),      # ragged rows -> two launches
    (2048, 4, 4, 64, 512, True, 1.0, 0),        # head dim 64 -> two launches
    (16384, 8, 4, 128, 4096, True, 1.0, 1),     # the benchmark's own product
], ids=["frame", "frame_adapt_hf", "frame_scaled_q", "one_round", "ragged_rows", "d64", "7b_shape"])
def test_wqkv_product_with_split_and_rotary_in_its_epilogue_equals_the_two_launches_bit_for_bit(dev, T, hkv, qpk, d, Kd, interleaved, scale, fused):
    """ie_gemm_qkv_rotary_fwd (a3 + a4; round 6): where the persistent GEMM frame takes the wqkv product of a d = 128 model, the GQA split, the even / odd
    de-interleave, the cos / sin gather by position and the rotation sit in the product's epilogue (gemm_p5_k<false, 3>) and the [T, N] product never reaches
    memory; elsewhere product + ie_qkv_rotary_fwd_scaled.  Same arithmetic on the same bf16-rounded products: q and kv must be IDENTICAL to the two launches
    (which the oracle test above pins), for both slot orders, with the q scale, and the dispatch must be what the shape says."""
    Kk = K()
    L = Kk._L()
    N = hkv * (qpk + 2) * d
    gen = torch.Generator(device=dev).manual_seed(23)
    x = torch.randn(T, Kd, device=dev, generator=gen).to(torch.bfloat16)
    w = (torch.randn(N, Kd, device=dev, generator=gen) * 0.05).to(torch.bfloat16)
    cos, sin = O.rotary_cos_sin(8192, d)
    cos, sin = cos.to(dev), sin.to(dev)
    pos = torch.randint(0, 8192, (T,), device=dev, generator=gen)
    assert int(L.ie_gemm_qkv_rotary_is_fused(T, hkv, qpk, d, Kd)) == fused
    out = {}
    for mode in (0, 1):
        L.ie_tune_qkv_rotary_fuse(mode)
        try:
            q = torch.full((T, hkv * qpk, d), 7.0, device=dev, dtype=torch.bfloat16)
            kv = torch.full((T, 2, hkv, d), 7.0, device=dev, dtype=torch.bfloat16)
            scratch = torch.full((T, N), 7.0, device=dev, dtype=torch.bfloat16)
            Kk.linear_qkv_rotary_fwd(x, w, cos, sin, pos, hkv, qpk, d, interleaved, q, kv, scratch, scale)
            torch.cuda.synchronize()
            out[mode] = (q, kv, scratch)
        finally:
            L.ie_tune_qkv_rotary_fuse(1)
    for name, a, b in zip(("q", "kv"), out[0], out[1]):
        assert torch.equal(a, b), f"{name}: {int((a != b).sum())} of {a.numel()} elements differ, max |diff| {float((a.float() - b.float()).abs().max())}"
    # the two-launch path IS product + the rotary kernel; the epilogue form leaves the scratch product untouched
    ref = Kk.linear_fwd(x, w)
    q2, kv2 = Kk.qkv_rotary_fwd(ref, cos, sin, pos, hkv, qpk, d, interleaved, q_scale=scale)
    assert torch.equal(out[0][2], ref) and torch.equal(out[0][0], q2) and torch.equal(out[0][1], kv2)
    if fused:
        assert bool((out[1][2] == 7.0).all()), "the fused form must not write the [T, N] product"


@pytest.mark.parametrize("M,N,Kd,rows,cols", [(6144, 4096, 2048, slice(4096, 6144), slice(None)), (4096, 14336, 1024, slice(None), slice(12288, 14336)),
                                              (4096, 4096, 2048, None, None)], ids=["wqkv_384_tiles", "w2_896_tiles", "whole_round_no_split"])
def test_weight_gradient_tail_k_split_is_the_two_half_products_added_in_fixed_order(dev, M, N, Kd, rows, cols):
    """ie_gemm_set_wgrad_ksplit_workspace (round 6; linear_bias_wgrad, model/utils.py:293-299): a weight-gradient product whose 256x256 tiling ends in a round that is
    at most half full computes the remainder rectangle as two half-k products in one launch + a fix-up.  Pinned: the rectangle in front of the cut is the unsplit
    product bit for bit; the remainder is EXACTLY bf16(p0 + p1) of the two half products computed on their own (and bf16(old + that) when accumulating) -- so it
    differs from the unsplit product by the bf16 rounding of the two PARTIAL sums (half a bf16 ulp of each, as every per-micro-batch partial of the reference's
    bf16 `.grad +=` carries) and by nothing else; a tiling of whole rounds is untouched; two runs give the same bits."""
    Kk = K()
    gen = torch.Generator(device=dev).manual_seed(29)
    dy = torch.randn(Kd, M, device=dev, generator=gen).to(torch.bfloat16)      # [tokens, out features]: A k-major
    x = torch.randn(Kd, N, device=dev, generator=gen).to(torch.bfloat16)       # [tokens, in features]: B k-major
    old = (torch.randn(M, N, device=dev, generator=gen) * 4).to(torch.bfloat16)
    Kk.enable_wgrad_ksplit(dev, False)
    try:
        plain = Kk.linear_wgrad(dy, x)
        plain_acc = Kk.linear_wgrad(dy, x, old.clone(), accumulate=True)
        h = Kd // 2
        p0, p1 = Kk.linear_wgrad(dy[:h], x[:h]), Kk.linear_wgrad(dy[h:], x[h:])
        Kk.enable_wgrad_ksplit(dev, True)
        split = Kk.linear_wgrad(dy, x)
        again = Kk.linear_wgrad(dy, x)
        split_acc = Kk.linear_wgrad(dy, x, old.clone(), accumulate=True)
    finally:
        Kk.enable_wgrad_ksplit(dev, True)    # (the engines' default)
    assert torch.equal(split, again)
    if rows is None:
        assert torch.equal(split, plain) and torch.equal(split_acc, plain_acc)
        return
    mask = torch.zeros(M, N, dtype=torch.bool, device=dev)
    mask[rows, cols] = True
    assert torch.equal(split[~mask], plain[~mask]) and torch.equal(split_acc[~mask], plain_acc[~mask]), "the whole rounds must be the unsplit product"
    want = (p0.float() + p1.float()).to(torch.bfloat16)
    assert torch.equal(split[mask], want[mask]), "remainder != bf16(half 0 + half 1)"
    want_acc = (old.float() + want.float()).to(torch.bfloat16)
    assert torch.equal(split_acc[mask], want_acc[mask]), "accumulating remainder != bf16(old + bf16(half 0 + half 1))"
    assert not torch.equal(split[mask], plain[mask]) or Kd < 256   # (it IS another rounding order ...)
    err = (split.float() - plain.float()).abs()
    bound = 2.0 ** -8 * (p0.float().abs() + p1.float().abs() + 2 * plain.float().abs()) + 1e-6   # half an ulp of each partial + the roundings of the two results
    assert bool((err <= bound).all()), f"... beyond the partial sums' roundings: worst {float((err / bound).max()):.2f} x the bound"


# ---------------------------------------------------------------------------------------------- K8
@pytest.mark.parametrize("rows,cols", [(16, 1792), (4096, 14336), (3, 8)])
def test_swiglu(dev, rows, cols):
    ab = bf(torch.randn(rows, 2 * cols, generator=g(22)) * 2)
    a, b = ab[:, :cols], ab[:, cols:]
    abd = ab.to(dev)
    out = K().swiglu_fwd(abd[:, :cols], abd[:, cols:])
    close(out, O.swiglu(a, b), 4e-3, 1e-30, "swiglu fwd (<= 1 bf16 ulp)")
    do = bf(torch.randn(rows, cols, generator=g(23)))
    a32, b32 = a.float().requires_grad_(True), b.float().requires_grad_(True)
    (O.swiglu(a32, b32) * do.float()).sum().backward()
    act = torch.empty(rows, cols, dtype=torch.bfloat16, device=dev)
    da, db = K().swiglu_bwd(do.to(dev), abd[:, :cols], abd[:, cols:], act_out=act)
    close(da, a32.grad, 2e-2, 1e-3, "swiglu da")
    close(db, b32.grad, 2e-2, 1e-3, "swiglu db")
    close(act, O.swiglu(a, b), 4e-3, 1e-30, "swiglu recomputed act")


# ---------------------------------------------------------------------------------------------- K4
@pytest.mark.parametrize("rows,V,dtype,ls", [(64, 1024, torch.bfloat16, 0.0), (16, 92544, torch.bfloat16, 0.0), (33, 1000, torch.float32, 0.0),
                                             (20, 517, torch.bfloat16, 0.0), (32, 1024, torch.float32, 0.1)])
def test_cross_entropy(dev, rows, V, dtype, ls):
    logits = (torch.randn(rows, V, generator=g(24)) * 3).to(dtype)
    labels = torch.randint(0, V, (rows,), generator=g(25))
    labels[::5] = -100
    lo32 = logits.float().requires_grad_(True)
    ref = O.cross_entropy(lo32, labels, ls)
    ref.backward()
    ld = logits.to(dev).clone()
    loss_rows, lse, loss, count = K().ce_fwd(ld, labels.to(dev), -100, ls)
    close(loss, ref.reshape(1), 2e-5, 1e-5, "CE mean loss")
    close(count, torch.tensor([float((labels != -100).sum())]), 0, 0, "CE count")
    close(lse, torch.logsumexp(logits.float(), -1), 2e-5, 1e-5, "CE lse")
    dloss = torch.tensor([3.0], device=dev)
    K().ce_bwd(ld, labels.to(dev), lse, dloss, count, 0.5, -100, ls)  # in place; grad_out = 3 * 0.5
    close(ld, lo32.grad * 1.5, 1.6e-2 if dtype == torch.bfloat16 else 1e-4, 1e-6, "CE dlogits (in place)")


# ---------------------------------------------------------------------------------------------- K6
@pytest.mark.parametrize("n,dtype", [(1, torch.bfloat16), (1000003, torch.bfloat16), (4096 * 4096, torch.bfloat16), (77777, torch.float32)])
def test_sumsq(dev, n, dtype):
    x = torch.randn(n, generator=g(26)).to(dtype)
    got = K().sumsq(x.to(dev))
    close(got, x.double().pow(2).sum().float().reshape(1), 1e-5, 0, "sumsq")


def test_sumsq_list_and_overflow(dev):
    xs = [bf(torch.randn(n, generator=g(27 + i))) for i, n in enumerate([5, 4096, 100001])]
    got = K().sumsq([x.to(dev) for x in xs])
    close(got.sqrt(), O.l2_norm(xs).reshape(1), 1e-5, 0, "multi-tensor l2norm")
    bad = xs[1].clone()
    bad[7] = float("inf")
    assert math.isinf(K().sumsq(bad.to(dev)).item())
    bad[7] = float("nan")
    assert math.isnan(K().sumsq(bad.to(dev)).item())


# ---------------------------------------------------------------------------------------------- a15 / a17
def test_step_control_matches_reference_logic(dev):
    from internevo_amd._lib import IeScalerConfig

    k = K()
    cfg = IeScalerConfig(2.0, 0.5, 1.0, float(2**24), 3, 2, 1.0, 1)  # growth every 3 clean steps for the test
    st = k.step_state_new(dev, 2.0**16)
    ref = O.DynamicGradScaler(2**16, 2, 0.5, 3, 1, 2**24, 2)
    seq = [4.0e9, float("inf"), 1.0e8, float("inf"), float("inf"), 9.0e6, 1e3, float("nan"), 2.5e9, 1e2, 1e2, 1e2]
    adam_steps = 0
    for ss in seq:
        k.step_control(st, torch.tensor([ss], device=dev), cfg)
        s = k.step_state_read(st)
        backup = ref.scale
        found_inf, found_nan = math.isinf(ss), math.isnan(ss)
        ref.update(found_inf)
        assert s.loss_scale == ref.scale and s.growth_step == ref.growth_step and s.hysteresis_step == ref.hysteresis_step
        assert s.loss_scale_used == backup
        assert s.skip == int(found_inf or found_nan)
        if not s.skip:
            adam_steps += 1
            norm = ss**0.5
            comb = O.unscale_clip_factor(norm, backup, 1.0)
            assert abs(s.inv_scale - 1.0 / comb) <= 1e-7 / comb
            assert abs(s.grad_norm - norm / backup) <= 1e-6 * norm / backup
        assert s.adam_step == adam_steps


def test_adamw_matches_torch(dev):
    k = K()
    from internevo_amd._lib import IeScalerConfig

    n = 100003
    p0 = torch.randn(n, generator=g(30))
    p_ref, m_ref, v_ref = p0.clone(), torch.zeros(n), torch.zeros(n)
    p32, m, v = p0.to(dev), torch.zeros(n, device=dev), torch.zeros(n, device=dev)
    p16 = torch.empty(n, dtype=torch.bfloat16, device=dev)
    st = k.step_state_new(dev, 65536.0)
    cfg = IeScalerConfig(2.0, 0.5, 1.0, float(2**24), 1000, 2, 1.0, 1)
    lr, b1, b2, eps, wd = 1e-4, 0.9, 0.95, 1e-8, 0.01
    for step in range(1, 4):
        gr = bf(torch.randn(n, generator=g(30 + step)) * 65536 * 0.01)
        ss = k.sumsq(gr.to(dev))
        k.step_control(st, ss, cfg)
        k.adamw_step(gr.to(dev), p32, m, v, p16, st, lr, b1, b2, eps, wd)
        norm = float(gr.double().pow(2).sum().sqrt())
        comb = O.unscale_clip_factor(norm, 65536.0, 1.0)
        g32 = gr.float() * (1.0 / comb)
        O.adamw_step(p_ref, g32, m_ref, v_ref, step, lr, b1, b2, eps, wd)
        close(p32, p_ref, 2e-6, 1e-7, f"adam p step {step}")
        close(m, m_ref, 2e-5, 1e-9, f"adam m step {step}")
        close(v, v_ref, 2e-5, 1e-12, f"adam v step {step}")
        close(p16, p_ref.to(torch.bfloat16), 8e-3, 0, "bf16 shadow")
    # a skipped step leaves everything untouched
    k.step_control(st, torch.tensor([float("inf")], device=dev), cfg)
    before = p32.clone()
    k.adamw_step(gr.to(dev), p32, m, v, p16, st, lr, b1, b2, eps, wd)
    assert torch.equal(before, p32)


@pytest.mark.parametrize("dim", [4096, 6144, 520, 516])
@pytest.mark.parametrize("accumulate", [False, True])
def test_embedding_backward_is_the_fp32_sum_in_token_order_bit_for_bit(dev, dim, accumulate):
    """embedding_bwd_k: per vocabulary row the fp32 sum of its tokens' gradient rows IN TOKEN ORDER, rounded to bf16 once (+ the old gradient when
    accumulating) -- torch's CPU index_add_ in fp32 walks the index in order, so the comparison is exact (the kernel lists a pass's hits and loads sixteen
    rows before the sixteen adds: the order of the adds must not change).  Heavy duplication (300 tokens on 7 ids: more than sixteen hits per pass), an id at
    the end of the vocabulary, untouched rows, widths off the 256-column block."""
    V, T = 1000, 1300
    ids = torch.randint(0, 200, (T,), generator=g(60))
    ids[:300] = torch.randint(0, 7, (300,), generator=g(61))
    ids[77] = V - 1
    dout = bf(torch.randn(T, dim, generator=g(62)))
    old = bf(torch.randn(V, dim, generator=g(63)))
    dw = old.to(dev).clone() if accumulate else torch.full((V, dim), 7.0, dtype=torch.bfloat16, device=dev)
    K().embedding_bwd(dout.to(dev), ids.to(dev), dw, accumulate=accumulate)
    summed = torch.zeros(V, dim).index_add_(0, ids, dout.float()).to(torch.bfloat16)
    ref = (summed.float() + old.float()).to(torch.bfloat16) if accumulate else summed
    touched = torch.zeros(V, dtype=torch.bool).index_fill_(0, ids, True)
    if accumulate:
        ref[~touched] = old[~touched]
    assert torch.equal(dw.cpu(), ref), f"max diff {(dw.cpu().float() - ref.float()).abs().max()}"


@pytest.mark.parametrize("how", [1, 16, 64, 128])
@pytest.mark.parametrize("gdt", ["bf16", "f32"])
def test_adamw_on_a_few_cus_is_bit_identical_to_the_whole_chip_kernel(dev, how, gdt):
    """ie_tune_adamw_cus(n): n workgroups pinned to a CU each -- the same arithmetic element by element (p, m, v, the bf16 shadow and a skipped step),
    sizes with a ragged tail, odd multiples of the workgroup stride, and below the size the few-CU path takes at all."""
    k = K()
    from internevo_amd._lib import IeScalerConfig

    cfg = IeScalerConfig(2.0, 0.5, 1.0, float(2**24), 1000, 2, 1.0, 1)
    lr, b1, b2, eps, wd = 1e-3, 0.9, 0.95, 1e-8, 0.01
    cus = how
    for n in (cus * 8192 - 5, cus * 8192, cus * 8192 * 3 + 4 * 1024 * cus + 7, 1_000_003 if cus <= 16 else 2_000_003):
        runs = []
        for c in (0, how):
            p32 = torch.randn(n, generator=g(50)).to(dev)
            m, v = torch.zeros(n, device=dev), torch.zeros(n, device=dev)
            p16 = torch.empty(n, dtype=torch.bfloat16, device=dev)
            st = k.step_state_new(dev, 65536.0)
            try:
                k.tune_adamw_cus(c)
                for step in range(1, 3):
                    gr = bf(torch.randn(n, generator=g(50 + step)) * 65536 * 0.01)
                    gr = gr.to(dev) if gdt == "bf16" else gr.float().to(dev)
                    k.step_control(st, k.sumsq(gr), cfg)
                    k.adamw_step(gr, p32, m, v, p16, st, lr, b1, b2, eps, wd)
                kept = p32.clone()
                k.step_control(st, torch.tensor([float("inf")], device=dev), cfg)
                k.adamw_step(gr, p32, m, v, p16, st, lr, b1, b2, eps, wd)
                assert torch.equal(kept, p32), "a skipped step touched the parameters"
            finally:
                k.tune_adamw_cus(0)
            runs.append((p32, m, v, p16))
        for a, b_, what in zip(runs[0], runs[1], ("p", "m", "v", "bf16 shadow")):
            assert torch.equal(a, b_), f"{what} differs at n = {n}, {cus} CUs"


# ---------------------------------------------------------------------------------------------- a10
def test_embedding(dev):
    V, dim, T = 1000, 512, 300
    w = bf(torch.randn(V, dim, generator=g(40)))
    ids = torch.randint(0, 30, (T,), generator=g(41))
    ids[5] = 999
    out = K().embedding_fwd(w.to(dev), ids.to(dev))
    close(out, w[ids], 0, 0, "embedding fwd (exact)")
    dout = bf(torch.randn(T, dim, generator=g(42)))
    dw = torch.full((V, dim), 7.0, dtype=torch.bfloat16, device=dev)
    K().embedding_bwd(dout.to(dev), ids.to(dev), dw, accumulate=False)
    ref = torch.zeros(V, dim).index_add_(0, ids, dout.float())
    close(dw, ref, 8e-3, 1e-3, "embedding bwd")
    K().embedding_bwd(dout.to(dev), ids.to(dev), dw, accumulate=True)
    close(dw, 2 * ref, 1.6e-2, 2e-3, "embedding bwd accumulate")


def test_embedding_gradient_of_the_benchmark_batch_against_fp64(dev):
    """Round-3 review: `oracle.ops.embedding_grad_in_fp32` (the oracle switched to the accelerator kernel's arithmetic for the 7B-width comparisons) must
    rest on ground truth, not on the product.  The token ids of the BENCHMARK's step (SyntheticLoader, 4 x 4096 tokens: a handful of tokens, each thousands
    of times) with an output gradient that has what makes bf16 accumulation swamp -- a common component of every row next to a noise part: the exact sum
    in fp64 is the truth; embedding_bwd_k (fp32 sums per occurring token, rounded once) must sit at bf16 rounding from it and be closer to it than torch's
    CPU kernel on bf16 (F.embedding's backward: rows added one by one into the bf16 gradient -- the arithmetic of the reference's CPU runs) by >= 10x in
    relative l2; the oracle's fp32 switch must give the kernel's arithmetic, and the CPU-bf16 path must show the loss the 7B-width tests ran into."""
    import torch.nn.functional as F

    from internevo_amd.data import SyntheticLoader
    from oracle import ops as O

    batch, _ = next(iter(SyntheticLoader(4096, 1, 4, True, 4000)))
    ids = batch["input_ids"].reshape(-1)
    V, dim = 92544, 1024
    assert ids.numel() == 16384 and int(torch.bincount(ids).max()) >= 500, "the benchmark's data: few tokens, hundreds to thousands of occurrences each"
    dout = bf(torch.randn(ids.numel(), dim, generator=g(46)) * 0.05 + torch.randn(1, dim, generator=g(47)))
    truth = torch.zeros(V, dim, dtype=torch.float64).index_add_(0, ids, dout.double())
    dw = torch.zeros(V, dim, dtype=torch.bfloat16, device=dev)
    K().embedding_bwd(dout.to(dev), ids.to(dev), dw, accumulate=False)
    w = torch.zeros(V, dim, dtype=torch.bfloat16, requires_grad=True)
    F.embedding(ids, w).backward(dout)                     # torch's CPU kernel on bf16
    w32 = torch.zeros(V, dim, dtype=torch.bfloat16, requires_grad=True)
    with O.embedding_grad_in_fp32():
        O.embedding(ids, w32).backward(dout)               # the oracle's switch
    err = lambda x: float((x.double().cpu() - truth).norm() / truth.norm())  # noqa: E731
    e_hip, e_cpu, e_sw = err(dw), err(w.grad), err(w32.grad)
    print(f"embedding gradient of the benchmark batch vs fp64: HIP kernel {e_hip:.2e}, CPU bf16 kernel {e_cpu:.2e}, oracle fp32 switch {e_sw:.2e}; "
          f"CPU bf16 keeps {float(w.grad.double().norm() / truth.norm()):.3f} of the gradient's norm")
    assert e_hip <= 3e-3, "fp32 sums rounded once: bf16 rounding of the result"
    assert e_cpu >= 10 * e_hip, (e_cpu, e_hip)
    # the oracle's fp32 switch is the kernel's arithmetic: fp32 sums rounded once (the order of an fp32 sum may move a rounding here and there)
    differ = float((dw.cpu() != w32.grad).float().mean())
    print(f"elements where the kernel and the oracle's fp32 switch differ: {differ:.2e}")
    assert differ <= 1e-3 and float((dw.cpu().double() - w32.grad.double()).norm() / truth.norm()) <= 1e-3


def test_add_and_cast(dev):
    a = bf(torch.randn(100003, generator=g(43)))
    b = bf(torch.randn(100003, generator=g(44)))
    close(K().add_bf16(a.to(dev), b.to(dev)), a + b, 0, 0, "add (exact)")
    x = torch.randn(4099, generator=g(45))
    close(K().cast(x.to(dev), torch.bfloat16), x.to(torch.bfloat16), 0, 0, "cast fp32->bf16 (exact RNE)")


# ---------------------------------------------------------------------------------------------- K3
GEMM_SHAPES = [(128, 128, 64), (256, 256, 256), (128, 384, 512), (200, 136, 72), (8, 8, 8), (1000, 1016, 200), (512, 1792, 512),
               (1000, 1016, 192), (8, 8, 64), (136, 264, 128), (520, 776, 320)]  # K % 64 == 0 -> LDS-DMA kernels, ragged M/N edges


@pytest.mark.parametrize("M,N,Kd", GEMM_SHAPES)
@pytest.mark.parametrize("akm,bkm", [(False, False), (False, True), (True, True), (True, False)])
def test_gemm_small(dev, M, N, Kd, akm, bkm):
    A = bf(torch.randn((Kd, M) if akm else (M, Kd), generator=g(50)))
    B = bf(torch.randn((Kd, N) if bkm else (N, Kd), generator=g(51)))
    ref = (A.float().t() if akm else A.float()) @ (B.float() if bkm else B.float().t())
    C = K().gemm(A.to(dev), B.to(dev), akm, bkm)
    close(C, ref, 8e-3, 2e-3 * math.sqrt(Kd), f"gemm {M}x{N}x{Kd} akm={akm} bkm={bkm}")


@pytest.mark.parametrize("variant", [4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15, 16, 17, 18, 19])
def test_gemm_every_dma_tile_variant(dev, variant):
    """Each LDS-DMA schedule (ie_gemm_bf16_tile) on ragged M/N edges, a single k-tile, two and many k-tiles, all four operand
    layouts, a strided A view and accumulate -- the dispatcher only ever picks some of them for a given shape."""
    for (M, N, Kd) in [(520, 392, 192), (264, 256, 64), (8, 520, 128), (304, 1000, 1024)]:
        for akm, bkm in ((False, False), (False, True), (True, True), (True, False)):
            Abig = bf(torch.randn((Kd, M + 8) if akm else (M, Kd + 64), generator=g(57)))
            A = Abig[:, :M] if akm else Abig[:, :Kd]
            B = bf(torch.randn((Kd, N) if bkm else (N, Kd), generator=g(58)))
            ref = (A.float().t() if akm else A.float()) @ (B.float() if bkm else B.float().t())
            C0 = bf(torch.randn(M, N, generator=g(59)))
            Ad = Abig.to(dev)
            Ad = Ad[:, :M] if akm else Ad[:, :Kd]
            C = K().gemm(Ad, B.to(dev), akm, bkm, variant=variant)
            close(C, ref, 8e-3, 2e-3 * math.sqrt(Kd), f"variant {variant} {M}x{N}x{Kd} akm={akm} bkm={bkm}")
            Cd = C0.to(dev).clone()
            K().gemm(Ad, B.to(dev), akm, bkm, out=Cd, accumulate=True, variant=variant)
            close(Cd, (C0.float() + ref.to(torch.bfloat16).float()), 1.6e-2, 2e-3 * math.sqrt(Kd) + 0.05, f"variant {variant} accumulate")


@pytest.mark.parametrize("variant,akm,bkm", [(20, False, False), (20, False, True), (21, True, True)], ids=["fwd_refill", "dgrad_refill", "wgrad_ring"])
def test_gemm_schedules_on_16x16x32_mfma(dev, variant, akm, bkm):
    """The schedules issued as v_mfma_f32_16x16x32_bf16 (round 4): variant 20 = the operand-wise refill schedule, the dispatcher's choice for the forward product
    and for the long input-gradient products (B k-major: 16-column fragments by transposing reads of an image whose k-rows k and k + 8 swap 32-byte halves);
    variant 21 = the one-wave-per-SIMD ring for the weight-gradient product (measured 0.5-2 % behind variant 17, selectable only).  Ragged M/N edges, one, two,
    three and many k-tiles, a strided A view and accumulate; layouts a schedule is not written for are refused, not mis-read."""
    for (M, N, Kd) in [(520, 392, 192), (264, 256, 64), (8, 520, 128), (304, 1000, 1024), (1024, 768, 4096)]:
        Abig = bf(torch.randn((Kd, M + 8) if akm else (M, Kd + 64), generator=g(57)))
        A = Abig[:, :M] if akm else Abig[:, :Kd]
        B = bf(torch.randn((Kd, N) if bkm else (N, Kd), generator=g(58)))
        ref = (A.float().t() if akm else A.float()) @ (B.float() if bkm else B.float().t())
        C0 = bf(torch.randn(M, N, generator=g(59)))
        Ad = Abig.to(dev)
        Ad = Ad[:, :M] if akm else Ad[:, :Kd]
        C = K().gemm(Ad, B.to(dev), akm, bkm, variant=variant)
        close(C, ref, 8e-3, 2e-3 * math.sqrt(Kd), f"variant {variant} {M}x{N}x{Kd} akm={akm} bkm={bkm}")
        Cd = C0.to(dev).clone()
        K().gemm(Ad, B.to(dev), akm, bkm, out=Cd, accumulate=True, variant=variant)
        close(Cd, (C0.float() + ref.to(torch.bfloat16).float()), 1.6e-2, 2e-3 * math.sqrt(Kd) + 0.05, f"variant {variant} accumulate")
    A = bf(torch.randn(256, 128, generator=g(57))).to(dev)
    with pytest.raises(Exception):   # the other layout
        K().gemm(A, A, not akm, bkm, variant=variant)


@pytest.mark.parametrize("M,N", [(4096, 6144), (6144, 4096), (4096, 14336)])
def test_gemm_tail_split_is_bit_identical(dev, M, N):
    """With ie_tune_gemm_tail_split on, the automatic GEMM cuts a half-empty last round of 256x256 tiles off into a launch of 128x256 tiles (wqkv: cut along N for
    fwd, along M for wgrad; w2 dgrad / wgrad: along N): same k order per element, so the result must not change by one bit, for
    every operand layout, with and without accumulate, on strided operands."""
    from internevo_amd._lib import load as lib
    Kd = 256
    try:
        for akm, bkm in ((False, False), (False, True), (True, True)):
            Abig = bf(torch.randn((Kd, M + 64) if akm else (M, Kd + 64), generator=g(60))).to(dev)
            A = Abig[:, :M] if akm else Abig[:, :Kd]
            B = bf(torch.randn((Kd, N) if bkm else (N, Kd), generator=g(61))).to(dev)
            C0 = bf(torch.randn(M, N, generator=g(62))).to(dev)
            outs = {}
            for mode in (0, 1, 2, 3):
                assert lib().ie_tune_gemm_tail_split(mode) == 0
                plain = K().gemm(A, B, akm, bkm)
                acc = C0.clone()
                K().gemm(A, B, akm, bkm, out=acc, accumulate=True)
                outs[mode] = (plain, acc)
            ref = (A.float().t() if akm else A.float()) @ (B.float() if bkm else B.float().t())
            close(outs[0][0], ref, 8e-3, 2e-3 * math.sqrt(Kd), "tail split off")
            for mode in (1, 2, 3):
                assert torch.equal(outs[mode][0], outs[0][0]), f"mode {mode} akm={akm} bkm={bkm}"
                assert torch.equal(outs[mode][1], outs[0][1]), f"mode {mode} accumulate akm={akm} bkm={bkm}"
    finally:
        lib().ie_tune_gemm_tail_split(0)
    assert lib().ie_tune_gemm_tail_split(4) != 0


@pytest.mark.parametrize("M,N,Kd,takes", [(4096, 8192, 1024, 1), (4352, 4352, 512, 1), (16384, 4096, 4096, 1), (4096, 4096, 1024, 0), (1000, 4096, 1024, 0)],
                         ids=["frame", "uneven_walk", "7b_wo", "one_round", "ragged_rows"])
def test_residual_add_in_the_product_epilogue_equals_product_then_add_bit_for_bit(dev, M, N, Kd, takes):
    """ie_linear_fwd_add (round 6): out = bf16(bf16(x w^T) + addend) in the persistent frame's epilogue -- the block's residual add behind wo / w2 -- against the
    two-step form the engine ran before (the product, then ie_add_rmsnorm_fwd's r = bf16(a + b)): identical r, and the plain norm of it identical to the fused
    add + norm's y and rstd; on a strided x; where the frame does not take the product the call declines and leaves the output alone."""
    xb = bf(torch.randn(M, Kd + 64, generator=g(100))).to(dev)
    x = xb[:, :Kd]
    w = bf(torch.randn(N, Kd, generator=g(101)) * 0.05).to(dev)
    add = bf(torch.randn(M, N, generator=g(102))).to(dev)
    nw = bf(1.0 + 0.1 * torch.randn(N, generator=g(103))).to(dev)
    out = torch.full((M, N), 7.0, dtype=torch.bfloat16, device=dev)
    took = K().linear_fwd_add(x, w, add, out)
    assert bool(took) == bool(takes)
    if not takes:
        assert bool((out == 7.0).all())
        return
    prod = K().linear_fwd(x, w)
    r_ref, y_ref, rstd_ref = K().add_rmsnorm_fwd(prod, add, nw, 1e-5)
    assert torch.equal(out, r_ref), f"residual sum differs: {(out.float() - r_ref.float()).abs().max()}"
    y, rstd = K().rmsnorm_fwd(out, nw, 1e-5)
    assert torch.equal(y, y_ref) and torch.equal(rstd, rstd_ref), "the plain norm of the summed rows differs from the fused add + norm"
    close(out, x.float().cpu() @ w.float().cpu().t() + add.float().cpu(), 8e-3, 2e-3 * math.sqrt(Kd), "against fp32")


@pytest.mark.parametrize("memset", [0, 1])
def test_persistent_frame_recycles_its_queue_slots(dev, memset):
    """The persistent kernel's tile queues: one slot per stream (zero at module load, every launch's LAST block zeroes it for the next launch of that stream;
    ie_tune_gemm_queue_memset(0), the default since round 6: no memset kernels between the products) -- 100 launches on each of three streams that run at the
    same time (8 blocks per launch: a dozen launches fit on the chip side by side), different tile counts per launch, every result the plain launch's bit for
    bit; then 70 more streams, one launch each: the 64th and later share the last slot and take turns on it.  The same with the memset switched on."""
    from internevo_amd._lib import load as lib
    shapes = [(1024, 1024, 256), (2048, 1280, 256), (1536, 512, 512), (768, 2304, 256)]
    ops = []
    for i, (M, N, Kd) in enumerate(shapes):
        A = bf(torch.randn(M, Kd, generator=g(80 + i))).to(dev)
        B = bf(torch.randn(N, Kd, generator=g(90 + i))).to(dev)
        ops.append((A, B, K().gemm(A, B, False, False, variant=20)))
    torch.cuda.synchronize()
    streams = [torch.cuda.current_stream(dev)] + [torch.cuda.Stream(device=dev) for _ in range(2)]
    try:
        assert lib().ie_tune_gemm_persistent(8) == 0 and lib().ie_tune_gemm_queue_memset(memset) == 0
        outs = []
        for n in range(300):
            A, B, _ = ops[(n // 3 + n) % len(ops)]
            with torch.cuda.stream(streams[n % 3]):
                outs.append(((n // 3 + n) % len(ops), K().gemm(A, B, False, False, variant=22)))
        more = [torch.cuda.Stream(device=dev) for _ in range(70)]
        for n, st in enumerate(more):
            A, B, _ = ops[n % len(ops)]
            with torch.cuda.stream(st):
                outs.append((n % len(ops), K().gemm(A, B, False, False, variant=22)))
        torch.cuda.synchronize()
        for n, (which, o) in enumerate(outs):
            assert torch.equal(o, ops[which][2]), f"launch {n} (memset {memset})"
    finally:
        lib().ie_tune_gemm_persistent(1)
        lib().ie_tune_gemm_queue_memset(0)
    assert lib().ie_tune_gemm_queue_memset(2) != 0


@pytest.mark.parametrize("M,N,Kd,grid", [(1024, 1024, 512, 8), (2048, 1280, 256, 8), (1536, 2560, 1152, 16), (4096, 4096, 1024, 1)])
def test_gemm_persistent_frame_is_bit_identical(dev, M, N, Kd, grid):
    """gemm_p5_k (variant 22; the dispatcher's choice for the eligible forward / input-gradient products since round 5): the 16x16x32 refill schedule in its
    persistent frame -- blocks that walk their tiles, the transfers continued through the tile boundary, the accumulators out through a wave-private LDS turn --
    against the plain launch of the same schedule (variant 20): the k order per element is the same, so not one bit may differ; both layouts (B k-contiguous /
    k-major), with and without accumulate, on strided operands, with blocks that walk 2 - 10 tiles (uneven counts: some blocks one tile more) and with the
    production block count (grid 1 = 256 blocks, one tile each on this shape); plus the oracle product, and the automatic dispatch taking the frame."""
    from internevo_amd._lib import load as lib
    try:
        assert lib().ie_tune_gemm_persistent(grid) == 0
        for bkm in (False, True):
            Abig = bf(torch.randn(M, Kd + 64, generator=g(70))).to(dev)
            A = Abig[:, :Kd]
            B = bf(torch.randn((Kd, N) if bkm else (N, Kd), generator=g(71))).to(dev)
            C0 = bf(torch.randn(M, N, generator=g(72))).to(dev)
            ref = A.float() @ (B.float() if bkm else B.float().t())
            plain = K().gemm(A, B, False, bkm, variant=20)
            pers = K().gemm(A, B, False, bkm, variant=22)
            close(plain, ref, 8e-3, 2e-3 * math.sqrt(Kd), f"variant 20 {M}x{N}x{Kd} bkm={bkm}")
            assert torch.equal(pers, plain), f"persistent frame differs from the plain launch: {M}x{N}x{Kd} bkm={bkm} grid={grid}"
            a20, a22 = C0.clone(), C0.clone()
            K().gemm(A, B, False, bkm, out=a20, accumulate=True, variant=20)
            K().gemm(A, B, False, bkm, out=a22, accumulate=True, variant=22)
            assert torch.equal(a22, a20), f"persistent frame, accumulate: {M}x{N}x{Kd} bkm={bkm} grid={grid}"
            auto = K().gemm(A, B, False, bkm)     # (the automatic choice: the frame when the product has more tiles than blocks; the same bits either way)
            assert torch.equal(auto, plain)
        assert K().gemm(bf(torch.randn(512, 320, generator=g(73))).to(dev), bf(torch.randn(512, 320, generator=g(74))).to(dev), False, False).shape == (512, 512)
        # the w1 | w3 forward product with the SwiGLU gate in the same frame (gemm_p5_k<false, 1>: every wave holds gate and up of its 64 columns): identical to
        # the plain fused launch and to the two-launch path, blocks walking several tiles
        F = N
        x = bf(torch.randn(M, Kd, generator=g(75))).to(dev)
        w13 = bf(torch.randn(2 * F, Kd, generator=g(76)) * 0.05).to(dev)
        res = {}
        for mode in (0, grid):
            assert lib().ie_tune_gemm_persistent(mode) == 0
            h13 = torch.full((M, 2 * F), 7.0, device=dev, dtype=torch.bfloat16)
            act = torch.full((M, F), 7.0, device=dev, dtype=torch.bfloat16)
            K().linear_swiglu_fwd(x, w13, h13, act)
            res[mode] = (h13, act)
        assert torch.equal(res[grid][0], res[0][0]) and torch.equal(res[grid][1], res[0][1]), "fused w1 | w3 product: persistent frame differs from the plain launch"
        lib().ie_tune_gemm_persistent(0)
        ref = K().linear_fwd(x, w13)
        assert torch.equal(res[0][0], ref) and torch.equal(res[0][1], K().swiglu_fwd(ref[:, :F], ref[:, F:]))
    finally:
        lib().ie_tune_gemm_persistent(1)
    assert lib().ie_tune_gemm_persistent(12) != 0 and lib().ie_tune_gemm_persistent(-1) != 0


def test_gemm_accumulate_and_strided(dev):
    M, N, Kd = 256, 384, 320
    Abig = bf(torch.randn(M, Kd + 64, generator=g(52)))
    B = bf(torch.randn(N, Kd, generator=g(53)))
    A = Abig[:, :Kd]
    C0 = bf(torch.randn(M, N, generator=g(54)))
    Cd = C0.to(dev).clone()
    K().gemm(Abig.to(dev)[:, :Kd], B.to(dev), out=Cd, accumulate=True)
    prod = (A.float() @ B.float().t()).to(torch.bfloat16)
    close(Cd, (C0.float() + prod.float()).to(torch.bfloat16), 8e-3, 6e-2, "gemm accumulate")


@pytest.mark.parametrize("akm,bkm", [(False, False), (False, True), (True, True)])
def test_gemm_large_vs_torch(dev, akm, bkm):
    M, N, Kd = 4096, 4096, 4096
    A = bf(torch.randn((Kd, M) if akm else (M, Kd), generator=g(55))).to(dev)
    B = bf(torch.randn((Kd, N) if bkm else (N, Kd), generator=g(56))).to(dev)
    ref = (A.float().t() if akm else A.float()) @ (B.float() if bkm else B.float().t())  # torch fp32 on GPU: test reference only
    C = K().gemm(A, B, akm, bkm)
    close(C, ref, 8e-3, 0.5, "gemm 4096^3")


# ---------------------------------------------------------------------------------------------- K1
def _attn_case(dev, lens, hq, hkv, d, causal, seed):
    T = sum(lens)
    cu = torch.tensor([0] + [sum(lens[: i + 1]) for i in range(len(lens))], dtype=torch.int32)
    q = bf(torch.randn(T, hq, d, generator=g(seed)))
    kv = bf(torch.randn(T, 2, hkv, d, generator=g(seed + 1)))
    do = bf(torch.randn(T, hq, d, generator=g(seed + 2)))
    q32, kv32 = q.float().requires_grad_(True), kv.float().requires_grad_(True)
    ref = O.attention_varlen(q32, kv32, cu, causal)
    (ref * do.float()).sum().backward()
    qd, kvd = q.to(dev), kv.to(dev)
    out, lse = K().flash_attn_fwd(qd, kvd[:, 0], kvd[:, 1], cu.to(dev), max(lens), None, causal)
    close(out, ref, 1.6e-2, 2e-2, f"flash fwd lens={lens} hq={hq} hkv={hkv} d={d} causal={causal}", rms=FLASH_RMS)
    dq, dk, dv = K().flash_attn_bwd(do.to(dev), qd, kvd[:, 0], kvd[:, 1], out, lse, cu.to(dev), max(lens), None, causal)
    close(dq, q32.grad, 2e-2, 3e-2, "flash dq", rms=FLASH_RMS)
    close(dk, kv32.grad[:, 0], 2e-2, 3e-2, "flash dk", rms=FLASH_RMS)
    close(dv, kv32.grad[:, 1], 2e-2, 3e-2, "flash dv", rms=FLASH_RMS)


@pytest.mark.parametrize("lens,hq,hkv,d,causal", [
    ([128], 4, 4, 128, True),
    ([64], 2, 1, 64, True),
    ([256], 8, 2, 128, True),
    ([37, 200, 19], 4, 1, 128, True),
    ([1, 129, 300, 64], 4, 2, 64, True),
    ([100, 157], 2, 2, 128, False),
    ([512], 8, 2, 64, True),
])
def test_flash_attention(dev, lens, hq, hkv, d, causal):
    _attn_case(dev, lens, hq, hkv, d, causal, 60)


@pytest.mark.timeout(1200)
def test_flash_attention_benchmark_regime_matches_oracle(dev):
    """The attention call of the BENCHMARK step itself (bench.py, configs[1] with the micro-batches merged): four packed sequences of 4096
    tokens = 16 384 rows, 32 query / 8 kv heads of 128, causal, through the DEFAULT dispatch -- flash_fwd64_k (automatic from 2048 tokens per
    sequence), the multi-round heaviest-first dK/dV grid with its automatic head split + reduce kernel, the dQ kernel -- forward and
    backward against the CPU oracle (dense fp32 attention per sequence, autograd), element-wise and in relative l2."""
    _attn_case(dev, [4096] * 4, 32, 8, 128, True, 64)


@pytest.mark.timeout(900)
@pytest.mark.parametrize("lens,hq,hkv,d,causal", [([4096, 4096], 8, 2, 128, True), ([2048, 3000], 4, 4, 128, False), ([1024, 777], 4, 2, 64, True)])
def test_flash_forward_folded_softmax_long_sequences(dev, lens, hq, hkv, d, causal):
    """flash_fwd64f_k (variant 3) at benchmark-length sequences against the oracle: forward, the saved log-sum-exp (natural log, from
    log2-unit scores: (mhat + log2 l) ln 2), and the backward through its (out, lse).  Rows of negative-only and of large scores: q is
    shifted for a quarter of the heads so that whole rows sit far below / above the initial reference 0."""
    from internevo_amd import _lib

    L = _lib.load()
    T = sum(lens)
    cu = torch.tensor([0] + [sum(lens[: i + 1]) for i in range(len(lens))], dtype=torch.int32)
    q = bf(torch.randn(T, hq, d, generator=g(90)))
    kv = bf(torch.randn(T, 2, hkv, d, generator=g(91)))
    kv[:, 0, 0, :8] += 2.0                                   # a common component in the keys of kv head 0 ...
    q[:, 0, :8] -= 3.0                                       # ... against which q head 0 scores uniformly low (all scores << 0)
    q[:, hq - 1, :8] += 3.0 if hkv == 1 else 0.0
    do = bf(torch.randn(T, hq, d, generator=g(92)))
    q32, kv32 = q.float().requires_grad_(True), kv.float().requires_grad_(True)
    ref = O.attention_varlen(q32, kv32, cu, causal)
    (ref * do.float()).sum().backward()
    qd, kvd = q.to(dev), kv.to(dev)
    try:
        assert L.ie_tune_flash_fwd_variant(3) == 0
        out, lse = K().flash_attn_fwd(qd, kvd[:, 0], kvd[:, 1], cu.to(dev), max(lens), None, causal)
    finally:
        L.ie_tune_flash_fwd_variant(-1)
    close(out, ref, 1.6e-2, 2e-2, f"folded fwd lens={lens}", rms=FLASH_RMS)
    # log-sum-exp of the first sequence's first kv group, dense
    n0 = lens[0]
    kk = kv[:n0, 0].float().repeat_interleave(hq // hkv, 1)
    sc = torch.einsum("thd,shd->hts", q[:n0].float(), kk) / math.sqrt(d)
    if causal:
        sc = sc.masked_fill(torch.arange(n0)[None, :] > torch.arange(n0)[:, None], float("-inf"))
    close(lse[:, :n0], torch.logsumexp(sc, -1), 2e-3, 2e-2, "folded lse")
    dq, dk, dv = K().flash_attn_bwd(do.to(dev), qd, kvd[:, 0], kvd[:, 1], out, lse, cu.to(dev), max(lens), None, causal)
    close(dq, q32.grad, 2e-2, 3e-2, "dq through the folded forward", rms=FLASH_RMS)
    close(dk, kv32.grad[:, 0], 2e-2, 3e-2, "dk through the folded forward", rms=FLASH_RMS)
    close(dv, kv32.grad[:, 1], 2e-2, 3e-2, "dv through the folded forward", rms=FLASH_RMS)


@pytest.mark.parametrize("lens,hq,hkv,d", [([2048, 2500], 8, 2, 128), ([300, 90], 4, 2, 64)])
def test_attention_with_the_softmax_scale_on_q(dev, lens, hq, hkv, d):
    """The engine's arrangement (engine.py: q_scale / attn_scale / dq_scale): ie_qkv_rotary_fwd_scaled stores q~ = bf16(q * scale * log2 e),
    attention runs with softmax_scale = ln 2 (for long head-dim-128 sequences that selects the folded-softmax kernel), the backward's dq
    is dL/dq~ and ie_qkv_rotary_bwd_scaled multiplies it by the same scale * log2 e (chain rule).  Here with the rotation switched off
    (cos = 1, sin = 0) so that the oracle is plain attention on q: out, lse, dq, dk, dv must match it, and the automatic dispatch must have
    taken variant 3 for the long case (bit-identical to the forced variant)."""
    from internevo_amd import _lib

    L = _lib.load()
    T = sum(lens)
    qpk = hq // hkv
    cu = torch.tensor([0] + [sum(lens[: i + 1]) for i in range(len(lens))], dtype=torch.int32)
    qkv = bf(torch.randn(T, hkv, qpk + 2, d, generator=g(95)))
    do = bf(torch.randn(T, hq, d, generator=g(96)))
    cos = torch.ones(max(lens), d // 2, dtype=torch.bfloat16)
    sin = torch.zeros(max(lens), d // 2, dtype=torch.bfloat16)
    pos = torch.cat([torch.arange(n) for n in lens]).to(torch.int64)
    scale = d ** -0.5
    LOG2E, LN2 = 1.4426950408889634, 0.6931471805599453
    # oracle: q / k / v as the unscaled split gives them (non-interleaved layout: no shuffle)
    q_ref = qkv[:, :, :qpk].reshape(T, hq, d)
    kv_ref = torch.stack([qkv[:, :, qpk], qkv[:, :, qpk + 1]], dim=1)
    q32, kv32 = q_ref.float().requires_grad_(True), kv_ref.float().requires_grad_(True)
    ref = O.attention_varlen(q32, kv32, cu, True)
    (ref * do.float()).sum().backward()
    k = K()
    qs, kvd = k.qkv_rotary_fwd(qkv.reshape(T, -1).to(dev), cos.to(dev), sin.to(dev), pos.to(dev), hkv, qpk, d, False, q_scale=scale * LOG2E)
    close(qs, q_ref.float() * (scale * LOG2E), 8e-3, 1e-6, "pre-scaled q")
    out, lse = k.flash_attn_fwd(qs, kvd[:, 0], kvd[:, 1], cu.to(dev), max(lens), LN2, True)
    if d == 128:
        try:
            assert L.ie_tune_flash_fwd_variant(3) == 0
            out3, lse3 = k.flash_attn_fwd(qs, kvd[:, 0], kvd[:, 1], cu.to(dev), max(lens), LN2, True)
        finally:
            L.ie_tune_flash_fwd_variant(-1)
        assert torch.equal(out, out3) and torch.equal(lse, lse3), "softmax_scale = ln 2 on a long head-dim-128 problem must pick the folded kernel"
    close(out, ref, 1.6e-2, 2e-2, "attention on pre-scaled q", rms=FLASH_RMS)
    dq, dk, dv = k.flash_attn_bwd(do.to(dev), qs, kvd[:, 0], kvd[:, 1], out, lse, cu.to(dev), max(lens), LN2, True)
    dkv = torch.stack([dk, dv], dim=1)
    dqkv = k.qkv_rotary_bwd(dq, dkv, cos.to(dev), sin.to(dev), pos.to(dev), hkv, qpk, d, False, dq_scale=scale * LOG2E).reshape(T, hkv, qpk + 2, d)
    close(dqkv[:, :, :qpk].reshape(T, hq, d), q32.grad, 2e-2, 3e-2, "dq through the scaled pair", rms=FLASH_RMS)
    close(dqkv[:, :, qpk], kv32.grad[:, 0], 2e-2, 3e-2, "dk through the scaled pair", rms=FLASH_RMS)
    close(dqkv[:, :, qpk + 1], kv32.grad[:, 1], 2e-2, 3e-2, "dv through the scaled pair", rms=FLASH_RMS)


def test_flash_attention_lse_and_big_scores(dev):
    # large-magnitude scores exercise the online-softmax rescale path (guide section 5.4 rule 26)
    T, hq, hkv, d = 300, 2, 1, 128
    cu = torch.tensor([0, T], dtype=torch.int32)
    q = bf(torch.randn(T, hq, d, generator=g(70)))
    kv = bf(torch.randn(T, 2, hkv, d, generator=g(71)))
    kv[200, 0] *= 12.0  # one spiky key late in the sequence
    out, lse = K().flash_attn_fwd(q.to(dev), kv.to(dev)[:, 0], kv.to(dev)[:, 1], cu.to(dev), T, None, True)
    ref = O.attention_varlen(q.float(), kv.float(), cu, True)
    close(out, ref, 1.6e-2, 2e-2, "flash fwd with spike", rms=FLASH_RMS)
    k = kv[:, 0].float().repeat_interleave(hq // hkv, 1)
    s = torch.einsum("thd,shd->hts", q.float(), k) / math.sqrt(d)
    s = s.masked_fill(torch.arange(T)[None, :] > torch.arange(T)[:, None], float("-inf"))
    close(lse, torch.logsumexp(s, -1), 1e-3, 1e-2, "flash lse")


@pytest.mark.parametrize("variant", [1, 2, 3, 4, 5, 6])
@pytest.mark.parametrize("lens,hq,hkv,d,causal", [
    ([300, 700, 257], 4, 2, 128, True),    # ragged: 256-row blocks with idle waves, a tail block of one row
    ([1, 129, 64, 512], 4, 1, 64, True),
    ([333, 90], 2, 2, 128, False),
    ([1024], 8, 2, 128, True),
])
def test_flash_forward_64_rows_per_wave(dev, variant, lens, hq, hkv, d, causal):
    """(Variants 4 - 6: the eight-wave kernel flash_fwd8_k -- half-steps, quarters, half-steps with the pipelined matrix phase -- on the same ragged
    packs: blocks with idle waves, one-row tails, one-tile sequences.)
    The one-wave-per-SIMD forward (flash_fwd64_k; picked automatically for long head-dim-128 sequences) against the oracle, with
    the exact rescale (variant 1) and the deferred rescale (variant 2), and the folded-softmax kernel flash_fwd64f_k (variant 3: Q
    prescaled, the reference maximum subtracted by an extra MFMA k-step, per-lane partial row sums), including a late spiky key that
    forces the rescale branch of the deferred forms (guide section 5.4 rule 26) and the log-sum-exp the backward consumes."""
    from internevo_amd import _lib

    L = _lib.load()
    T = sum(lens)
    cu = torch.tensor([0] + [sum(lens[: i + 1]) for i in range(len(lens))], dtype=torch.int32)
    q = bf(torch.randn(T, hq, d, generator=g(80)))
    kv = bf(torch.randn(T, 2, hkv, d, generator=g(81)))
    kv[T - 40, 0] *= 12.0   # raw score far above everything before it: the running maximum jumps past any deferral threshold
    do = bf(torch.randn(T, hq, d, generator=g(82)))
    q32, kv32 = q.float().requires_grad_(True), kv.float().requires_grad_(True)
    ref = O.attention_varlen(q32, kv32, cu, causal)
    (ref * do.float()).sum().backward()
    qd, kvd = q.to(dev), kv.to(dev)
    try:
        assert L.ie_tune_flash_fwd_variant(variant) == 0
        out, lse = K().flash_attn_fwd(qd, kvd[:, 0], kvd[:, 1], cu.to(dev), max(lens), None, causal)
    finally:
        L.ie_tune_flash_fwd_variant(-1)
    close(out, ref, 1.6e-2, 2e-2, f"flash fwd64 variant {variant} lens={lens}", rms=FLASH_RMS)
    if variant == 3:
        # Forced onto UNSCALED q, the folded kernel scales q itself: one more bf16 rounding of q, worth |score| * 2^-9 in the exponent.  Its output
        # is a correct softmax of those scores (checked above), but its lse belongs to them, while the backward recomputes the scores from the
        # unrounded q: on this test's 12x key (scores of +-100) the two differ by up to 0.2 and dQ through that lse is off by 2.5e-2 in l2 --
        # which is why the dispatcher takes this kernel only for q stored pre-scaled (test_attention_with_the_softmax_scale_on_q), where forward
        # and backward see the same scores.
        return
    # the saved log-sum-exp must serve the backward: gradients through the variant's (out, lse)
    dq, dk, dv = K().flash_attn_bwd(do.to(dev), qd, kvd[:, 0], kvd[:, 1], out, lse, cu.to(dev), max(lens), None, causal)
    # (the spiky key makes gradients of magnitude ~10: absolute tolerance relative to the largest reference entry.  dQ's l2 bound is wider
    # here: the rows that see the spike have P ~ one-hot, so dS = P (dP - delta) is a cancellation whose bf16 rounding error is multiplied by
    # the 12x key -- measured 5e-3 ... 1e-2 on these inputs against 2.5e-3 without the spike; dK / dV keep the common bound)
    close(dq, q32.grad, 2e-2, 1e-2 * float(q32.grad.abs().max()), "flash dq from fwd64 lse", rms=4 * FLASH_RMS)
    close(dk, kv32.grad[:, 0], 2e-2, 1e-2 * float(kv32.grad[:, 0].abs().max()), "flash dk from fwd64 lse", rms=FLASH_RMS)
    close(dv, kv32.grad[:, 1], 2e-2, 1e-2 * float(kv32.grad[:, 1].abs().max()), "flash dv from fwd64 lse", rms=FLASH_RMS)


# ---------------------------------------------------------------------------------------------- a18
@pytest.mark.parametrize("A,B,S,C", [(96, 1, 2, 256), (40, 2, 4, 128), (7, 2, 8, 64), (128, 1, 1, 512)])
def test_seq_head_permute_matches_seq_all_to_all_layout(dev, A, B, S, C):
    """[A tokens][B][S head blocks][C] -> [S][A][B][C] = what _SeqAllToAll's tensor_split(+contiguous) along the head dim hands
    to all_to_all (multi_head_attention.py:41-45), and the inverse = torch.cat of the received chunks along the head dim."""
    x = bf(torch.randn(A, B, S, C, generator=g(60)))
    chunks = torch.stack([t.contiguous() for t in torch.tensor_split(x, S, dim=2)])  # [S][A][B][1][C]
    want = chunks.reshape(S, A, B, C)
    got = K().seq_head_permute(x.to(dev), torch.empty(S, A, B, C, dtype=torch.bfloat16, device=dev), A, B, S, C, inverse=False)
    close(got, want, 0, 0, "pack (exact)")
    back = K().seq_head_permute(got, torch.empty(A, B, S, C, dtype=torch.bfloat16, device=dev), A, B, S, C, inverse=True)
    close(back, x, 0, 0, "unpack (exact)")
    y = bf(torch.randn(1001, generator=g(61)))
    close(K().scale_bf16(y.to(dev).clone(), 4.0), y * 4.0, 0, 0, "scale by a power of two (exact)")


@pytest.mark.parametrize("split", [1, 2, 4])
def test_flash_bwd_dkdv_head_split_and_trailing_tokens(dev, split):
    """Every head split of the dK/dV kernel (deterministic fp32 partial sums) gives the same gradients; tokens behind the last
    sequence of the packed buffer (cu_seqlens[-1] < T) are left untouched in dk / dv."""
    k = K()
    lens, hq, hkv, d = [70, 130], 8, 2, 128
    T = sum(lens) + 24  # 24 padding tokens at the end, in no sequence
    cu = torch.tensor([0, 70, 200], dtype=torch.int32)
    q = bf(torch.randn(T, hq, d, generator=g(70)))
    kv = bf(torch.randn(T, 2, hkv, d, generator=g(71)))
    do = bf(torch.randn(T, hq, d, generator=g(72)))
    q32, kv32 = q[:200].float().requires_grad_(True), kv[:200].float().requires_grad_(True)
    ref = O.attention_varlen(q32, kv32, cu, True)
    (ref * do[:200].float()).sum().backward()
    qd, kvd = q.to(dev), kv.to(dev)
    out, lse = k.flash_attn_fwd(qd, kvd[:, 0], kvd[:, 1], cu.to(dev), 130, None, True)
    dkv = torch.full((T, 2, hkv, d), 7.0, dtype=torch.bfloat16, device=dev)
    try:
        k._L().ie_tune_flash_dkdv_split(split)
        dq, dk, dv = k.flash_attn_bwd(do.to(dev), qd, kvd[:, 0], kvd[:, 1], out, lse, cu.to(dev), 130, None, True, dk=dkv[:, 0], dv=dkv[:, 1])
    finally:
        k._L().ie_tune_flash_dkdv_split(0)
    close(dk[:200], kv32.grad[:, 0], 2e-2, 3e-2, f"dk split {split}", rms=FLASH_RMS)
    close(dv[:200], kv32.grad[:, 1], 2e-2, 3e-2, f"dv split {split}", rms=FLASH_RMS)
    assert bool((dkv[200:] == 7.0).all()), "rows of tokens outside every sequence must not be written"


@pytest.mark.parametrize("lens,hq,hkv,d,causal,split", [
    ([300, 700, 257], 4, 2, 128, True, 0),     # ragged; key blocks of 128 with idle waves; 1 .. 11 query tiles per block
    ([1, 129, 64, 512], 4, 1, 64, True, 0),    # a one-token sequence, head dim 64 (8 MFMA gaps per phase)
    ([333, 90], 2, 2, 128, False, 0),          # full attention: every block sees every query tile
    ([1100], 8, 2, 128, True, 2),              # more tiles per block than LDS stages, with the head split (fp32 partial sums)
    ([64, 65, 127, 128, 129], 2, 1, 128, True, 0),   # 1 .. 3 tiles per block: fewer than the four pipeline stages
])
def test_flash_bwd_four_wave_dkdv_blocks(dev, lens, hq, hkv, d, causal, split):
    """ie_tune_flash_bwd_variant(1): the dK/dV kernel with four waves per block, four LDS stages, transfers requested two tiles ahead and
    the tile barrier in front of the last phase (the next tile's first fragments and start values are fetched under it).  Against the
    oracle, and bit-identical to the default two-wave kernel (same arithmetic per key, same order of the query tiles)."""
    from internevo_amd import _lib

    L = _lib.load()
    T = sum(lens)
    cu = torch.tensor([0] + [sum(lens[: i + 1]) for i in range(len(lens))], dtype=torch.int32)
    q = bf(torch.randn(T, hq, d, generator=g(90)))
    kv = bf(torch.randn(T, 2, hkv, d, generator=g(91)))
    do = bf(torch.randn(T, hq, d, generator=g(92)))
    q32, kv32 = q.float().requires_grad_(True), kv.float().requires_grad_(True)
    ref = O.attention_varlen(q32, kv32, cu, causal)
    (ref * do.float()).sum().backward()
    qd, kvd, cud = q.to(dev), kv.to(dev), cu.to(dev)
    out, lse = K().flash_attn_fwd(qd, kvd[:, 0], kvd[:, 1], cud, max(lens), None, causal)
    got = {}
    try:
        L.ie_tune_flash_dkdv_split(split)
        for variant in (0, 1):
            assert L.ie_tune_flash_bwd_variant(variant) == 0
            got[variant] = [t.clone() for t in K().flash_attn_bwd(do.to(dev), qd, kvd[:, 0], kvd[:, 1], out, lse, cud, max(lens), None, causal)]
    finally:
        L.ie_tune_flash_bwd_variant(0)
        L.ie_tune_flash_dkdv_split(0)
    dq, dk, dv = got[1]
    close(dq, q32.grad, 2e-2, 3e-2, "dq (four-wave dK/dV build)", rms=FLASH_RMS)
    close(dk, kv32.grad[:, 0], 2e-2, 3e-2, "dk four waves", rms=FLASH_RMS)
    close(dv, kv32.grad[:, 1], 2e-2, 3e-2, "dv four waves", rms=FLASH_RMS)
    assert torch.equal(got[0][1], dk) and torch.equal(got[0][2], dv), "the two block shapes must give bit-identical dK / dV"


@pytest.mark.parametrize("lens,hq,hkv,fused", [
    ([64, 65, 127, 128, 129, 5], 2, 2, 1),          # ragged sequences, rows past a tile's end, one q head per kv head
    ([333, 90, 700, 1], 6, 2, 1),                   # three q heads per kv head (an odd group: no head split), a one-token sequence
    ([4096, 2048], 32, 8, 1),                       # InternLM2-7B's head geometry, enough key blocks for no head split
    ([4096, 4096, 4096, 4096], 32, 8, 1),           # the benchmark's own call
    ([512, 256], 4, 2, 0),                          # a small problem whose dK / dV kernel splits the heads: not fused -> False, nothing touched
], ids=["ragged", "odd_group", "7b_heads", "7b_call", "head_split"])
def test_attention_backward_with_rotary_and_gqa_rearrange_in_its_stores_equals_the_two_calls_bit_for_bit(dev, lens, hq, hkv, fused):
    """ie_flash_attn_bwd_qkv_rotary (round 6): dQ, dK (rotated back by the tokens' positions) and dV written by the two attention kernels straight into the wqkv
    product's output gradient [T, hkv, hq / hkv + 2, d] -- against ie_flash_attn_bwd into [T, hq, d] / [T, 2, hkv, d] followed by ie_qkv_rotary_bwd: the same
    roundings in the same order (gradient rounded to bf16, rotated in fp32, rounded), so not one bit may differ; packed position ids that restart per sequence."""
    d, T = 128, sum(lens)
    qpk = hq // hkv
    cu = torch.tensor([0] + [sum(lens[: i + 1]) for i in range(len(lens))], dtype=torch.int32).to(dev)
    pos = torch.cat([torch.arange(n) for n in lens]).to(torch.int64).to(dev)
    cos, sin = O.rotary_cos_sin(max(lens), d)
    cos, sin = cos.to(dev), sin.to(dev)
    q = bf(torch.randn(T, hq, d, generator=g(390))).to(dev)
    kv = bf(torch.randn(T, 2, hkv, d, generator=g(391))).to(dev)
    do = bf(torch.randn(T, hq, d, generator=g(392))).to(dev)
    out, lse = K().flash_attn_fwd(q, kv[:, 0], kv[:, 1], cu, max(lens), None, True)
    dqkv = torch.full((T, hkv * (qpk + 2) * d), 7.0, dtype=torch.bfloat16, device=dev)
    took = K().flash_attn_bwd_qkv_rotary(do, q, kv[:, 0], kv[:, 1], out, lse, cu, max(lens), cos, sin, pos, dqkv)
    assert bool(took) == bool(fused)
    if not fused:
        assert bool((dqkv == 7.0).all()), "the unfused shape must leave the output alone"
        return
    dq, dk, dv = K().flash_attn_bwd(do, q, kv[:, 0], kv[:, 1], out, lse, cu, max(lens), None, True)
    ref = K().qkv_rotary_bwd(dq.contiguous(), torch.stack([dk, dv], dim=1).contiguous(), cos, sin, pos, hkv, qpk, d, False)
    if not torch.equal(dqkv, ref):
        a, b_ = dqkv.view(T, hkv, qpk + 2, d), ref.view(T, hkv, qpk + 2, d)
        bad = (a != b_).nonzero()
        raise AssertionError(f"{bad.shape[0]} elements differ; first at (token, kv head, slot, d) = {bad[0].tolist()}: {a[tuple(bad[0])].item()} vs {b_[tuple(bad[0])].item()}")
    # and against the delta kernel's path of the same library (ie_tune_flash_bwd_variant 4 switches the fused call off)
    from internevo_amd import _lib
    L = _lib.load()
    try:
        assert L.ie_tune_flash_bwd_variant(4) == 0
        assert not K().flash_attn_bwd_qkv_rotary(do, q, kv[:, 0], kv[:, 1], out, lse, cu, max(lens), cos, sin, pos, torch.empty_like(dqkv))
    finally:
        L.ie_tune_flash_bwd_variant(0)


@pytest.mark.parametrize("lens,hq,hkv,d,causal", [
    ([1, 129, 64, 512], 4, 2, 64, True),
    ([333, 90], 3, 3, 128, False),
    ([2100], 8, 2, 128, True),
    ([64, 65, 127, 128, 129, 5], 2, 1, 128, True),
    ([4096], 8, 2, 128, True),
])
def test_flash_bwd_delta_in_the_dq_prologue_is_the_delta_kernels_bits(dev, lens, hq, hkv, d, causal):
    """Round 6: the dQ kernel computes delta = sum_d dO * O of its query rows in its prologue (and -delta, -lse / scale for the dK / dV kernel behind it);
    flash_delta_k stays behind ie_tune_flash_bwd_variant bit 2.  The sixteen (eight at d = 64) chunk sums are added in the delta kernel's shuffle tree, so dQ,
    dK and dV of the two paths must agree bit for bit -- ragged sequences, rows past a tile's end, both head dims, causal and full attention."""
    from internevo_amd import _lib

    L = _lib.load()
    T = sum(lens)
    cu = torch.tensor([0] + [sum(lens[: i + 1]) for i in range(len(lens))], dtype=torch.int32).to(dev)
    q = bf(torch.randn(T, hq, d, generator=g(290))).to(dev)
    kv = bf(torch.randn(T, 2, hkv, d, generator=g(291))).to(dev)
    do = bf(torch.randn(T, hq, d, generator=g(292))).to(dev)
    out, lse = K().flash_attn_fwd(q, kv[:, 0], kv[:, 1], cu, max(lens), None, causal)
    res = {}
    try:
        for variant in (0, 4):
            assert L.ie_tune_flash_bwd_variant(variant) == 0
            res[variant] = [t.clone() for t in K().flash_attn_bwd(do, q, kv[:, 0], kv[:, 1], out, lse, cu, max(lens), None, causal)]
    finally:
        L.ie_tune_flash_bwd_variant(0)
    for a, b_, what in zip(res[0], res[4], ("dq", "dk", "dv")):
        assert torch.equal(a, b_), f"{what}: delta in the dQ prologue differs from the delta kernel ({(a.float() - b_.float()).abs().max()})"
    assert L.ie_tune_flash_bwd_variant(8) != 0


@pytest.mark.parametrize("lens,hq,hkv,d,causal,four", [
    ([300, 700, 257], 4, 1, 128, True, False),          # ragged, one kv head for four q heads: a block of the dQ kernel = the four heads of one query tile
    ([300, 700, 257], 4, 1, 128, True, True),           # ... with four-wave dK/dV blocks (the upper key half above the diagonal goes to the spare image)
    ([1, 129, 64, 512], 4, 2, 64, True, False),         # a one-token sequence, head dim 64, two q heads per kv head: two query tiles per dQ block
    ([333, 90], 3, 3, 128, False, True),                # full attention (rectangular image table), one q head per kv head: four query tiles per dQ block
    ([2100], 8, 2, 128, True, True),                    # more key blocks than LDS stages in both kernels
    ([64, 65, 127, 128, 129, 5], 2, 1, 128, True, False),
])
def test_flash_bwd_five_product_spill_path(dev, lens, hq, hkv, d, causal, four):
    """The opt-in five-product backward (ie_flash_attn_bwd_set_spill + ie_tune_flash_bwd_variant bit 1): the dK/dV kernel writes dS^T, dQ is formed
    from it by flash_dq_from_ds_k.  Against the oracle at the flash tests' bounds; dK / dV bit-identical to the default path (the spill changes
    nothing in their arithmetic); without a buffer the same switch runs the default path."""
    from internevo_amd import _lib

    L = _lib.load()
    T = sum(lens)
    cu = torch.tensor([0] + [sum(lens[: i + 1]) for i in range(len(lens))], dtype=torch.int32)
    q = bf(torch.randn(T, hq, d, generator=g(190)))
    kv = bf(torch.randn(T, 2, hkv, d, generator=g(191)))
    do = bf(torch.randn(T, hq, d, generator=g(192)))
    q32, kv32 = q.float().requires_grad_(True), kv.float().requires_grad_(True)
    ref = O.attention_varlen(q32, kv32, cu, causal)
    (ref * do.float()).sum().backward()
    qd, kvd, cud = q.to(dev), kv.to(dev), cu.to(dev)
    out, lse = K().flash_attn_fwd(qd, kvd[:, 0], kvd[:, 1], cud, max(lens), None, causal)
    base = [t.clone() for t in K().flash_attn_bwd(do.to(dev), qd, kvd[:, 0], kvd[:, 1], out, lse, cud, max(lens), None, causal)]
    try:
        assert L.ie_tune_flash_dkdv_split(1) == 0   # (the automatic head split of small problems takes the default path)
        need = K().flash_attn_bwd_spill(True, len(lens), max(lens), hq, causal, dev, four_waves=four)
        assert need == L.ie_flash_attn_bwd_spill_bytes(len(lens), max(lens), hq, int(causal)) > 0
        dq = torch.full((T, hq, d), float("nan"), dtype=torch.bfloat16, device=dev)
        dq, dk, dv = K().flash_attn_bwd(do.to(dev), qd, kvd[:, 0], kvd[:, 1], out, lse, cud, max(lens), None, causal, dq=dq)
        dq, dk, dv = dq.clone(), dk.clone(), dv.clone()
        # no buffer: the switch alone must not change the path
        assert L.ie_flash_attn_bwd_set_spill(None, 0) == 0
        nobuf = [t.clone() for t in K().flash_attn_bwd(do.to(dev), qd, kvd[:, 0], kvd[:, 1], out, lse, cud, max(lens), None, causal)]
    finally:
        K().flash_attn_bwd_spill(False)
        L.ie_tune_flash_dkdv_split(0)
    close(dq, q32.grad, 2e-2, 3e-2, "dq (from the spilled dS^T)", rms=FLASH_RMS)
    close(dk, kv32.grad[:, 0], 2e-2, 3e-2, "dk (spill path)", rms=FLASH_RMS)
    close(dv, kv32.grad[:, 1], 2e-2, 3e-2, "dv (spill path)", rms=FLASH_RMS)
    assert torch.isfinite(dq.float()).all()
    assert torch.equal(nobuf[1], dk) and torch.equal(nobuf[2], dv), "the spill must not change dK / dV"
    assert torch.equal(nobuf[0], base[0]), "without a buffer the default dQ kernel runs"


@pytest.mark.parametrize("scale,norm_head", [(0.1, True), (1.0, True), (0.25, False)])
@pytest.mark.parametrize("rows,cols,view", [(512, 256, False), (37, 1000, False), (64, 4096, True)])
def test_head_weight_function_and_embedding_gradient_scale(dev, scale, norm_head, rows, cols, view):
    """ScaleColumnParallelLinearWithNormHead.forward's weight expression (ops/linear.py:124-136) and its gradient, and the embedding's
    s x + (1 - s) x.detach() (modeling_internlm2.py:970-973), against torch autograd on the same bf16 expressions."""
    import torch.nn.functional as F

    w = bf(torch.randn(rows, cols + (8 if view else 0), generator=g(95)) * 0.05)
    dy = bf(torch.randn(rows, cols, generator=g(96)))
    wv = w[:, :cols]
    wr = wv.clone().requires_grad_(True)
    e = wr * scale + (1 - scale) * wr.detach() if scale != 1 else wr
    y = F.normalize(e) if norm_head else e
    (y.float() * dy.float()).sum().backward()
    wd = w.to(dev)
    out = torch.empty(rows, cols, dtype=torch.bfloat16, device=dev)
    inv = torch.empty(rows, dtype=torch.float32, device=dev)
    K().head_weight_fwd(wd[:, :cols], scale, norm_head, out, inv)
    close(out, y.detach(), 8e-3, 1e-6, "head weight")             # one bf16 rounding of the quotient where torch rounds norm and quotient
    acc0 = bf(torch.randn(rows, cols, generator=g(97)))
    dw = acc0.to(dev).clone()
    K().head_weight_bwd(dy.to(dev), out, inv, scale, norm_head, dw, True)
    close(dw, acc0.float() + wr.grad.float(), 2e-2, 2e-2 * float(wr.grad.float().abs().max()) + 8e-3, "head weight gradient (accumulated)")
    K().head_weight_bwd(dy.to(dev), out, inv, scale, norm_head, dw, False)
    close(dw, wr.grad, 2e-2, 2e-2 * float(wr.grad.float().abs().max()), "head weight gradient")
    x = bf(torch.randn(1000, generator=g(98)))
    want = scale * x + (1 - scale) * x
    close(K().grad_scale_mix(x.to(dev).clone(), scale), want, 0, 0, "embedding mix (exact: the same three roundings)")


# ---------------------------------------------------------------------------------------------- edge cases of the C ABI
def test_empty_inputs_are_no_ops(dev):
    """Zero rows / zero tokens / zero-sized products: every entry point returns success without launching (the reference's torch
    ops accept empty tensors too)."""
    k = K()
    e = lambda *s, dt=torch.bfloat16: torch.empty(*s, dtype=dt, device=dev)  # noqa: E731
    w = torch.ones(64, dtype=torch.bfloat16, device=dev)
    y, rstd = k.rmsnorm_fwd(e(0, 64), w, 1e-5)
    assert y.shape == (0, 64) and rstd.numel() == 0
    assert k.swiglu_fwd(e(0, 128), e(0, 128)).shape == (0, 128)
    assert k.gemm(e(0, 64), e(16, 64)).shape == (0, 16)
    C0 = torch.full((8, 16), 3.0, dtype=torch.bfloat16, device=dev)
    assert bool((k.gemm(e(8, 0), e(16, 0), out=C0.clone(), accumulate=True) == 3.0).all()), "K = 0 with accumulate leaves C alone"
    assert k.sumsq(e(0)).item() == 0.0
    cu = torch.zeros(1, dtype=torch.int32, device=dev)
    out, lse = k.flash_attn_fwd(e(0, 4, 64), e(0, 2, 64), e(0, 2, 64), cu, 0, None, True)
    assert out.shape == (0, 4, 64)
    loss_rows, lse2, mean, cnt = k.ce_fwd(e(0, 32), torch.empty(0, dtype=torch.int64, device=dev))
    assert loss_rows.numel() == 0
    k.embedding_fwd(e(10, 64), torch.empty(0, dtype=torch.int64, device=dev), e(0, 64))


def test_bad_arguments_raise_not_crash(dev):
    """Error convention of the boundary (include/internevo_hip.h): a negative code + ie_last_error(), surfaced as InternEvoHipError
    -- never a device fault -- for unsupported head dims, misaligned views, mismatched contractions."""
    from internevo_amd._lib import InternEvoHipError

    k = K()
    e = lambda *s, dt=torch.bfloat16: torch.zeros(*s, dtype=dt, device=dev)  # noqa: E731
    cu = torch.tensor([0, 16], dtype=torch.int32, device=dev)
    with pytest.raises(InternEvoHipError):
        k.flash_attn_fwd(e(16, 2, 32), e(16, 2, 32), e(16, 2, 32), cu, 16, None, True)  # head dim 32
    with pytest.raises(ValueError):
        k.gemm(e(8, 64), e(8, 32))  # contraction mismatch
    with pytest.raises(InternEvoHipError):
        k.gemm(e(8, 72)[:, 4:68], e(8, 64))  # A view starts 8 bytes off a 16-byte boundary
    with pytest.raises(InternEvoHipError):
        k.gemm(e(64, 64), e(64, 64), variant=99)
    with pytest.raises(ValueError):
        k.rmsnorm_fwd(torch.zeros(4, 64), torch.ones(64), 1e-5)  # host tensors: no CPU fallback


def test_flash_attention_many_short_and_one_long_sequence(dev):
    """Packed batch with sequences of length 1, lengths that are not multiples of any tile, and one sequence longer than all the
    others together (max_seqlen >> mean): varlen indexing, early exits of empty tiles, the longest-job-first dispatch."""
    _attn_case(dev, [1, 1, 63, 2, 65, 1, 700, 3, 129, 1], 8, 2, 128, True, 90)
    _attn_case(dev, [5, 1000, 7], 4, 4, 64, True, 91)


@pytest.mark.timeout(600)
def test_flash_attention_seq32768_properties(dev):
    """BASELINE.json configs[3] (7B_isp_sft: seq 32768, SP = 8): the per-rank attention problem after the Ulysses exchange is
    32768 tokens x 4 q heads x 1 kv head.  Too large for a dense oracle, so size-independent properties:
      * V == 1  =>  every output element is exactly 1 (rows of P sum to one), dV column sums equal dO sums;
      * causality: the first 4096 rows equal the result on the truncated sequence (prefix property), bit for bit;
      * the last 256 query rows against a blocked fp32 reference (S = q_blk K^T is only 256 x 32768);
      * backward: dQ / dK / dV of the truncated problem equal the corresponding slices when dO is zero behind token 4096."""
    k = K()
    T, hq, hkv, d = 32768, 4, 1, 128
    q = bf(torch.randn(T, hq, d, generator=g(100)) * 0.5).to(dev)
    kv = bf(torch.randn(T, 2, hkv, d, generator=g(101)) * 0.5).to(dev)
    cu = torch.tensor([0, T], dtype=torch.int32, device=dev)
    out, lse = k.flash_attn_fwd(q, kv[:, 0], kv[:, 1], cu, T, None, True)
    # prefix property
    P = 4096
    cu_p = torch.tensor([0, P], dtype=torch.int32, device=dev)
    out_p, lse_p = k.flash_attn_fwd(q[:P].contiguous(), kv[:P, 0], kv[:P, 1], cu_p, P, None, True)
    assert torch.equal(out[:P], out_p) and torch.equal(lse[:, :P], lse_p)
    # V == 1
    kv1 = kv.clone()
    kv1[:, 1] = 1.0
    out1, _ = k.flash_attn_fwd(q, kv1[:, 0], kv1[:, 1], cu, T, None, True)
    assert float((out1.float() - 1.0).abs().max()) <= 8e-3
    # last 256 rows vs blocked fp32 reference
    qb = q[-256:].float()                                   # [256, hq, d]
    kf, vf = kv[:, 0, 0].float(), kv[:, 1, 0].float()       # [T, d]
    s = torch.einsum("qhd,td->hqt", qb, kf) / math.sqrt(d)
    rows = torch.arange(T - 256, T, device=dev)[:, None]
    s = s.masked_fill(torch.arange(T, device=dev)[None, :] > rows, float("-inf"))
    ref = torch.einsum("hqt,td->qhd", torch.softmax(s, -1), vf)
    close(out[-256:], ref.cpu(), 1.6e-2, 1e-2, "last 256 rows at T = 32768")
    # backward prefix property
    do = torch.zeros(T, hq, d, dtype=torch.bfloat16, device=dev)
    do[:P] = bf(torch.randn(P, hq, d, generator=g(102))).to(dev)
    dq, dk, dv = k.flash_attn_bwd(do, q, kv[:, 0], kv[:, 1], out, lse, cu, T, None, True)
    dq_p, dk_p, dv_p = k.flash_attn_bwd(do[:P].contiguous(), q[:P].contiguous(), kv[:P, 0], kv[:P, 1], out_p, lse_p, cu_p, P, None, True)
    assert torch.equal(dq[:P], dq_p) and float(dq[P:].float().abs().max()) == 0.0
    close(dk[:P], dk_p.cpu(), 1e-2, 1e-2, "dK prefix (the head split of the long problem may differ: fp32 partial order)", rms=FLASH_RMS)
    close(dv[:P], dv_p.cpu(), 1e-2, 1e-2, "dV prefix", rms=FLASH_RMS)
    assert float(dk[P:].float().abs().max()) == 0.0 and float(dv[P:].float().abs().max()) == 0.0


@pytest.mark.parametrize("rows,cols,view", [(4096, 1536, False), (37, 256, False), (5, 8, False), (130, 24, True), (64, 100, False), (1, 264, False)])
def test_colsum_bias_gradient(dev, rows, cols, view):
    """ie_colsum_bf16 (the bias gradient of linear_bias_wgrad / the InternLM-1 block's biases): the 16-byte row-split kernel (cols % 8 == 0)
    and the scalar fallback (cols = 100), ragged row counts, a strided column view."""
    x = bf(torch.randn(rows, cols + (16 if view else 0), generator=g(71)))
    xd = x.to(dev)
    xs = xd[:, 8 : 8 + cols] if view else xd
    ref = (x[:, 8 : 8 + cols] if view else x).float().sum(0)
    close(K().colsum(xs), ref, 8e-3, 2e-3 * math.sqrt(rows), f"colsum {rows}x{cols}")


@pytest.mark.parametrize("layout", ["nt", "nn", "tn"])
@pytest.mark.parametrize("Z,M,N,Kd", [(4, 264, 520, 192), (3, 512, 256, 1024), (2, 40, 72, 40)])
def test_gemm_strided_batch_equals_separate_products(dev, layout, Z, M, N, Kd):
    """ie_gemm_bf16_batched (the experts of a MoE layer in one launch): every product of the batch equals the single-product GEMM bit for
    bit, in the three operand layouts, with accumulate, on shapes that take the LDS-DMA kernels (K % 64 == 0) and the ragged fallback."""
    akm, bkm = layout[0] == "t", layout[1] == "n"
    A = bf(torch.randn((Z, Kd, M) if akm else (Z, M, Kd), generator=g(81))).to(dev)
    B = bf(torch.randn((Z, Kd, N) if bkm else (Z, N, Kd), generator=g(82))).to(dev)
    C0 = bf(torch.randn(Z, M, N, generator=g(83))).to(dev)
    out = torch.empty(Z, M, N, dtype=torch.bfloat16, device=dev)
    K().gemm_batched(A, B, out, akm, bkm)
    for z in range(Z):
        assert torch.equal(out[z], K().gemm(A[z], B[z], akm, bkm)), (layout, z)
    acc = C0.clone()
    K().gemm_batched(A, B, acc, akm, bkm, accumulate=True)
    for z in range(Z):
        one = C0[z].clone()
        K().gemm(A[z], B[z], akm, bkm, out=one, accumulate=True)
        assert torch.equal(acc[z], one), (layout, z, "accumulate")


@pytest.mark.parametrize("M,F,Kd", [(4096, 8192, 1024), (4000, 8192, 1024), (1000, 512, 200), (16384, 14336, 4096)], ids=["tiles", "ragged_rows", "two_launch_shape", "7b_shape"])
def test_ffn_products_with_the_swiglu_epilogues_equal_the_two_launch_path_bit_for_bit(dev, M, F, Kd):
    """ie_gemm_swiglu_fwd / _bwd (a7): the w1 | w3 forward product writes silu(gate) * up from its epilogue (a 256-column tile = 128 gate
    columns + the same 128 up columns), the w2 input-gradient product applies the gate's backward in its epilogue (d(act) never reaches
    memory).  Same arithmetic as swiglu_fwd_k / swiglu_bwd_k (shared device functions) on the same bf16-rounded products: the results must
    be IDENTICAL to the product followed by the elementwise kernel (which the oracle tests pin), also with a ragged last row tile, and
    shapes off the fused schedule must take the two-launch path by themselves."""
    Kk = K()
    L = Kk._L()
    g = torch.Generator(device=dev).manual_seed(11)
    x = torch.randn(M, Kd, device=dev, generator=g).to(torch.bfloat16)
    w13 = (torch.randn(2 * F, Kd, device=dev, generator=g) * 0.05).to(torch.bfloat16)
    w2 = (torch.randn(Kd, F, device=dev, generator=g) * 0.05).to(torch.bfloat16)
    dy = torch.randn(M, Kd, device=dev, generator=g).to(torch.bfloat16)
    out = {}
    for mode in (0, 3, 5):    # 0: two launches everywhere; 3: both products fused on every shape of the schedule; 5 (the default): the forward product, and the
        L.ie_tune_ffn_fuse(mode)   # input-gradient product where the persistent frame takes it (gemm_p5_k<true, 2>: whole tiles, more than one round)
        try:
            if mode == 3:
                fused = (int(L.ie_gemm_swiglu_is_fused(0, M, F, Kd)), int(L.ie_gemm_swiglu_is_fused(1, M, F, Kd)))
                assert fused == ((0, 0) if F == 512 else (1, 1)), fused
            if mode == 5:
                fused = (int(L.ie_gemm_swiglu_is_fused(0, M, F, Kd)), int(L.ie_gemm_swiglu_is_fused(1, M, F, Kd)))
                assert fused == ((0, 0) if F == 512 else (1, 1 if M % 256 == 0 else 0)), fused
            h13 = torch.full((M, 2 * F), 7.0, device=dev, dtype=torch.bfloat16)
            act = torch.full((M, F), 7.0, device=dev, dtype=torch.bfloat16)
            dh13 = torch.full((M, 2 * F), 7.0, device=dev, dtype=torch.bfloat16)
            dact = torch.empty(M, F, device=dev, dtype=torch.bfloat16)
            Kk.linear_swiglu_fwd(x, w13, h13, act)
            Kk.linear_dgrad_swiglu_bwd(dy, w2, h13, dh13, dact)
            torch.cuda.synchronize()
            out[mode] = (h13, act, dh13)
        finally:
            L.ie_tune_ffn_fuse(5)
    for other in (3, 5):
        for name, a, b in zip(("h13", "act", "dh13"), out[0], out[other]):
            assert torch.equal(a, b), f"mode {other} {name}: {int((a != b).sum())} of {a.numel()} elements differ, max |diff| {float((a.float() - b.float()).abs().max())}"
    # and the two-launch path is the plain product + the elementwise kernels
    ref = Kk.linear_fwd(x, w13)
    assert torch.equal(out[0][0], ref) and torch.equal(out[0][1], Kk.swiglu_fwd(ref[:, :F], ref[:, F:]))
