"""fp8 (OCP e4m3) operands for the expert products (the KERNELS: round 5 removed the engine route built on them in round 4 -- it lost 3 % in the
MoE step, profiles/r04_fp8_expert_gemm.md -- and kept the product, its quantiser and these tests)
fp8 (OCP e4m3) operands (include/internevo_hip.h: ie_fp8_amax, ie_fp8_quantize, ie_gemm_fp8).
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
