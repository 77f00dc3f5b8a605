"""GPU tests of drop-in boundary #2: the shim modules registered under the reference's third-party import names
(flash_attn, rotary_emb, fused_dense_lib, apex, amp_C) behave like the ops the reference calls, checked against
the CPU oracle.  Call patterns mirror the reference's call sites (cited inline)."""
import math

import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import ops as O  # noqa: E402


@pytest.fixture(scope="module", autouse=True)
def _install():
    import internevo_amd.plugin as plugin

    plugin.install(force=True)


def g(seed):
    return torch.Generator().manual_seed(seed)


def bf(t):
    return t.to(torch.bfloat16)


def close(got, ref, rtol, atol, what):
    got, ref = got.detach().float().cpu(), ref.detach().float().cpu()
    err = (got - ref).abs()
    bad = err > atol + rtol * ref.abs()
    assert not bad.any(), f"{what}: {int(bad.sum())}/{bad.numel()} mismatches, max abs err {err.max():.3e}"


def test_import_sites_resolve():
    import amp_C  # noqa: F401
    import fused_dense_lib  # noqa: F401
    import rotary_emb  # noqa: F401
    from apex.multi_tensor_apply import multi_tensor_applier  # noqa: F401
    from apex.normalization.fused_layer_norm import MixedFusedRMSNorm  # noqa: F401
    from flash_attn import flash_attn_varlen_kvpacked_func  # noqa: F401
    from flash_attn.flash_attn_interface import FlashAttnVarlenKVPackedFunc  # noqa: F401
    from flash_attn.losses.cross_entropy import CrossEntropyLoss  # noqa: F401
    from flash_attn.modules.embedding import ParallelGPT2Embeddings  # noqa: F401
    from flash_attn.modules.mha import FlashCrossAttention, FlashSelfAttention  # noqa: F401
    from flash_attn.modules.mlp import ParallelFusedMLP  # noqa: F401
    from flash_attn.ops.layer_norm import dropout_add_layer_norm  # noqa: F401


def test_flash_attn_varlen_kvpacked_func_autograd(dev):
    # modeling_internlm2.py:446-468
    from flash_attn import flash_attn_varlen_kvpacked_func

    lens = [70, 186]
    T, H, Hk, D = sum(lens), 8, 2, 128
    cu = torch.tensor([0, 70, 256], dtype=torch.int32)
    q = bf(torch.randn(T, H, D, generator=g(1)))
    kv = bf(torch.randn(T, 2, Hk, D, generator=g(2)))
    do = bf(torch.randn(T, H, D, generator=g(3)))
    qd, kvd = q.to(dev).requires_grad_(True), kv.to(dev).requires_grad_(True)
    out = flash_attn_varlen_kvpacked_func(q=qd, kv=kvd, cu_seqlens_q=cu.to(dev), cu_seqlens_k=cu.to(dev), max_seqlen_q=186, max_seqlen_k=186,
                                          dropout_p=0.0, softmax_scale=None, causal=True)
    out.backward(do.to(dev))
    q32, kv32 = q.float().requires_grad_(True), kv.float().requires_grad_(True)
    ref = O.attention_varlen(q32, kv32, cu, True)
    (ref * do.float()).sum().backward()
    close(out, ref, 1.6e-2, 2e-2, "kvpacked fwd")
    close(qd.grad, q32.grad, 2e-2, 3e-2, "kvpacked dq")
    close(kvd.grad, kv32.grad, 2e-2, 3e-2, "kvpacked dkv")


def test_flash_self_and_cross_attention_modules(dev):
    # multi_head_attention.py:381-392,646-659 (v1 qkv-packed) and modeling_internlm2.py:158-165 (padded kv-packed)
    from flash_attn.modules.mha import FlashCrossAttention, FlashSelfAttention

    B, S, H, D = 2, 96, 4, 64
    qkv = bf(torch.randn(B, S, 3, H, D, generator=g(4)))
    ref = O.attention_dense(qkv[:, :, 0].float(), torch.stack([qkv[:, :, 1], qkv[:, :, 2]], 2).float(), True)
    out = FlashSelfAttention(causal=True)(qkv.to(dev))
    close(out, ref, 1.6e-2, 2e-2, "FlashSelfAttention padded")
    cu = torch.tensor([0, S, 2 * S], dtype=torch.int32, device=dev)
    out2 = FlashSelfAttention(causal=True)(qkv.reshape(B * S, 3, H, D).to(dev), cu_seqlens=cu, max_seqlen=S)
    close(out2.reshape(B, S, H, D), ref, 1.6e-2, 2e-2, "FlashSelfAttention packed")
    q = bf(torch.randn(B, S, 8, D, generator=g(5)))
    kv = bf(torch.randn(B, S, 2, 2, D, generator=g(6)))
    out3 = FlashCrossAttention(causal=True)(q.to(dev), kv.to(dev))
    close(out3, O.attention_dense(q.float(), kv.float(), True), 1.6e-2, 2e-2, "FlashCrossAttention GQA")


def test_cross_entropy_loss_module(dev):
    # losses/ce_loss.py:31-36 : reduction="mean", inplace_backward=True on the fp32 logits NaiveAMP produces
    from flash_attn.losses.cross_entropy import CrossEntropyLoss

    rows, V = 48, 1000
    logits = torch.randn(rows, V, generator=g(7)) * 2
    labels = torch.randint(0, V, (rows,), generator=g(8))
    labels[::7] = -100
    l32 = logits.clone().requires_grad_(True)
    ref = O.cross_entropy(l32, labels)
    (ref * 65536.0 / 4).backward()
    ld = logits.to(dev).requires_grad_(True)
    x = ld * 1.0  # non-leaf, so the in-place backward may overwrite it (as NaiveAMP's .float() copy is)
    loss = CrossEntropyLoss(reduction="mean", inplace_backward=True, process_group=None, label_smoothing=0)(x, labels.to(dev))
    (loss * 65536.0 / 4).backward()
    close(loss.reshape(1), ref.reshape(1), 2e-5, 1e-5, "CE loss")
    close(ld.grad, l32.grad, 2e-4, 1e-4, "CE grad")


def test_mixed_fused_rmsnorm_module(dev):
    # model/utils.py:662-675; called with fp32 input + bf16 weight at modeling_internlm2.py:725,1002
    from apex.normalization.fused_layer_norm import MixedFusedRMSNorm

    norm = MixedFusedRMSNorm(512, eps=1e-5).to(dev).to(torch.bfloat16)
    with torch.no_grad():
        norm.weight.copy_(bf(1 + 0.1 * torch.randn(512, generator=g(9))))
    w = norm.weight.detach().cpu()
    for xdt in (torch.float32, torch.bfloat16):
        x = (torch.randn(3, 40, 512, generator=g(10)) * 2).to(xdt)
        xd = x.to(dev).requires_grad_(True)
        y = norm(xd)
        assert y.dtype == torch.bfloat16 and y.shape == x.shape
        close(y, O.rms_norm(x, w, 1e-5), 1.6e-2, 1e-6, f"MixedFusedRMSNorm fwd {xdt}")
        dy = bf(torch.randn(3, 40, 512, generator=g(11)))
        norm.weight.grad = None
        y.backward(dy.to(dev))
        dx_ref, dw_ref = O.rms_norm_bwd_fp32(dy, x, w, 1e-5)
        assert xd.grad.dtype == xdt
        close(xd.grad, dx_ref, 1.6e-2, 2e-2, "MixedFusedRMSNorm dx")
        close(norm.weight.grad, dw_ref, 1.6e-2, 0.3, "MixedFusedRMSNorm dw")


def test_rotary_emb_apply_rotary_like_the_reference(dev):
    # embedding.py:104-120: x_ro.chunk(2, dim=-1) views, cos/sin rearranged "s d -> s 1 d", out = empty_like(x)
    import rotary_emb

    x = bf(torch.randn(1, 50, 6, 128, generator=g(12))).to(dev)
    cos, sin = O.rotary_cos_sin(50, 128)
    out = torch.empty_like(x)
    x1, x2 = x.chunk(2, dim=-1)
    o1, o2 = out.chunk(2, dim=-1)
    rotary_emb.apply_rotary(x1, x2, cos.to(dev)[:, None, :], sin.to(dev)[:, None, :], o1, o2, False)
    close(out, O.apply_rotary_emb(x.cpu(), cos, sin), 8e-3, 1e-6, "rotary_emb.apply_rotary")
    # packed qkv form, in place, conj (ApplyRotaryEmbQKV_.backward, embedding.py:239-256)
    dqkv = bf(torch.randn(70, 3, 4, 128, generator=g(13))).to(dev)
    cos70, sin70 = O.rotary_cos_sin(70, 128)
    want1, want2 = O.apply_rotary(dqkv[:, 0, :, :64].cpu(), dqkv[:, 0, :, 64:].cpu(), cos70[:, None, :], sin70[:, None, :], True)
    dq_ro = dqkv[:, 0, :, :128]
    dq1, dq2 = dq_ro.chunk(2, dim=-1)
    rotary_emb.apply_rotary(dq1, dq2, cos70.to(dev)[:, None, :], sin70.to(dev)[:, None, :], dq1, dq2, True)
    close(dqkv[:, 0, :, :64], want1, 8e-3, 1e-6, "apply_rotary in-place conj (1)")
    close(dqkv[:, 0, :, 64:], want2, 8e-3, 1e-6, "apply_rotary in-place conj (2)")


def test_fused_dense_lib_linear_bias_wgrad(dev):
    # model/utils.py:293-299
    import fused_dense_lib

    x = bf(torch.randn(384, 512, generator=g(14)))
    dy = bf(torch.randn(384, 256, generator=g(15)))
    dw, db = fused_dense_lib.linear_bias_wgrad(x.to(dev), dy.to(dev), True)
    close(dw, dy.float().t() @ x.float(), 8e-3, 5e-2, "linear_bias_wgrad dW")
    close(db, dy.float().sum(0), 8e-3, 5e-2, "linear_bias_wgrad db")
    dw2, db2 = fused_dense_lib.linear_bias_wgrad(x.to(dev), dy.to(dev), False)
    assert db2 is None and torch.equal(dw2, dw)


def test_amp_c_multi_tensor_l2norm(dev):
    # solver/optimizer/utils.py:191-204
    import amp_C
    from apex.multi_tensor_apply import multi_tensor_applier

    grads = [torch.randn(n, generator=g(20 + i)).to(dev) for i, n in enumerate([3, 1000, 65537])]
    buf = torch.tensor([0], device=dev, dtype=torch.int32)
    norm, _ = multi_tensor_applier(amp_C.multi_tensor_l2norm, buf, [grads], False)
    ref = O.l2_norm([t.cpu() for t in grads])
    close(norm, ref.reshape(1), 1e-5, 0, "multi_tensor_l2norm")
    norm2, per = multi_tensor_applier(amp_C.multi_tensor_l2norm, buf, [grads], True)
    close(per, torch.stack([t.cpu().norm() for t in grads]), 1e-5, 0, "per-tensor norms")
