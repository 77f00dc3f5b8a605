"""The drop-in route end to end on the GPU: a 2-layer InternLM2 model built from torch autograd + ONLY the plugin shims
(`internevo_amd.plugin`: flash_attn, rotary_emb, fused_dense_lib, apex, flash_attn.losses), composed the way the reference composes
its third-party ops on the packed path, trained for one forward + backward, against the native engine on the same weights and batch.

`/root/reference` does not exist on the GPU box, so the reference's own modules cannot run there; this is the closest thing: every
call below mirrors a reference call site (cited), with the reference's own Python glue restated in a few lines each:
  PackedFlashLlama1D.forward         internlm/model/modeling_internlm2.py:966-1009   embedding -> layers -> norm -> head
  PackedFlashLlamaLayer1D._forward   :684-740      pre-norm blocks; attention_norm sees the residual in the weight dtype, ffn_norm in fp32
  MHA._packed_forward                :404-478      wqkv -> "(h gs d)" split -> even/odd shuffle -> rotary(indexes) -> kv pack ->
                                                   flash_attn_varlen_kvpacked_func(q, kv, cu, cu, max, max, 0.0, scale, causal) -> wo
  ApplyRotaryEmb                     internlm/model/modules/embedding.py:91-160     rotary_emb.apply_rotary(x1, x2, cos, sin, o1, o2, conj)
  FusedDenseFunc                     internlm/model/utils.py:236-330                F.linear forward, fused_dense_lib.linear_bias_wgrad backward
  FeedForward / Silu                 internlm/model/modules/mlp.py:82-86, model/utils.py:684-688
  RMSNorm                            apex.normalization.fused_layer_norm.MixedFusedRMSNorm (model/utils.py:668)
  FlashGPTLMLoss                     internlm/model/losses/ce_loss.py:26-58         flash_attn.losses.cross_entropy.CrossEntropyLoss("mean", inplace_backward)
Checked: the loss, and every parameter gradient against the engine's flat gradient buffer (same loss scale)."""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module", autouse=True)
def _install():
    import internevo_amd.plugin as plugin

    plugin.install(force=True)


class _Rotary(torch.autograd.Function):
    """ApplyRotaryEmb (embedding.py:91-160): NeoX halves, cos / sin already gathered by `indexes` ([S, d/2])."""

    @staticmethod
    def forward(ctx, x, cos, sin):
        import rotary_emb

        x1, x2 = x.chunk(2, dim=-1)
        out = torch.empty_like(x)
        o1, o2 = out.chunk(2, dim=-1)
        rotary_emb.apply_rotary(x1, x2, cos[:, None, :], sin[:, None, :], o1, o2, False)
        ctx.save_for_backward(cos, sin)
        return out

    @staticmethod
    def backward(ctx, do):
        import rotary_emb

        cos, sin = ctx.saved_tensors
        do = do.contiguous()
        d1, d2 = do.chunk(2, dim=-1)
        dx = torch.empty_like(do)
        dx1, dx2 = dx.chunk(2, dim=-1)
        rotary_emb.apply_rotary(d1, d2, cos[:, None, :], sin[:, None, :], dx1, dx2, True)
        return dx, None, None


class _Dense(torch.autograd.Function):
    """FusedDenseFunc without tensor parallelism (model/utils.py:236-330)."""

    @staticmethod
    def forward(ctx, x, weight):
        ctx.save_for_backward(x, weight)
        return F.linear(x, weight)

    @staticmethod
    def backward(ctx, dy):
        import fused_dense_lib

        x, weight = ctx.saved_tensors
        dy = dy.contiguous()
        dx = F.linear(dy, weight.t())
        dw, _ = fused_dense_lib.linear_bias_wgrad(x.reshape(-1, x.shape[-1]), dy.reshape(-1, dy.shape[-1]), False)
        return dx, dw


def _dropin_loss(params, mc, ids, labels, indexes, cu, max_seqlen):
    from apex.normalization.fused_layer_norm import MixedFusedRMSNorm
    from flash_attn import flash_attn_varlen_kvpacked_func
    from flash_attn.losses.cross_entropy import CrossEntropyLoss

    from oracle import ops as O

    dev = ids.device
    hkv, qpk, d, S = mc.num_kv_attention_heads, mc.q_per_kv, mc.head_dim, ids.shape[0]

    def norm(name, x):
        m = MixedFusedRMSNorm(mc.hidden_size, eps=mc.layer_norm_epsilon)
        del m.weight
        m.weight = params[name]                       # the test's own leaf tensor, so that .grad lands on params[name]
        return m(x)

    cos, sin = O.rotary_cos_sin(int(indexes.max()) + 1, d, mc.rope_base, torch.bfloat16)   # RotaryEmbedding._update_cos_sin_cache
    cos, sin = cos.to(dev)[indexes], sin.to(dev)[indexes]                                   # _single_forward: cached[indexes]
    h = F.embedding(ids, params["tok_embeddings.weight"])
    for l in range(mc.num_layers):
        pre = f"layers.{l}."
        residual = h
        x = norm(pre + "attention_norm.weight", residual.to(params[pre + "attention_norm.weight"].dtype))
        qkv = _Dense.apply(x, params[pre + "attention.wqkv.weight"]).reshape(S, hkv, qpk + 2, d)    # "t (h gs d) -> t h gs d"
        q, k, v = qkv[:, :, :qpk, :].reshape(S, hkv * qpk, d), qkv[:, :, -2, :], qkv[:, :, -1, :]
        if not mc.adapt_hf:   # rot_embed_HF_impl False: even / odd shuffle before the NeoX rotation (:424-426)
            q = torch.cat([q[..., ::2], q[..., 1::2]], dim=-1)
            k = torch.cat([k[..., ::2], k[..., 1::2]], dim=-1)
        q = _Rotary.apply(q.contiguous(), cos, sin)
        k = _Rotary.apply(k.contiguous(), cos, sin)
        kv = torch.concat([k.unsqueeze(1), v.unsqueeze(1)], dim=1)
        ctx = flash_attn_varlen_kvpacked_func(q, kv, cu, cu, max_seqlen, max_seqlen, 0.0, None, causal=True)
        attn = _Dense.apply(ctx.reshape(S, -1), params[pre + "attention.wo.weight"])
        residual = attn + residual
        x = norm(pre + "ffn_norm.weight", residual.to(torch.float32))
        a = _Dense.apply(x, params[pre + "feed_forward.w1.weight"])
        b = _Dense.apply(x, params[pre + "feed_forward.w3.weight"])
        h = _Dense.apply(F.silu(a) * b, params[pre + "feed_forward.w2.weight"]) + residual
    x = norm("norm.weight", h.float())
    logits = _Dense.apply(x, params["output.weight"]).float()                                # NaiveAMPModel casts the output to fp32
    return CrossEntropyLoss(reduction="mean", inplace_backward=True)(logits, labels)


@pytest.mark.parametrize("adapt_hf", [False, True])
def test_model_built_from_the_plugin_shims_matches_the_engine(dev, adapt_hf):
    from internevo_amd.config import tiny
    from internevo_amd.data import SyntheticLoader
    from internevo_amd.engine import InternLM2Engine
    from oracle.model import formula_init

    cfg = tiny(256, 2, 4, 2, 512, 384, 1, 1e-3, 4)
    cfg.model.adapt_hf = adapt_hf
    cfg.train.fixed_random_dataset_seqlen = False            # ragged packed sequences: several cu_seqlens segments per row
    eng = InternLM2Engine(cfg, dev, init_fn=formula_init)
    batch, labels = next(iter(SyntheticLoader(384, 1, 1, False, 4000)))
    assert len(batch["cu_seqlens"][0]) > 2, "the batch should hold more than one packed sequence"

    params = {n: p.detach().clone().requires_grad_(True) for n, p in eng.named_parameters()}
    loss_e = eng.forward_backward(batch, labels)
    scale = float(eng.read_state().loss_scale)

    ids, lab = batch["input_ids"][0].to(dev), labels[0].to(dev)
    idx, cu = batch["indexes"][0].to(dev), batch["cu_seqlens"][0].to(dev)
    max_seqlen = int((batch["cu_seqlens"][0][1:] - batch["cu_seqlens"][0][:-1]).max())
    loss_p = _dropin_loss(params, cfg.model, ids, lab, idx, cu, max_seqlen)
    (loss_p * scale).backward()

    le, lp = float(loss_e), float(loss_p.detach())
    print(f"[drop-in] loss engine {le:.6f} plugin-composed {lp:.6f}")
    assert abs(le - lp) <= 1e-3 * abs(lp)
    worst = 0.0
    for n, p in params.items():
        ge, gp = eng.g[n].float(), p.grad.float()
        rel = float((ge - gp).norm() / gp.norm().clamp_min(1e-12))
        worst = max(worst, rel)
        # both sides are bf16 gradients of the same function computed in different summation orders
        assert rel <= 2e-2, f"{n}: relative gradient difference {rel:.3e}"
    print(f"[drop-in] worst relative gradient difference {worst:.3e}")
