"""TEST INFRASTRUCTURE -- CPU oracle of the InternLM2 training step (model forward + loss), plain torch.

A restatement of the reference's own pure-torch path (`use_flash_attn=False`), op by op and dtype by
dtype, so that autograd on CPU yields the reference's gradients:
  PackedFlashLlama1D.forward        internlm/model/modeling_internlm2.py:966-1009
  PackedFlashLlamaLayer1D._forward  :684-740
  MHA._forward (unpacked)           :191-402        (MHA._packed_forward :404-478 computes the same values
                                                     per packed sequence; `cu_seqlens` selects that variant)
  FeedForward.forward               internlm/model/modules/mlp.py:82-86
  NaiveAMPModel.forward             internlm/core/naive_amp.py:137-159 (logits -> fp32)
  FlashGPTLMLoss.forward            internlm/model/losses/ce_loss.py:42-58
Pinned against the real reference by tests/test_oracle_golden.py (tests/golden/train_*.json).

Parameters are a dict name -> tensor using the reference's parameter names.
"""
import zlib

import numpy as np
import torch
import torch.nn.functional as F

from . import ops as O


def formula_init(name, shape):
    """Closed-form deterministic weights shared by the reference harness, this oracle and the HIP engine:
    N(0, 0.02) matrices, 1 + N(0, 0.05) norm gains, seeded by crc32(name), rounded to bf16-representable
    values (so fp32 and bf16 runs start from the same numbers)."""
    rs = np.random.RandomState(zlib.crc32(name.encode()) & 0x7FFFFFFF)
    if len(shape) == 1:
        w = 1.0 + 0.05 * rs.standard_normal(shape)
    else:
        w = 0.02 * rs.standard_normal(shape)
    return torch.from_numpy(w.astype(np.float32)).to(torch.bfloat16).float()


def moe_formula_init(name, shape):
    """formula_init for the INTERNLM_MoE family (modeling_moe.py): linear biases and the fp32 gate weight are small normal numbers
    instead of the 1 + N(0, 0.05) that formula_init gives every 1-D tensor (a norm gain)."""
    rs = np.random.RandomState(zlib.crc32(name.encode()) & 0x7FFFFFFF)
    if name.endswith(".bias"):
        return torch.from_numpy((0.02 * rs.standard_normal(shape)).astype(np.float32)).to(torch.bfloat16).float()
    if name.endswith("gate.wg.weight"):
        return torch.from_numpy((0.1 * rs.standard_normal(shape)).astype(np.float32)).to(torch.bfloat16).float()
    return formula_init(name, shape)


def param_shapes(mc):
    """name -> shape, in the reference's naming (PackedFlashLlama1D.named_parameters())."""
    h, f, v = mc.hidden_size, mc.ffn_dim, mc.vocab_size
    out = {"tok_embeddings.weight": (v, h)}
    for l in range(mc.num_layers):
        p = f"layers.{l}."
        if getattr(mc, "model_type", "INTERNLM2_PUBLIC") == "LLAMA2":  # modeling_llama.py:126-148: three separate projections
            out[p + "attention.wq.weight"] = (mc.num_attention_heads * mc.head_dim, h)
            out[p + "attention.wk.weight"] = (mc.num_kv_attention_heads * mc.head_dim, h)
            out[p + "attention.wv.weight"] = (mc.num_kv_attention_heads * mc.head_dim, h)
        else:
            out[p + "attention.wqkv.weight"] = (mc.qkv_dim, h)
        out[p + "attention.wo.weight"] = (h, h)
        out[p + "feed_forward.w1.weight"] = (f, h)
        out[p + "feed_forward.w3.weight"] = (f, h)
        out[p + "feed_forward.w2.weight"] = (h, f)
        out[p + "attention_norm.weight"] = (h,)
        out[p + "ffn_norm.weight"] = (h,)
    out["norm.weight"] = (h,)
    out["output.weight"] = (v, h)
    return out


def build_params(mc, dtype, init_fn=formula_init):
    return {n: init_fn(n, s).to(dtype).requires_grad_(True) for n, s in param_shapes(mc).items()}


def forward_logits(params, mc, input_ids, indexes=None, cu_seqlens=None):
    """input_ids [S] (one packed row).  Returns fp32 logits [S, V]."""
    p = params
    dt = p["tok_embeddings.weight"].dtype
    S = input_ids.shape[0]
    hkv, qpk, d = mc.num_kv_attention_heads, mc.q_per_kv, mc.head_dim
    if indexes is None:
        indexes = torch.arange(S)
    cos, sin = O.rotary_cos_sin(int(indexes.max()) + 1, d, mc.rope_base, dt)
    if cu_seqlens is None:
        cu_seqlens = torch.tensor([0, S], dtype=torch.int32)
    h = O.embedding(input_ids, p["tok_embeddings.weight"])   # (F.embedding; its backward in the CPU kernel's or the accelerator kernel's arithmetic, ops.py)
    egs = float(getattr(mc, "embed_grad_scale", 1.0))
    if egs != 1:   # modeling_internlm2.py:970-973: the value is (nearly) unchanged, the gradient into the embedding scaled by egs
        h = egs * h + (1 - egs) * h.detach()
    for l in range(mc.num_layers):
        pre = f"layers.{l}."
        residual = h
        x = O.rms_norm(residual.to(p[pre + "attention_norm.weight"].dtype), p[pre + "attention_norm.weight"], mc.layer_norm_epsilon)
        if getattr(mc, "model_type", "INTERNLM2_PUBLIC") == "LLAMA2":
            # modeling_llama.py:412-426: q, k, v projected separately, heads contiguous; q head j attends kv head j // q_per_kv.
            # Arranged as InternLM2's [kv group][q_per_kv q heads, k, v] the rest of the path is shared.
            qh = F.linear(x, p[pre + "attention.wq.weight"]).reshape(S, hkv, qpk, d)
            kh = F.linear(x, p[pre + "attention.wk.weight"]).reshape(S, hkv, 1, d)
            vh = F.linear(x, p[pre + "attention.wv.weight"]).reshape(S, hkv, 1, d)
            qkv = torch.cat([qh, kh, vh], dim=2).reshape(S, -1)
        else:
            qkv = F.linear(x, p[pre + "attention.wqkv.weight"])
        q, kv = O.qkv_split_rotary(qkv, cos, sin, indexes, hkv, qpk, d, interleaved=not mc.adapt_hf)
        ctx = O.attention_varlen(q, kv, cu_seqlens, causal=True)
        attn_out = F.linear(ctx.reshape(S, -1), p[pre + "attention.wo.weight"])
        residual = attn_out + residual
        x = O.rms_norm(residual.to(torch.float32), p[pre + "ffn_norm.weight"], mc.layer_norm_epsilon)
        a = F.linear(x, p[pre + "feed_forward.w1.weight"])
        b = F.linear(x, p[pre + "feed_forward.w3.weight"])
        ffn = F.linear(O.swiglu(a, b), p[pre + "feed_forward.w2.weight"])
        h = ffn + residual
    x = O.rms_norm(h.float(), p["norm.weight"], mc.layer_norm_epsilon)
    w = p["output.weight"]
    if egs != 1:   # ScaleColumnParallelLinear.forward (ops/linear.py:125-128): weight_scale = embed_grad_scale
        w = w * egs + (1 - egs) * w.detach()
    if getattr(mc, "norm_head", False):   # ops/linear.py:129-136 (training branch)
        w = F.normalize(w)
    logits = F.linear(x, w)
    return logits.float()


def micro_loss(params, mc, input_ids, labels, indexes=None, cu_seqlens=None, label_smoothing=0.0):
    return O.cross_entropy(forward_logits(params, mc, input_ids, indexes, cu_seqlens), labels, label_smoothing)
