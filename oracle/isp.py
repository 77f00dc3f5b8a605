"""TEST INFRASTRUCTURE -- CPU oracle of the reference's ISP mode (parallel.tensor = dict(size=sp, mode="isp"), parallel.weight = dict(size=wp);
configs/7B_isp_sft.py) restated in ONE process.  Never imported by the product.

What an sp-rank ISP run of the reference computes, as its code reads and as its two-process CPU runs show
(tests/golden/train_isp2_*_rank*.json = InternLM2 blocks, train_isp2v1_*_rank*.json = InternLM-1 blocks; make_golden.py --run-mp):

  * forward / backward: the model of one rank on the whole sequence -- the weights are whole when a linear runs (ISPLinear all-gathers its rows over
    the WEIGHT group, model/utils.py:466-586), every rank works on S / sp tokens of a micro-batch (modules/embedding.py:57-58) and the loss is the mean over
    the gathered sequence (ops/linear.py:146-153).  The reference's CPU-runnable path is the UNPACKED one, where a rank's rotary positions restart at
    its chunk (embedding.py:329-365 `_eval_forward` / `_single_eval_forward` with seqlen_offset 0 on the local chunk), and
      - InternLM2 (modeling_internlm2.py:215-229): attention goes through `self.inner_cross_attn` of the LOCAL chunk without the DistributedAttention
        exchange having any effect on the CPU path's [b, s / sp] layout -> attention local to each chunk = the varlen model with cu_seqlens at the
        chunk boundaries;
      - InternLM-1 (multi_head_attention.py:394, :634-660): `self.inner_attn(qkv)` IS DistributedAttention(SelfAttention): the qkv-packed all-to-all
        gathers the sequence and scatters the heads -> causal attention over the WHOLE sequence (with the restarting positions).
      - InternLM2 WITH the exchange (round 6, train_isp2u_*): tests/golden/make_golden.py wraps the block's own CrossAttention in the reference's own
        DistributedAttention (harness-side, the keywords of the packed path: q [b, S/sp, h, d] <-> [b, S, h/sp, d], kv [b, S/sp, 2, hkv, d] <-> [b, S, 2, hkv/sp, d]),
        so the reference executes the Ulysses exchange of a GQA block on its CPU path -> causal attention over the WHOLE sequence, restarting positions.
  * gradient rule (engine._apply_isp_grad_rule): ISPLinear weights AND biases are reduce-scattered with AVG over the weight group (model/utils.py:556-561) and
    all-reduced AVG over WEIGHT_DATA (hybrid_zero_optim.py:98,169); norm weights AVG over the weight group (:318-324) -- per-rank gradients that each cover
    1 / sp of the tokens, so the result is 1 / sp of the mean gradient; embedding and head (the "embed_head" group, train/utils.py:42-43, reduced over DATA)
    see the gathered sequence: the full mean gradient.
  * optimizer groups (train/utils.py:11-80): "0_default" (bf16 layer + norm parameters), "1_embed_head", "2_fp32" (fp32 parameters: EVERYTHING but
    embedding / head in an fp32 run).  Every group has its own norm and is unscaled / clipped by its OWN factor (hybrid_zero_optim.py:760-779,863-876); an fp32
    run neither unscales nor clips (:773).
"""
import math

import torch

from . import ops as O


def isp_positions(seq_len, sp, family, ulysses=False):
    """(indexes [S], cu_seqlens) under which the single-process varlen model equals the reference's CPU-runnable sp-rank ISP forward.
    ulysses: the InternLM2 runs whose CrossAttention the harness wrapped in the reference's DistributedAttention (make_golden.py `ulysses=True`,
    train_isp2u_*): attention over the gathered sequence as for the InternLM-1 block, positions still restarting per rank."""
    chunk = seq_len // sp
    idx = torch.arange(chunk, dtype=torch.int64).repeat(sp)
    if family == "INTERNLM" or ulysses:
        return idx, torch.tensor([0, seq_len], dtype=torch.int32)
    return idx, (torch.arange(sp + 1, dtype=torch.int32) * chunk)


class OracleISPTrainer:
    """base: an OracleTrainer (InternLM2 / LLAMA2) or an OracleMoETrainer of the dense InternLM-1 model -- its parameters, fp32 state, scaler and
    schedules are used; this class runs the step with the ISP positions, the gradient rule and the two clipping groups."""

    def __init__(self, base, sp, ulysses=False):
        self.base, self.sp, self.ulysses = base, sp, ulysses
        self.family = "INTERNLM" if "embedding.weight" in base.params else "INTERNLM2"
        self.embed_head = ("embedding.weight", "head.weight") if self.family == "INTERNLM" else ("tok_embeddings.weight", "output.weight")

    def _loss(self, ids, labels, idx, cu):
        b = self.base
        if self.family == "INTERNLM":
            from .moe_model import forward_logits

            logits, _ = forward_logits(b.params, b.mc, ids, None, idx, cu)
        else:
            from .model import forward_logits

            logits = forward_logits(b.params, b.mc, ids, idx, cu)
        return O.cross_entropy(logits, labels, b.tc.label_smoothing)

    def train_step(self, batch, labels):
        b, sp = self.base, self.sp
        tc = b.tc
        for p in b.params.values():
            p.grad = None
        M, S = batch["input_ids"].shape
        idx, cu = isp_positions(S, sp, self.family, self.ulysses)
        total = 0.0
        for i in range(M):
            loss = self._loss(batch["input_ids"][i], labels[i], idx, cu) / M
            total += float(loss.detach())
            (b.scaler.scale * loss).backward()
        groups = {"0_default": [], "1_embed_head": [], "2_fp32": []}
        with torch.no_grad():
            for n in b.names:
                if n in self.embed_head:
                    groups["1_embed_head"].append(n)
                else:
                    b.params[n].grad.div_(sp)     # THE RULE (exact: a power of two)
                    groups["2_fp32" if b.dtype == torch.float32 else "0_default"].append(n)
        sq = {}
        for g, names in groups.items():
            acc = 0.0
            for n in names:
                acc = acc + torch.norm(b.params[n].grad.float(), 2.0) ** 2.0
            sq[g] = float(acc)
        found_inf, found_nan = any(math.isinf(v) for v in sq.values()), any(math.isnan(v) for v in sq.values())
        loss_scale = b.scaler.scale
        if b.dtype != torch.float32:
            b.scaler.update(found_inf)
        if found_inf or found_nan:
            return {"loss": total, "grad_norm": {g: (-1.0 if found_inf else -2.0) for g in groups}, "ok": False, "loss_scale": b.scaler.scale}
        lr, beta2 = b._lr(), b._beta2()
        b.adam_step += 1
        norms = {}
        with torch.no_grad():
            for g, names in groups.items():
                gnorm = sq[g] ** 0.5
                norms[g] = gnorm / loss_scale
                inv = 1.0
                if b.dtype != torch.float32 and tc.clip_grad_norm > 0:
                    inv = 1.0 / O.unscale_clip_factor(gnorm, loss_scale, tc.clip_grad_norm)
                for n in names:
                    gr = b.params[n].grad.float()
                    gr.mul_(inv)
                    O.adamw_step(b.master[n], gr, b.m[n], b.v[n], b.adam_step, lr, tc.adam_beta1, beta2, tc.adam_eps, tc.weight_decay)
                    b.params[n].copy_(b.master[n])
        b.k += 1
        b.beta2_iter += 1
        return {"loss": total, "grad_norm": norms, "ok": True, "loss_scale": b.scaler.scale, "lr": lr}
