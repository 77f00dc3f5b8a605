"""TEST INFRASTRUCTURE -- CPU oracle of the InternLM-1 model families -- INTERNLM_MoE (BASELINE configs[4], configs/7B_MoE4_sft.py) and, with
num_experts = 1, the dense INTERNLM model (modeling_internlm.py, configs/7B_sft.py: the same block with a plain SwiGLU FeedForward) -- and of
their training step.  Restates in plain torch:
  PackedFlashInternLm1D / PackedFlashBaseLayer1D         internlm/model/modeling_moe.py:33-257,262-444 (embedding -> blocks -> norm -> head;
      block = norm1 -> MHA -> residual -> norm2 -> MoE -> residual, norms fed the fp32-cast residual)
  MHA (InternLM-1 attention: packed Wqkv "(three h d)" with bias, NeoX rotary on q / k, out_proj with bias)
                                                          internlm/model/modules/multi_head_attention.py:298-478
  MoE / GShardMOELayer                                    oracle/moe.py (pinned on the real layer)
  the moe loss (sum of the layers' l_aux * loss.moe_loss_coeff, divided by the accumulation steps, added to the loss)
                                                          internlm/core/scheduler/no_pipeline_scheduler.py:120-145
  parameter groups default / fp32 (the gates) / moe (the experts), each with its OWN gradient norm and its OWN clipping factor
                                                          internlm/train/utils.py:25-80, solver/optimizer/hybrid_zero_optim.py:760-779,863-876
Pinned by tests/golden/train_moe_{fp32,bf16}.json = the unmodified reference training loop on CPU (make_golden.py --run moe_*).
"""
import math

import torch
import torch.nn.functional as F

from . import moe as MO
from . import ops as O
from .model import moe_formula_init
from .step import OracleTrainer


def ffn_dim(mc):
    f = int(mc.hidden_size * mc.mlp_ratio)
    return mc.multiple_of * ((f + mc.multiple_of - 1) // mc.multiple_of)


def param_shapes(mc):
    """name -> shape in the reference's naming (PackedFlashInternLm1D.named_parameters() order)."""
    h, f, v, E = mc.hidden_size, ffn_dim(mc), mc.vocab_size, mc.num_experts
    out = {"embedding.weight": (v, h)}
    for l in range(mc.num_layers):
        p = f"blocks.{l}."
        out[p + "mixer.Wqkv.weight"], out[p + "mixer.Wqkv.bias"] = (3 * h, h), (3 * h,)
        out[p + "mixer.out_proj.weight"], out[p + "mixer.out_proj.bias"] = (h, h), (h,)
        out[p + "norm1.weight"], out[p + "norm2.weight"] = (h,), (h,)
        if E <= 1:   # the dense InternLM-1 model (modeling_internlm.py:118-131; modeling_moe.py with num_experts = 1): a plain SwiGLU FeedForward
            out[p + "mlp.w1.weight"], out[p + "mlp.w2.weight"], out[p + "mlp.w3.weight"] = (f, h), (h, f), (f, h)
            continue
        out[p + "mlp.moe_layer.gate.wg.weight"] = (E, h)
        for e in range(E):
            q = p + f"mlp.moe_layer.experts.wrapped_experts.{e}."
            out[q + "w1.weight"], out[q + "w2.weight"], out[q + "w3.weight"] = (f, h), (h, f), (f, h)
    out["norm.weight"] = (h,)
    out["head.weight"] = (v, h)
    return out


def group_of(name):
    """create_param_groups (train/utils.py:25-80): fp32 parameters (the gates, set_fp32_attr_to_module) and expert parameters get their own
    optimizer groups."""
    if name.endswith("gate.wg.weight"):
        return "1_fp32"
    if ".experts." in name:
        return "2_moe_ep_size_1"
    return "0_default"


def forward_logits(params, mc, input_ids, noise_fn, indexes=None, cu_seqlens=None, routes=None, forced=None):
    """input_ids [S] (one packed row / one micro-batch).  noise_fn(layer) -> fp32 [S, E] Gumbel noise of that layer's gate.
    Returns (fp32 logits [S, V], [l_aux per layer])."""
    p = params
    S = input_ids.shape[0]
    if indexes is None:
        indexes = torch.arange(S)
    if cu_seqlens is None:
        cu_seqlens = torch.tensor([0, S], dtype=torch.int32)
    h = O.embedding(input_ids, p["embedding.weight"])
    h, l_auxes = run_blocks(p, mc, h, indexes, cu_seqlens, noise_fn, routes, forced)
    x = O.rms_norm(h.float(), p["norm.weight"], mc.layer_norm_epsilon)
    return F.linear(x, p["head.weight"]).float(), l_auxes


def run_blocks(p, mc, h, indexes, cu_seqlens, noise_fn=None, routes=None, forced=None):
    """The stack of PackedFlashBaseLayer1D blocks (modeling_internlm.py:56-260 / modeling_moe.py) on the residual stream h [S, hidden] of one packed row;
    the stream keeps the dtype it comes in with (the reference's own block test feeds fp32 rows to bf16 blocks).  -> (h, [l_aux per MoE layer])."""
    S = h.shape[0]
    H, d, E = mc.num_attention_heads, mc.hidden_size // mc.num_attention_heads, mc.num_experts
    dt = p["blocks.0.norm1.weight"].dtype
    cos, sin = O.rotary_cos_sin(int(indexes.max()) + 1, d, mc.rope_base, dt)
    l_auxes = []
    for l in range(mc.num_layers):
        pre = f"blocks.{l}."
        residual = h
        x = O.rms_norm(residual.float(), p[pre + "norm1.weight"], mc.layer_norm_epsilon)
        qkv = F.linear(x, p[pre + "mixer.Wqkv.weight"], p[pre + "mixer.Wqkv.bias"])          # "(three h d)"
        # the shared rotary / attention helpers take InternLM2's [kv group][q, k, v] order: with one q head per kv head that is [h][three][d]
        qkv = qkv.reshape(S, 3, H, d).permute(0, 2, 1, 3).reshape(S, 3 * H * d)
        q, kv = O.qkv_split_rotary(qkv, cos, sin, indexes, H, 1, d, interleaved=False, inplace_qkv_form=True)
        ctx = O.attention_varlen(q, kv, cu_seqlens, causal=True)
        attn = F.linear(ctx.reshape(S, -1), p[pre + "mixer.out_proj.weight"], p[pre + "mixer.out_proj.bias"])
        residual = attn + residual
        x = O.rms_norm(residual.float(), p[pre + "norm2.weight"], mc.layer_norm_epsilon)
        if E <= 1:   # FeedForward.forward (modules/mlp.py:82-86): w2(silu(w1 x) * w3 x)
            y = F.linear(O.swiglu(F.linear(x, p[pre + "mlp.w1.weight"]), F.linear(x, p[pre + "mlp.w3.weight"])), p[pre + "mlp.w2.weight"])
            h = y + residual
            continue
        ex = pre + "mlp.moe_layer.experts.wrapped_experts."
        w1 = torch.stack([p[ex + f"{e}.w1.weight"] for e in range(E)])
        w3 = torch.stack([p[ex + f"{e}.w3.weight"] for e in range(E)])
        w2 = torch.stack([p[ex + f"{e}.w2.weight"] for e in range(E)])
        y, l_aux, route = MO.moe_layer(x, p[pre + "mlp.moe_layer.gate.wg.weight"], w1, w3, w2, noise_fn(l), mc.moe_capacity_factor, mc.moe_min_capacity,
                                       None if forced is None else forced[l])
        l_auxes.append(l_aux)
        if routes is not None:   # (diagnostics: which expert / slot every token got in this layer)
            routes.append(route)
        h = y + residual
    return h, l_auxes


class OracleMoETrainer(OracleTrainer):
    """OracleTrainer with the MoE model, the moe loss and per-group norms / clipping.  noise_seed(call) -> seed of the call-th gating call
    of the run (layer-major inside a micro-batch), as the reference harness injects it."""

    def __init__(self, path_cfg, dtype=torch.bfloat16, init_fn=None, noise_seed=lambda call: 5000 + call):
        self.mc, self.tc = path_cfg.model, path_cfg.train
        self.dtype = dtype
        init = init_fn or moe_formula_init
        # the gate is an fp32 module whatever the model dtype (set_fp32_attr_to_module, modeling_moe.py:161)
        self.params = {n: init(n, s).to(torch.float32 if group_of(n) == "1_fp32" else dtype).requires_grad_(True) for n, s in param_shapes(self.mc).items()}
        self.names = list(self.params.keys())
        self.master = {n: p.detach().clone().float() for n, p in self.params.items()}
        self.m = {n: torch.zeros_like(t) for n, t in self.master.items()}
        self.v = {n: torch.zeros_like(t) for n, t in self.master.items()}
        tc = self.tc
        self.scaler = O.DynamicGradScaler(1.0 if dtype == torch.float32 else tc.initial_scale, tc.growth_factor, tc.backoff_factor, tc.growth_interval,
                                          tc.min_scale, tc.max_scale, tc.hysteresis)
        self.adam_step = self.k = self.beta2_iter = 0
        self.metric = None
        self.noise_seed, self.calls = noise_seed, 0

    def _noise(self, S):
        def fn(layer):
            n = MO.gumbel_noise((S, self.mc.num_experts), self.noise_seed(self.calls))
            self.calls += 1
            return n

        return fn

    def train_step(self, batch, labels, forced=None):
        total, moe_total = self.backward(batch, labels, forced)
        return self.update(total, moe_total)

    def backward(self, batch, labels, forced=None):
        """forward + (loss-scaled) backward over the micro-batches; gradients accumulate in params[n].grad.  -> (loss, moe loss).
        forced[i][l] = int64 [2, S] expert choices to use in micro-batch i, layer l (teacher-forced routing, see oracle.moe.top2gating);
        self.routes[i][l] keeps what was routed."""
        self.routes = []
        tc, mc = self.tc, self.mc
        for p in self.params.values():
            p.grad = None
        M = batch["input_ids"].shape[0]
        total = moe_total = 0.0
        for i in range(M):
            cu = batch["cu_seqlens"][i] if batch.get("cu_seqlens") is not None else None
            idx = batch["indexes"][i] if batch.get("indexes") is not None else None
            S = batch["input_ids"][i].shape[0]
            self.routes.append([])
            logits, l_auxes = forward_logits(self.params, mc, batch["input_ids"][i], self._noise(S), idx, cu, self.routes[-1], None if forced is None else forced[i])
            loss = O.cross_entropy(logits, labels[i], tc.label_smoothing)
            if l_auxes:
                moe_loss = sum(l_auxes) * mc.moe_loss_coeff        # model-dtype tensors (the gate's outputs are cast back to it)
                moe_loss = moe_loss / M
                loss = loss / M + moe_loss
                moe_total += float(moe_loss.detach())
            else:                                                  # the dense InternLM-1 model: no auxiliary loss
                loss = loss / M
            total += float(loss.detach())
            (self.scaler.scale * loss).backward()
        return total, moe_total

    def update(self, total=0.0, moe_total=0.0, moe_sq_scale=1.0):
        """HybridZeroOptimizer.step on the accumulated gradients: group norms, overflow check, scaler, per-group clip, AdamW.
        moe_sq_scale: factor on the squared norm of the expert group (expert-parallel runs: OracleMoEDataParallel)."""
        tc = self.tc
        groups = {"0_default": [], "1_fp32": [], "2_moe_ep_size_1": []}
        for n in self.names:
            groups[group_of(n)].append(n)
        sq = {}
        for gname, names in groups.items():
            acc = 0.0
            for n in names:
                acc = acc + torch.norm(self.params[n].grad.float(), 2.0) ** 2.0
            sq[gname] = float(acc) * (moe_sq_scale if gname.startswith("2_moe") else 1.0)
        found_inf, found_nan = any(math.isinf(v) for v in sq.values()), any(math.isnan(v) for v in sq.values())
        loss_scale = self.scaler.scale
        if self.dtype != torch.float32:
            self.scaler.update(found_inf)
        if found_inf or found_nan:
            return {"loss": total, "moe_loss": moe_total, "grad_norm": {g: (-1.0 if found_inf else -2.0) for g in groups}, "ok": False, "loss_scale": self.scaler.scale}
        lr, beta2 = self._lr(), self._beta2()
        self.adam_step += 1
        norms = {}
        with torch.no_grad():
            for gname, names in groups.items():
                gnorm = sq[gname] ** 0.5
                norms[gname] = gnorm / loss_scale
                inv = 1.0
                if self.dtype != torch.float32 and tc.clip_grad_norm > 0:   # every group is clipped by its OWN norm (hybrid_zero_optim.py:863-876)
                    inv = 1.0 / O.unscale_clip_factor(gnorm, loss_scale, tc.clip_grad_norm)
                for n in names:
                    g = self.params[n].grad.float()
                    g.mul_(inv)
                    O.adamw_step(self.master[n], g, self.m[n], self.v[n], self.adam_step, lr, tc.adam_beta1, beta2, tc.adam_eps, tc.weight_decay)
                    self.params[n].copy_(self.master[n])
        self.k += 1
        self.beta2_iter += 1
        return {"loss": total, "moe_loss": moe_total, "grad_norm": norms, "ok": True, "loss_scale": self.scaler.scale, "lr": lr}


class OracleMoEDataParallel(OracleMoETrainer):
    """`world` data-parallel ranks of the MoE family in ONE process, with the reference's AUTOMATIC expert parallelism
    (parallel_context.py:538-541: ep = min(dp, num_experts); here world <= num_experts, so ep = world and every expert lives on exactly one rank):
      * every rank gates its OWN micro-batches (noise seed 5000 + 1000 rank + call, as the harness injects it); its tokens visit experts on all
        ranks through the all_to_all of gshard_layer.py:453-474 -- numerically the layer with all experts local;
      * dense parameters and gates: gradients averaged over the data-parallel ranks (all-reduce AVG over DATA);
      * expert parameters: an expert's gradient is the SUM of what every rank's tokens contribute (they all pass through the one copy of the
        expert in one backward), then averaged over the expert-data group (hybrid_zero_optim.py:166-167) -- of size dp / ep = 1 here: no 1 / ep;
      * the squared norm of the moe group: every rank's local sum of squares / dp, summed over the expert group
        (solver/optimizer/utils.py:362-368) = (sum over ALL experts) / dp.
    Pinned by tests/golden/train_moe2_bf16_rank{0,1}.json = the unmodified reference on two gloo ranks (make_golden.py --run-mp moe2_bf16)."""

    def __init__(self, path_cfg, world, dtype=torch.bfloat16, init_fn=None):
        super().__init__(path_cfg, dtype, init_fn)
        if world > self.mc.num_experts or self.mc.num_experts % world:
            raise NotImplementedError("emulates ep = dp (world <= num_experts, dividing it)")
        self.world = world
        self.calls_of = [0] * world

    def train_step(self, batches, labels, forced=None):
        """batches / labels: one entry per rank.  -> list of per-rank result dicts (the norms are global)."""
        per_rank = []
        for r in range(self.world):
            self.noise_seed, self.calls = (lambda call, r=r: 5000 + 1000 * r + call), self.calls_of[r]
            total, moe_total = self.backward(batches[r], labels[r], None if forced is None else forced[r])
            self.calls_of[r] = self.calls
            per_rank.append(({n: p.grad.detach().clone() for n, p in self.params.items()}, total, moe_total, self.routes))
        with torch.no_grad():
            for n, p in self.params.items():
                gs = [pr[0][n] for pr in per_rank]
                acc = gs[0].clone()
                for g in gs[1:]:
                    acc += g                                  # (the reduction sums in the gradient's dtype)
                p.grad = acc if group_of(n).startswith("2_moe") else acc / self.world
        self.routes_of = [pr[3] for pr in per_rank]
        res = self.update(0.0, 0.0, moe_sq_scale=1.0 / self.world)
        return [dict(res, loss=pr[1], moe_loss=pr[2]) for pr in per_rank]
