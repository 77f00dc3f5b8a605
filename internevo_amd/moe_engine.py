"""Training-step engine of the INTERNLM_MoE model family (BASELINE configs[4], configs/7B_MoE4_sft.py) on the HIP kernels.

Host-side mirror of
  PackedFlashInternLm1D / PackedFlashBaseLayer1D     internlm/model/modeling_moe.py:33-444 (norm1 -> MHA -> residual -> norm2 -> MoE -> residual)
  MHA (InternLM-1: packed Wqkv "(three h d)" + bias, NeoX rotary, out_proj + bias)     modules/multi_head_attention.py:298-478
  MoE / GShardMOELayer                                internevo_amd/moe.py (csrc/moe.hip)
  the moe loss                                        core/scheduler/no_pipeline_scheduler.py:120-145
  the optimizer's parameter groups default / fp32 / moe, one norm and one clipping factor each
                                                      train/utils.py:25-80, solver/optimizer/hybrid_zero_optim.py:760-779,863-876
Same construction as engine.InternLM2Engine (explicit backward over pre-allocated buffers, flat bf16 parameters / gradients, device-
resident loss scale and step control, no host synchronisation inside a step); the scope of this engine is the single-rank and the
data-parallel step with the reference's automatic expert parallelism: ep = min(world, num_experts) (parallel_context.py:538-541), expert
groups of ep CONSECUTIVE ranks each holding num_experts / ep experts, the [E, C, M] dispatch / combine buffers exchanged by one all_to_all per
direction (moe.MoELayer), dense parameters and gates averaged by all-reduce over all ranks, an expert's gradient summed over its expert
group's tokens and averaged over its expert-data group only, the moe group's norm scaled as solver/optimizer/utils.py:362-368; optimizer
state replicated (`parallel.zero1.size = 1` semantics).
Megatron tensor parallelism (parallel.tensor = dict(size=tp, mode="mtp"); round 5): tensor groups of tp CONSECUTIVE ranks that read the same
micro-batches and draw the same gate noise; Wqkv (+ bias) cut by heads, out_proj by input columns (its bias added once, behind the all-reduce),
every expert's w1 / w3 by rows and w2 by columns (gshard_layer.py:421-433: the experts are FeedForward modules over the TENSOR group), the head by
vocabulary rows with the vocabulary-parallel loss of the dense engine; embedding, norms and gates whole on every rank.  Four all-reduces per layer
and micro-batch (out_proj / expert outputs forward, Wqkv / w1|w3 input gradients backward), data and expert parallelism over the ranks that hold the
same shard (expert groups = consecutive entries of a data-parallel group, process_group_initializer.py:493-524), group norms summed over the tensor
group with the replicated parameters counted once (solver/optimizer/utils.py:225-262,330-352).  Pinned on a 2-process run of the reference
(tests/golden/train_moe_tp2_bf16_rank*.json).  Sequence-sharded modes (msp / fsp), sequence and pipeline parallelism are refused.  Checkpoints under tensor
parallelism: every tensor rank its own model / expert / optimizer files (pinned on tests/golden/ckpt_ref_moe_tp2dp2/).
Weights: Wqkv is kept in the [head][q, k, v][d] row order of the shared rotary / attention kernels and converted at the naming boundary
(`named_parameters` / `load_named_parameters`), exactly like LLAMA2's wq / wk / wv in the dense engine.
"""
import math
import os

import torch
import torch.distributed as dist

from . import kernels as K
from ._lib import IeScalerConfig, check
from .comm import backend_for
from .config import PathConfig
from .moe import MoELayer
from .schedule import Beta2Scheduler, CosineWarmupLR
from .tensorpar import TensorParallel

BF16 = torch.bfloat16
GROUPS = ("0_default", "1_fp32", "2_moe_ep_size_1")


def ffn_dim(mc):
    f = int(mc.hidden_size * mc.mlp_ratio)
    return mc.multiple_of * ((f + mc.multiple_of - 1) // mc.multiple_of)


def group_of(name):
    if name.endswith("gate.wg.weight"):
        return 1
    return 2 if ".experts." in name else 0


class MoEEngine:
    def __init__(self, cfg: PathConfig, device, process_group=None, world_size=1, rank=0, seed=1024, init_fn=None, noise_fn=None, tp_size=None, expert_fp8=None):
        """tp_size: Megatron tensor-parallel size (default: cfg.train.tp_size).
        noise_fn(call_index, S, E) -> fp32 [S, E] device tensor: test hook that injects the Gumbel noise of the call-th gating call of the
        run (layer-major inside a micro-batch); None = generated on the device from (seed, layer, call)."""
        self.cfg, self.mc, self.tc = cfg, cfg.model, cfg.train
        mc, tc = self.mc, self.tc
        if mc.model_type not in ("INTERNLM_MoE", "INTERNLM"):
            raise ValueError("MoEEngine runs the InternLM-1 families: model_type INTERNLM_MoE, or the dense INTERNLM model")
        # dense: the InternLM-1 model proper (modeling_internlm.py; INTERNLM_MoE with num_experts = 1 builds the same block, modeling_moe.py:120-140):
        # a plain SwiGLU FeedForward in place of the MoE -- no gate, no auxiliary loss, one optimizer group
        self.dense = mc.model_type == "INTERNLM" or mc.num_experts < 2
        # expert_fp8 (OPT-IN: this argument, `moe = dict(..., expert_fp8=True)` in the config, or IE_EXPERT_FP8=1): the experts' two FORWARD products on OCP e4m3
        # operands (moe.MoELayer; BASELINE configs[4] "fp8 MFMA linear layers").  Off by default: slower in the step, and the reference has no fp8 arithmetic to pin
        want_fp8 = (bool(int(os.environ.get("IE_EXPERT_FP8", "0"))) or bool(getattr(mc, "moe_expert_fp8", False))) if expert_fp8 is None else bool(expert_fp8)
        self.expert_fp8 = want_fp8 and not self.dense
        if mc.num_kv_attention_heads != mc.num_attention_heads:
            raise NotImplementedError("the InternLM-1 block has no grouped-query attention")
        K._L()
        self.dev, self.world, self.rank, self.group = device, world_size, rank, process_group
        if world_size > 1 and not dist.is_initialized():
            raise RuntimeError("torch.distributed must be initialised for world_size > 1")
        self.be = backend_for(process_group) if world_size > 1 else None
        self.backend = self.be.name if self.be else None
        # Expert parallelism as the reference sets it up (parallel_context.py:538-541: ep = min(data-parallel size, num_experts); expert groups =
        # ep CONSECUTIVE data-parallel ranks, expert-data groups = the ranks with the same position in their expert group): every rank holds
        # E / ep experts, its tokens visit the others through the all_to_all of the dispatch buffers (moe.MoELayer).
        E_ = max(mc.num_experts, 1)
        tp = int(tp_size if tp_size is not None else getattr(tc, "tp_size", 1))
        if tp > 1 and getattr(tc, "tp_mode", "mtp") != "mtp":
            raise NotImplementedError("MoEEngine: tensor parallelism in mode 'mtp' only (msp / fsp shard the sequence in front of the gate)")
        if tp > 1 and process_group is not None:
            raise NotImplementedError("MoEEngine runs over the default group")
        if tp > 1 and tc.label_smoothing > 0:
            raise NotImplementedError("MoEEngine: label smoothing with the vocabulary-parallel loss (the dense engine has it)")
        if mc.num_attention_heads % tp or ffn_dim(mc) % tp or mc.vocab_size % tp:
            raise ValueError(f"heads {mc.num_attention_heads}, FFN units {ffn_dim(mc)} and vocabulary {mc.vocab_size} must split over {tp} tensor ranks")
        self.tpar = TensorParallel(tp, rank, world_size, vocab_parallel=True, embed_split=False)   # (collective: creates the tensor / data groups)
        self.tp, self.tp_rank = tp, self.tpar.tp_rank
        dpw, dpr = self.tpar.dp_world, self.tpar.dp_rank           # the ranks that hold the same shard: data (and expert) parallelism runs over them
        self.dp_world, self.dp_group = dpw, self.tpar.dp_group
        self.ep, self.ep_rank, self.ep_group, self.edp_group = 1, 0, None, None
        if dpw > 1 and not self.dense:
            if process_group is not None:
                raise NotImplementedError("MoEEngine runs over the default group (pure data parallelism + expert parallelism)")
            ep = min(dpw, E_)
            if dpw % ep or E_ % ep:
                raise NotImplementedError(f"expert parallel size {ep} must divide the data-parallel size {dpw} and the number of experts {E_}")
            # process_group_initializer.py:493-524: expert groups = ep consecutive entries of a data-parallel group [t, t + tp, t + 2 tp, ...],
            # expert-data groups = the entries with the same position in their expert group.  Collective: every rank creates every group, same order.
            for t in range(tp):
                dp_ranks = [t + k * tp for k in range(dpw)]
                for g in range(dpw // ep):
                    ranks = dp_ranks[g * ep : (g + 1) * ep]
                    grp = dist.new_group(ranks)
                    if rank in ranks:
                        self.ep_group = grp
                for j in range(ep):
                    ranks = dp_ranks[j::ep]
                    grp = dist.new_group(ranks)
                    if rank in ranks:
                        self.edp_group = grp
            self.ep, self.ep_rank = ep, dpr % ep
        self.El = E_ // self.ep
        self.groups = ("0_default", "1_fp32", f"2_moe_ep_size_{self.ep}")   # the optimizer groups' names (train/utils.py:25-80)
        self.noise_fn, self.calls = noise_fn, 0
        self.keep_routes = None   # set to [] to record (expert choices, gate logits) of every micro-batch and layer
        h, F, V, L, E = mc.hidden_size, ffn_dim(mc) // tp, mc.vocab_size, mc.num_layers, E_
        self.F = F                                   # this rank's FFN units
        self.Vl = Vl = V // tp                       # ... and vocabulary rows of the head
        H, d = mc.num_attention_heads // tp, mc.head_dim
        self.H = H                                   # ... and heads
        hl = H * d
        if d not in (64, 128):
            raise NotImplementedError("head dim must be 64 or 128")
        # ---- parameters: one flat bf16 buffer (+ gradients), one flat fp32 buffer for the gates (fp32 modules)
        specs = [("embedding.weight", (V, h))]
        for l in range(L):
            p = f"blocks.{l}."
            specs += [(p + "norm1.weight", (h,)), (p + "mixer.Wqkv.weight", (3 * hl, h)), (p + "mixer.Wqkv.bias", (3 * hl,)),
                      (p + "mixer.out_proj.weight", (h, hl)), (p + "mixer.out_proj.bias", (h,)), (p + "norm2.weight", (h,)),
                      (p + "mlp.w13", (self.El, 2 * F, h)), (p + "mlp.w2", (self.El, h, F))]   # this rank's experts: w1 | w3 fused per expert, adjacent
        specs += [("norm.weight", (h,)), ("head.weight", (Vl, h))]
        off, self.spec = 0, {}
        for n, shp in specs:
            numel = math.prod(shp)
            self.spec[n] = (off, shp)
            off += (numel + 7) // 8 * 8
        self.params = torch.zeros(off, dtype=BF16, device=device)
        self.grads = torch.zeros(off, dtype=BF16, device=device)
        self.p = {n: self.params[o : o + math.prod(s)].view(s) for n, (o, s) in self.spec.items()}
        self.g = {n: self.grads[o : o + math.prod(s)].view(s) for n, (o, s) in self.spec.items()}
        self.wg = torch.zeros(L, E, h, dtype=torch.float32, device=device)       # blocks.{l}.mlp.moe_layer.gate.wg.weight
        self.d_wg = torch.zeros_like(self.wg)
        self.master = torch.zeros(off, dtype=torch.float32, device=device)
        self.exp_avg, self.exp_avg_sq = torch.zeros_like(self.master), torch.zeros_like(self.master)
        self.wg_m, self.wg_v = torch.zeros_like(self.wg), torch.zeros_like(self.wg)
        # slices of the flat buffer by optimizer group (contiguous runs: experts of a layer vs everything else)
        self.runs = {0: [], 2: []}
        for n, (o, s) in self.spec.items():
            g = 2 if n.endswith(("mlp.w13", "mlp.w2")) and not self.dense else 0
            numel = (math.prod(s) + 7) // 8 * 8
            if self.runs[g] and self.runs[g][-1][1] == o:   # adjacent parameters of one group: one launch
                self.runs[g][-1] = (self.runs[g][-1][0], o + numel)
            else:
                self.runs[g].append((o, o + numel))
        # group 0 under tensor parallelism: what is cut over the tensor group (its squared norm is summed over the group) and what every rank holds whole
        # (counted once: reduce_grads, solver/optimizer/utils.py:225-262)
        self.cut0, self.whole0 = [], []
        for n, (o, s_) in self.spec.items():
            if n.endswith(("mlp.w13", "mlp.w2")) and not self.dense:
                continue
            cut = n.endswith(("Wqkv.weight", "Wqkv.bias", "out_proj.weight", "head.weight", "mlp.w13", "mlp.w2"))
            (self.cut0 if cut else self.whole0).append(self.grads[o : o + math.prod(s_)])
        # AdamW of the previous step runs on its own stream under the next forward pass (as in the dense engine): one event per run, and the
        # forward waits for the runs that hold the layer it is about to use
        self.opt_stream = torch.cuda.Stream(device=device)
        self._opt_events = None        # {run start offset: event}, "gate": event -- of the step() still in flight, if any

        def run_of(name):
            o = self.spec[name][0]
            return next(a for grp in (0, 2) for a, b in self.runs[grp] if a <= o < b)

        self._layer_runs = [sorted({run_of(f"blocks.{l}.norm1.weight"), run_of(f"blocks.{l}.mlp.w13"), run_of(f"blocks.{l}.mlp.w2")}) for l in range(L)]
        self._embed_run, self._head_runs = run_of("embedding.weight"), sorted({run_of("norm.weight"), run_of("head.weight")})
        self._init_params(seed, init_fn)
        self.master.copy_(self.params)
        # ---- step state
        self.state = K.step_state_new(device, tc.initial_scale)
        self.scaler_cfg = IeScalerConfig(tc.growth_factor, tc.backoff_factor, tc.min_scale, tc.max_scale, tc.growth_interval, tc.hysteresis, tc.clip_grad_norm, 1)
        self.lr_sched = CosineWarmupLR(tc.lr, tc.total_steps, tc.warmup_ratio, tc.eta_min, tc.init_steps)
        self.beta2_sched = Beta2Scheduler(tc.adam_beta2, tc.adam_beta2_c)
        self.sumsq = torch.zeros(3, dtype=torch.float32, device=device)
        self.group_inv = torch.zeros(3, dtype=torch.float32, device=device)
        self.group_norm = torch.zeros(3, dtype=torch.float32, device=device)
        self.scale_view = self.state[:4].view(torch.float32)
        # ---- rotary tables, activations
        inv_freq = 1.0 / (mc.rope_base ** (torch.arange(0, d, 2, dtype=torch.float32) / d))
        freqs = torch.outer(torch.arange(tc.seq_len, dtype=torch.float32), inv_freq)
        self.cos, self.sin = torch.cos(freqs).to(BF16).to(device), torch.sin(freqs).to(BF16).to(device)
        T = self.T = tc.packed_length

        def e(*shape, dtype=BF16):
            return torch.empty(shape, dtype=dtype, device=device)

        self.a_x = [e(T, h) for _ in range(L)]
        self.a_n1, self.a_rstd1 = [e(T, h) for _ in range(L)], [e(T, dtype=torch.float32) for _ in range(L)]
        self.a_q, self.a_kv, self.a_ctx = [e(T, H, d) for _ in range(L)], [e(T, 2, H, d) for _ in range(L)], [e(T, H, d) for _ in range(L)]
        self.a_lse = [e(H, T, dtype=torch.float32) for _ in range(L)]
        self.a_r2, self.a_n2, self.a_rstd2 = [e(T, h) for _ in range(L)], [e(T, h) for _ in range(L)], [e(T, dtype=torch.float32) for _ in range(L)]
        if self.dense:
            self.moe = None
            self.a_h13 = [e(T, 2 * F) for _ in range(L)]
            self.t_act, self.t_dact, self.t_dh13 = e(T, F), e(T, F), e(T, 2 * F)
        else:
            self.moe = [MoELayer(h, F, E, T, device, mc.moe_capacity_factor, mc.moe_min_capacity, seed=seed + 7919 * dpr, layer_index=l, ep_group=self.ep_group,
                                 ep_size=self.ep, ep_rank=self.ep_rank, tpar=self.tpar, expert_fp8=self.expert_fp8) for l in range(L)]
            # (every DATA-parallel rank gates its own tokens with its own noise; the ranks of a tensor group gate the same tokens with the same noise)
        self.a_xf, self.a_nf, self.a_rstdf = e(T, h), e(T, h), e(T, dtype=torch.float32)
        self.t_qkv, self.t_h0, self.t_h1, self.t_h2 = e(T, 3 * hl), e(T, h), e(T, h), e(T, h)
        self.t_dq, self.t_dkv = e(T, H, d), e(T, 2, H, d)
        self.t_logits, self.t_loss_rows, self.t_lse, self.t_loss = e(T, Vl), e(T, dtype=torch.float32), e(T, dtype=torch.float32), e(2, dtype=torch.float32)
        self.t_lab_local = torch.empty(T, dtype=torch.int64, device=device) if tp > 1 else None
        self.t_delta = e(K._L().ie_flash_attn_bwd_workspace(T, H, H, d), dtype=torch.float32)
        self.t_norm_ws = e(K._L().ie_rmsnorm_bwd_partials(T) * h, dtype=torch.float32)
        self.t_emb_ws = e(V + 1 + T, dtype=torch.int32)
        self.t_bias3, self.t_bias1 = e(3 * hl), e(h)
        self.loss_acc = torch.zeros(1, dtype=torch.float32, device=device)
        self.moe_acc = torch.zeros(1, dtype=torch.float32, device=device)
        self.step_count = 0

    # ------------------------------------------------------------------------------------------ parameters
    def reference_param_shapes(self):
        mc, out = self.mc, {}
        h, F, V, E = mc.hidden_size, self.F * self.tp, mc.vocab_size, max(mc.num_experts, 1)   # (the FULL shapes: a one-rank job's)
        out["embedding.weight"] = (V, h)
        for l in range(mc.num_layers):
            p = f"blocks.{l}."
            out[p + "mixer.Wqkv.weight"], out[p + "mixer.Wqkv.bias"] = (3 * h, h), (3 * h,)
            out[p + "mixer.out_proj.weight"], out[p + "mixer.out_proj.bias"] = (h, h), (h,)
            out[p + "norm1.weight"], out[p + "norm2.weight"] = (h,), (h,)
            if self.dense:
                out[p + "mlp.w1.weight"], out[p + "mlp.w2.weight"], out[p + "mlp.w3.weight"] = (F, h), (h, F), (F, h)
                continue
            out[p + "mlp.moe_layer.gate.wg.weight"] = (E, h)
            for e_ in range(E):
                q = p + f"mlp.moe_layer.experts.wrapped_experts.{e_}."
                out[q + "w1.weight"], out[q + "w2.weight"], out[q + "w3.weight"] = (F, h), (h, F), (F, h)
        out["norm.weight"], out["head.weight"] = (h,), (V, h)
        return out

    def _qkv_to_engine(self, t):
        """the FULL reference tensor, "(three h d)" rows -> this rank's heads as [h][three][d] rows (weights [3h, h] or biases [3h])"""
        Hf, H, d, r = self.mc.num_attention_heads, self.H, self.mc.head_dim, self.tp_rank
        v = t.reshape(3, Hf, d, -1)[:, r * H : (r + 1) * H]
        return v.permute(1, 0, 2, 3).reshape((3 * H * d,) + tuple(t.shape[1:]))

    def _qkv_to_reference(self, t):
        """this rank's [h][three][d] rows -> "(three h d)" rows of its heads: what the reference's tensor rank holds (multi_head_attention.py: the
        rearrange runs with the LOCAL head count)"""
        H, d = self.H, self.mc.head_dim
        return t.reshape(H, 3, d, -1).permute(1, 0, 2, 3).reshape(t.shape)

    def _cut(self, t, dim):
        """this tensor rank's 1 / tp of a full parameter along `dim`"""
        if self.tp == 1:
            return t
        n = t.shape[dim] // self.tp
        return t.narrow(dim, self.tp_rank * n, n)

    def load_named_parameters(self, named, sync_master=True, views=None, gates=None):
        """named: the reference's FULL parameter tensors by name (PackedFlashInternLm1D.named_parameters() of a one-rank job; a tensor rank keeps its
        part: tensorpar.py's rule).  views / gates: `_views(flat)` of another
        buffer of the same layout and the [L, E, h] fp32 tensor to fill instead of the bf16 parameters and the gate weights (checkpoints: master
        weights, moments)."""
        self._wait_optimizer()
        for lay in getattr(self, "moe", None) or ():   # (the constructor initialises the weights before the layers exist)
            lay.invalidate_fp8()
        F = self.F
        P = self.p if views is None else views
        WG = self.wg if gates is None else gates
        sync_master = sync_master and views is None
        for n, t in named.items():
            t = t.to(self.dev)
            if n.endswith("gate.wg.weight"):
                WG[int(n.split(".")[1])].copy_(t.float())
            elif ".experts." in n:
                parts = n.split(".")
                l, e_, w = int(parts[1]), int(parts[6]) - self.ep_rank * self.El, parts[7]   # (names carry the GLOBAL expert index)
                if not 0 <= e_ < self.El:
                    continue                                                                 # another rank's expert
                if w == "w2":
                    P[f"blocks.{l}.mlp.w2"][e_].copy_(self._cut(t, 1))
                else:
                    P[f"blocks.{l}.mlp.w13"][e_][(0 if w == "w1" else F) : (F if w == "w1" else 2 * F)].copy_(self._cut(t, 0))
            elif "mixer.Wqkv" in n:
                P[n].copy_(self._qkv_to_engine(t))
            elif n.endswith("mixer.out_proj.weight"):
                P[n].copy_(self._cut(t, 1))
            elif n == "head.weight":
                P[n].copy_(self._cut(t, 0))
            elif self.dense and ".mlp.w" in n:   # blocks.{l}.mlp.w1 / w2 / w3.weight -> the fused [1, 2F, h] / [1, h, F] tensors
                l, w = int(n.split(".")[1]), n.split(".")[3]
                if w == "w2":
                    P[f"blocks.{l}.mlp.w2"][0].copy_(self._cut(t, 1))
                else:
                    P[f"blocks.{l}.mlp.w13"][0][(0 if w == "w1" else F) : (F if w == "w1" else 2 * F)].copy_(self._cut(t, 0))
            else:
                P[n].copy_(t)
        if sync_master:
            self.master.copy_(self.params)

    def _views(self, flat):
        """The engine-named views of a flat buffer laid out like `params` (master weights, Adam moments)."""
        return {n: flat[o : o + math.prod(s)].view(s) for n, (o, s) in self.spec.items()}

    def named_parameters(self, views=None, gates=None):
        """(reference name, tensor) pairs (copies for the re-ordered / fused tensors); under tensor parallelism the tensors are this rank's parts, laid
        out as the reference's tensor rank holds them.  views / gates: `_views(flat)` of another buffer of the same
        layout and an [L, E, h] fp32 tensor (the fp32 master weights / moments, for checkpoints) instead of the bf16 parameters and the gate weights."""
        self._wait_optimizer()
        F, out = self.F, {}
        P = self.p if views is None else views
        WG = self.wg if gates is None else gates
        for n, shp in self.reference_param_shapes().items():
            if n.endswith("gate.wg.weight"):
                out[n] = WG[int(n.split(".")[1])]
            elif ".experts." in n:
                parts = n.split(".")
                l, e_, w = int(parts[1]), int(parts[6]) - self.ep_rank * self.El, parts[7]
                if not 0 <= e_ < self.El:
                    continue                       # (held by another rank of the expert group)
                out[n] = P[f"blocks.{l}.mlp.w2"][e_] if w == "w2" else P[f"blocks.{l}.mlp.w13"][e_][(0 if w == "w1" else F) : (F if w == "w1" else 2 * F)]
            elif "mixer.Wqkv" in n:
                out[n] = self._qkv_to_reference(P[n])
            elif self.dense and ".mlp.w" in n:
                l, w = int(n.split(".")[1]), n.split(".")[3]
                out[n] = P[f"blocks.{l}.mlp.w2"][0] if w == "w2" else P[f"blocks.{l}.mlp.w13"][0][(0 if w == "w1" else F) : (F if w == "w1" else 2 * F)]
            else:
                out[n] = P[n]
        return out.items()

    def _init_params(self, seed, init_fn):
        """modeling_moe.py:170-198: normal(0.006) Wqkv / w1 / w3, normal(0.0015) (scaled by 1/sqrt(2(l+1)) with use_scaled_init) out_proj / w2,
        zero biases, unit norms; embedding / head normal(0.0052) (:364-366,:413-415)."""
        if init_fn is not None:
            self.load_named_parameters({n: init_fn(n, s) for n, s in self.reference_param_shapes().items()}, sync_master=False)
            return
        gen = torch.Generator(device=self.dev).manual_seed(seed)
        named = {}
        for n, s in self.reference_param_shapes().items():
            l = int(n.split(".")[1]) if n.startswith("blocks.") else 0
            if n.endswith("bias"):
                t = torch.zeros(s, device=self.dev)
            elif len(s) == 1:
                t = torch.ones(s, device=self.dev)
            else:
                std = 0.0052 if n in ("embedding.weight", "head.weight") else 0.006
                if n.endswith(("out_proj.weight", "w2.weight")):
                    std = 0.006 / math.sqrt(2.0 * (l + 1)) if self.mc.use_scaled_init else 0.0015
                t = torch.empty(s, dtype=torch.float32, device=self.dev).normal_(0.0, std, generator=gen)
            named[n] = t
        self.load_named_parameters(named, sync_master=False)

    # ------------------------------------------------------------------------------------------ forward / backward of one micro-batch
    def _bias_add(self, y, b):
        check(K._L().ie_bias_add_bf16(K._p(y), y.stride(0), K._p(b), y.shape[0], y.shape[1], K._stream()), "ie_bias_add_bf16")

    def _noise(self, S, E):
        n = None if self.noise_fn is None else self.noise_fn(self.calls, S, E)
        self.calls += 1
        return n

    def _forward_micro(self, ids, labels, cu, pos, max_seqlen):
        mc, p = self.mc, self.p
        L, eps, H, d = mc.num_layers, mc.layer_norm_epsilon, self.H, mc.head_dim
        tpar = self.tpar
        self._wait_runs([self._embed_run, "gate"])
        K.embedding_fwd(p["embedding.weight"], ids, self.a_x[0])
        moe_out = None
        self.l_aux = []
        for l in range(L):
            pre = f"blocks.{l}."
            self._wait_runs(self._layer_runs[l])
            if l == 0:
                K.rmsnorm_fwd(self.a_x[0], p[pre + "norm1.weight"], eps, self.a_n1[0], self.a_rstd1[0])
            else:
                K.add_rmsnorm_fwd(moe_out, self.a_r2[l - 1], p[pre + "norm1.weight"], eps, self.a_x[l], self.a_n1[l], self.a_rstd1[l])
            K.linear_fwd(self.a_n1[l], p[pre + "mixer.Wqkv.weight"], self.t_qkv)
            self._bias_add(self.t_qkv, p[pre + "mixer.Wqkv.bias"])
            K.qkv_rotary_fwd(self.t_qkv, self.cos, self.sin, pos, H, 1, d, False, self.a_q[l], self.a_kv[l])
            K.flash_attn_fwd(self.a_q[l], self.a_kv[l][:, 0], self.a_kv[l][:, 1], cu, max_seqlen, None, True, self.a_ctx[l], self.a_lse[l])
            K.linear_fwd(self.a_ctx[l].view(self.T, -1), p[pre + "mixer.out_proj.weight"], self.t_h0)
            tpar.all_reduce_sum(self.t_h0)     # row-parallel: partial sums over the tensor group (no-op at tp = 1); the bias once, behind the sum
            self._bias_add(self.t_h0, p[pre + "mixer.out_proj.bias"])
            K.add_rmsnorm_fwd(self.t_h0, self.a_x[l], p[pre + "norm2.weight"], eps, self.a_r2[l], self.a_n2[l], self.a_rstd2[l])
            moe_out = self.t_h1
            if self.dense:   # FeedForward (modules/mlp.py:82-86): w2(silu(w1 x) * w3 x), w1 | w3 as one GEMM
                F_ = self.F
                K.linear_fwd(self.a_n2[l], p[pre + "mlp.w13"][0], self.a_h13[l])
                K.swiglu_fwd(self.a_h13[l][:, :F_], self.a_h13[l][:, F_:], self.t_act)
                K.linear_fwd(self.t_act, p[pre + "mlp.w2"][0], moe_out)
                tpar.all_reduce_sum(moe_out)
                continue
            self.l_aux.append(self.moe[l].forward(self.a_n2[l], self.wg[l], p[pre + "mlp.w13"], p[pre + "mlp.w2"], moe_out, noise=self._noise(self.T, mc.num_experts)).clone())
        self._wait_runs(self._head_runs)
        K.add_rmsnorm_fwd(moe_out, self.a_r2[L - 1], p["norm.weight"], eps, self.a_xf, self.a_nf, self.a_rstdf)
        K.linear_fwd(self.a_nf, p["head.weight"], self.t_logits)
        if self.tp == 1:
            K.ce_fwd(self.t_logits, labels, -100, self.tc.label_smoothing, self.t_loss_rows, self.t_lse, self.t_loss)
            return
        # vocabulary-parallel head (ScaleColumnParallelLinear + the parallel loss; engine.InternLM2Engine._cross_entropy): the fused kernel on this rank's
        # [T, V / tp] columns with the labels mapped into its range (-1 = valid, another rank's column), then ONE all-gather of (local log-sum-exp,
        # local target logit) per token gives the global log-sum-exp (kept for the backward) and the loss
        v0 = self.tp_rank * self.Vl
        here = (labels >= v0) & (labels < v0 + self.Vl)
        self.t_lab_local.copy_(torch.where(labels == -100, labels, torch.where(here, labels - v0, torch.full_like(labels, -1))))
        K.ce_fwd(self.t_logits, self.t_lab_local, -100, 0.0, self.t_loss_rows, self.t_lse, self.t_loss)
        stats = tpar.all_gather(torch.stack([self.t_lse, torch.where(here, self.t_lse - self.t_loss_rows, torch.zeros_like(self.t_loss_rows))]))   # [tp, 2, T]
        self.t_lse.copy_(torch.logsumexp(stats[:, 0], dim=0))
        self.t_loss_rows.copy_(torch.where(labels != -100, self.t_lse - stats[:, 1].sum(dim=0), torch.zeros_like(self.t_loss_rows)))
        K.ce_mean(self.t_loss_rows, labels, -100, self.t_loss)

    def _backward_micro(self, ids, labels, cu, pos, max_seqlen, acc, inv_m):
        mc, tc, p, g = self.mc, self.tc, self.p, self.g
        self._wait_optimizer()   # the backward overwrites the gradients the last step's AdamW reads
        L, H, d, T = mc.num_layers, self.H, mc.head_dim, self.T
        ws = self.t_norm_ws
        tpar = self.tpar
        # (vocabulary-parallel head: t_lse is the GLOBAL log-sum-exp, the labels the ones mapped into this rank's range)
        K.ce_bwd(self.t_logits, labels if self.tp == 1 else self.t_lab_local, self.t_lse, self.scale_view, self.t_loss[1:2], inv_m, -100, tc.label_smoothing)
        K.linear_dgrad(self.t_logits, p["head.weight"], self.t_h0)
        ar = tpar.all_reduce_sum_async(self.t_h0)    # column-parallel head: its input gradient is a partial sum; summed under the weight gradient
        K.linear_wgrad(self.t_logits, self.a_nf, g["head.weight"], acc)
        ar.wait()
        d_h = self.t_h1
        K.rmsnorm_bwd(self.t_h0, self.a_xf, p["norm.weight"], self.a_rstdf, None, g["norm.weight"], acc, ws, d_h)
        spare = [self.t_h0, self.t_h2]
        for l in range(L - 1, -1, -1):
            pre = f"blocks.{l}."
            d_n2 = spare[0]
            if self.dense:
                F_ = self.F
                K.linear_dgrad(d_h, p[pre + "mlp.w2"][0], self.t_dact)
                K.swiglu_bwd(self.t_dact, self.a_h13[l][:, :F_], self.a_h13[l][:, F_:], self.t_dh13[:, :F_], self.t_dh13[:, F_:], self.t_act)
                K.linear_wgrad(d_h, self.t_act, g[pre + "mlp.w2"][0], acc)
                K.linear_dgrad(self.t_dh13, p[pre + "mlp.w13"][0], d_n2)
                ar = tpar.all_reduce_sum_async(d_n2)
                K.linear_wgrad(self.t_dh13, self.a_n2[l], g[pre + "mlp.w13"][0], acc)
                ar.wait()
            else:
                self.moe[l].backward(d_h, self.wg[l], p[pre + "mlp.w13"], p[pre + "mlp.w2"], d_n2, self.d_wg[l], g[pre + "mlp.w13"], g[pre + "mlp.w2"],
                                     accumulate=acc, loss_scale_dev=self.scale_view, aux_factor=mc.moe_loss_coeff * inv_m)
            d_r2 = spare[1]
            K.rmsnorm_bwd(d_n2, self.a_r2[l], p[pre + "norm2.weight"], self.a_rstd2[l], d_h, g[pre + "norm2.weight"], acc, ws, d_r2)
            self._bias_grad(d_r2, g[pre + "mixer.out_proj.bias"], self.t_bias1, acc)
            d_ctx = d_n2.view(-1)[: T * H * d].view(T, H * d)   # (row-parallel backward: this rank's heads' columns, no exchange)
            K.linear_dgrad(d_r2, p[pre + "mixer.out_proj.weight"], d_ctx)
            K.linear_wgrad(d_r2, self.a_ctx[l].view(T, -1), g[pre + "mixer.out_proj.weight"], acc)
            # (round 6: delta, the rotary backward and the q | k | v re-packing inside the two attention kernels where the library fuses the shape -- the same bits)
            if not (os.environ.get("IE_ATTN_BWD_ROTARY_FUSE", "1") != "0" and
                    K.flash_attn_bwd_qkv_rotary(d_ctx.view(T, H, d), self.a_q[l], self.a_kv[l][:, 0], self.a_kv[l][:, 1], self.a_ctx[l], self.a_lse[l], cu, max_seqlen,
                                                self.cos, self.sin, pos, self.t_qkv, None, self.t_delta)):
                K.flash_attn_bwd(d_ctx.view(T, H, d), self.a_q[l], self.a_kv[l][:, 0], self.a_kv[l][:, 1], self.a_ctx[l], self.a_lse[l], cu, max_seqlen, None, True,
                                 self.t_dq, self.t_dkv[:, 0], self.t_dkv[:, 1], self.t_delta)
                K.qkv_rotary_bwd(self.t_dq, self.t_dkv, self.cos, self.sin, pos, H, 1, d, False, self.t_qkv)
            self._bias_grad(self.t_qkv, g[pre + "mixer.Wqkv.bias"], self.t_bias3, acc)
            d_n1 = d_n2
            K.linear_dgrad(self.t_qkv, p[pre + "mixer.Wqkv.weight"], d_n1)
            ar = tpar.all_reduce_sum_async(d_n1)
            K.linear_wgrad(self.t_qkv, self.a_n1[l], g[pre + "mixer.Wqkv.weight"], acc)
            ar.wait()
            d_x = d_h
            K.rmsnorm_bwd(d_n1, self.a_x[l], p[pre + "norm1.weight"], self.a_rstd1[l], d_r2, g[pre + "norm1.weight"], acc, ws, d_x)
            spare = [d_n2, d_r2]
            d_h = d_x
        K.embedding_bwd(d_h, ids, g["embedding.weight"], acc, self.t_emb_ws)

    def _bias_grad(self, dy, gb, tmp, acc):
        if acc:
            K.colsum(dy, tmp)
            K.add_bf16(gb, tmp, gb)
        else:
            K.colsum(dy, gb)

    def _wait_runs(self, keys):
        """The current stream waits for the AdamW launches of the given runs (no-op when no step is in flight)."""
        if self._opt_events is not None:
            cur = torch.cuda.current_stream()
            for k in keys:
                cur.wait_event(self._opt_events[k])

    def _wait_optimizer(self):
        """Everything the last step() queued on the optimizer stream is ordered in front of what the current stream does next."""
        if self._opt_events is not None:
            torch.cuda.current_stream().wait_stream(self.opt_stream)
            self._opt_events = None

    def forward_backward(self, batch, labels):
        """One NonPipelineScheduler.forward_backward_step.  Returns (loss incl. the moe loss, moe loss) as device scalars."""
        tc, mc = self.tc, self.mc
        M = batch["input_ids"].shape[0]
        assert M == tc.micro_num and batch["input_ids"].shape[1] == self.T
        self.loss_acc.zero_()
        self.moe_acc.zero_()
        ids_d = batch["input_ids"].to(self.dev, non_blocking=True)
        lab_d = labels.to(self.dev, non_blocking=True)
        pos_d = batch["indexes"].to(self.dev, non_blocking=True)
        for i in range(M):
            cu_h = batch["cu_seqlens"][i]
            max_seqlen = int((cu_h[1:] - cu_h[:-1]).max())
            cu = cu_h.to(self.dev, non_blocking=True)
            self._forward_micro(ids_d[i], lab_d[i], cu, pos_d[i], max_seqlen)
            if self.dense:   # no auxiliary loss
                self.loss_acc.add_(self.t_loss[0:1] / M)
                self._backward_micro(ids_d[i], lab_d[i], cu, pos_d[i], max_seqlen, i > 0, 1.0 / M)
                continue
            if self.keep_routes is not None:   # diagnostics / parity tests: the discrete decisions of this micro-batch, layer by layer
                self.keep_routes.append([(m.expert.clone(), m.logits.clone()) for m in self.moe])
            # moe loss: sum of the layers' l_aux (model-dtype scalars) * coeff / micro_num, in the model dtype (no_pipeline_scheduler.py:134-145)
            s = self.l_aux[0].to(BF16)
            for la in self.l_aux[1:]:
                s = s + la.to(BF16)
            moe = ((s * mc.moe_loss_coeff) / M).float()
            self.moe_acc.add_(moe)
            self.loss_acc.add_(self.t_loss[0:1] / M + moe)
            self._backward_micro(ids_d[i], lab_d[i], cu, pos_d[i], max_seqlen, i > 0, 1.0 / M)
        return self.loss_acc, self.moe_acc

    # ------------------------------------------------------------------------------------------ optimizer
    def _all_reduce(self, t, group, size, avg=True):
        """In-place all-reduce (AVG, or SUM) over `group` of `size` ranks (comm.Backend)."""
        if size == 1:
            return
        self.be.all_reduce(t, group, avg=avg).wait()

    def sync_replicas(self):
        """sync_model_param (utils/parallel.py:71-107): dense parameters and gates from rank 0, every expert from the first rank of its
        expert-data group."""
        if self.world == 1:
            return
        self._wait_optimizer()

        def bcast(t, src, group):
            self.be.broadcast(t, src, group).wait()

        # (tensor parallelism: a shard is replicated over the ranks of its data-parallel group, whose first rank is the global rank tp_rank; an expert
        # over its expert-data group, whose first rank holds the same position in the first expert group)
        if self.dp_world > 1:
            for a, b in self.runs[0]:
                bcast(self.params[a:b], self.tp_rank, self.dp_group)
            bcast(self.wg, self.tp_rank, self.dp_group)
        if self.dp_world > self.ep:
            for a, b in self.runs[2]:
                bcast(self.params[a:b], self.tp_rank + self.ep_rank * self.tp, self.edp_group)
        self.master.copy_(self.params)

    def step(self):
        """HybridZeroOptimizer.step with the three parameter groups; stream-ordered, no host sync."""
        tc = self.tc
        for lay in self.moe or ():
            lay.invalidate_fp8()   # (expert_fp8: the experts' weights are about to change)
        # data parallel, every rank keeps the (replicated) optimizer state of what it holds.  Dense parameters and gates: AVG over all ranks.
        # Experts: an expert's gradient already sums what the tokens of every rank of its expert group contributed (one copy, one backward);
        # the reference averages it over the expert-data group only (hybrid_zero_optim.py:166-167) -- no 1 / ep.
        dpw = self.dp_world
        for a, b in self.runs[0]:
            self._all_reduce(self.grads[a:b], self.dp_group, dpw)
        self._all_reduce(self.d_wg, self.dp_group, dpw)
        for a, b in self.runs[2]:
            self._all_reduce(self.grads[a:b], self.edp_group, dpw // self.ep)
        if self.tp == 1:
            K.sumsq([self.grads[a:b] for a, b in self.runs[0]], self.sumsq[0:1])
        else:   # what is cut over the tensor group: summed over it; what every rank holds whole: once (solver/optimizer/utils.py:225-262,330-352)
            K.sumsq(self.cut0, self.sumsq[0:1])
        if self.runs[2]:
            K.sumsq([self.grads[a:b] for a, b in self.runs[2]], self.sumsq[2:3])
        else:
            self.sumsq[2:3].zero_()   # (the dense model has no expert group)
        if self.tp > 1:
            self.sumsq[1:2].zero_()
            self.tpar.all_reduce_sum(self.sumsq)
            K.sumsq(self.whole0, self.sumsq[0:1], accumulate=True)
        K.sumsq(self.d_wg, self.sumsq[1:2])
        if self.ep > 1:   # the moe group's squared norm: local sum of squares / dp, summed over the expert group (solver/optimizer/utils.py:362-368)
            self.sumsq[2:3].div_(dpw)
            self._all_reduce(self.sumsq[2:3], self.ep_group, self.ep, avg=False)
        check(K._L().ie_step_control_groups(K._p(self.state), K._p(self.sumsq), 3, self.scaler_cfg, K._p(self.group_inv), K._p(self.group_norm), K._stream()),
              "ie_step_control_groups")
        lr, beta2 = self.lr_sched.lr(), self.beta2_sched.beta2()
        L_ = K._L()

        def adam(gr, p32, m, v, p16, group):
            check(L_.ie_adamw_step_group(K._p(gr), K._dt(gr), K._p(p32), K._p(m), K._p(v), K._p(p16), p32.numel(), K._p(self.state),
                                         K._p(self.group_inv[group : group + 1]), lr, tc.adam_beta1, beta2, tc.adam_eps, tc.weight_decay, K._stream()),
                  "ie_adamw_step_group")

        # on the optimizer stream, in the order the next forward pass needs the parameters (gates, then run by run from the embedding up)
        self._wait_optimizer()
        self.opt_stream.wait_stream(torch.cuda.current_stream())
        events = {}
        with torch.cuda.stream(self.opt_stream):
            if not self.dense:
                adam(self.d_wg.view(-1), self.wg.view(-1), self.wg_m.view(-1), self.wg_v.view(-1), None, 1)
            events["gate"] = torch.cuda.Event()
            events["gate"].record(self.opt_stream)
            # (as engine.step(): the first runs over the whole chip -- the next forward has nothing to do until they are done --, the others on CUs of their own
            # beside it: ie_tune_adamw_cus, IE_ADAMW_CUS, default 128; same results bit for bit)
            cus, full = int(os.environ.get("IE_ADAMW_CUS", "128") or 0), int(os.environ.get("IE_ADAMW_FULL_BUCKETS", "2") or 0)
            for i, (a, b, grp) in enumerate(sorted((a, b, grp) for grp in (0, 2) for a, b in self.runs[grp])):
                if cus:
                    K.tune_adamw_cus(cus if i >= full else 0)
                adam(self.grads[a:b], self.master[a:b], self.exp_avg[a:b], self.exp_avg_sq[a:b], self.params[a:b], grp)
                events[a] = torch.cuda.Event()
                events[a].record(self.opt_stream)
            if cus:
                K.tune_adamw_cus(0)
        self._opt_events = events
        self.lr_sched.step()
        self.beta2_sched.step()
        self.step_count += 1

    # ---- checkpoints (the dense model): InternEvo's files, internevo_amd/checkpoint.py -------------------------------------------------------
    def _checkpoint_guard(self):
        # (any data-parallel size since round 4: checkpoint.save_moe_checkpoint / load_moe_checkpoint)
        if self.tp > 1 and self.dense:
            raise NotImplementedError("MoEEngine: checkpoints of the dense InternLM-1 model under tensor parallelism go through InternLM2Engine (model_type INTERNLM)")

    def _reference_local(self, named):
        """named_parameters() output -> what tensor rank tp_rank of the reference holds: this engine keeps the embedding whole on every tensor rank (the reference
        cuts its hidden columns) and out_proj's bias on every rank (the reference: tensor rank 0 only, ops/linear.py:317-324)."""
        if self.tp == 1:
            return dict(named)
        out = {}
        for n, t in named:
            if n == "embedding.weight":
                t = self._cut(t, 1)
            elif n.endswith("mixer.out_proj.bias") and self.tp_rank:
                continue
            out[n] = t
        return out

    def save_checkpoint(self, folder):
        """model_tp0_pp0.pt + the hybrid-ZeRO optimizer shards in the reference's whole-parameter partition (hybrid_zero_optim.py:254-284): this
        engine keeps the optimizer state replicated on every data-parallel rank, so rank r writes the reference's partition r of it (a W-rank
        reference job, or this engine at any world size, resumes from the folder).  Collective."""
        from . import checkpoint as C

        self._checkpoint_guard()
        st = self.read_state()
        self._wait_optimizer()
        torch.cuda.synchronize(self.dev)
        tc, W, r = self.tc, self.world, self.rank
        if not self.dense:   # the MoE model: the model file without the experts, one file per expert, three optimizer groups (checkpoint.save_moe_checkpoint)
            hyper = dict(weight_decay=tc.weight_decay, betas=(tc.adam_beta1, tc.adam_beta2), eps=tc.adam_eps, initial_lr=tc.lr)
            scaler = dict(scale=st.loss_scale, growth_step=st.growth_step, hysteresis_step=st.hysteresis_step)
            cpu = lambda views, gates: {n: t.detach().to("cpu", copy=True) for n, t in self._reference_local(self.named_parameters(views, gates)).items()}  # noqa: E731
            # every data-parallel rank writes what the reference's rank would (this engine keeps the optimizer state of the dense parameters and the gates on
            # every rank, and of its own experts: data rank d cuts the reference's partition d out of it), under tensor parallelism every tensor rank its own files
            # (round 5; pinned on tests/golden/ckpt_ref_moe_tp2dp2/); collective
            dpw, dpr = self.dp_world, self.tpar.dp_rank
            if r == 0:
                C.remove_stale_shards(folder, dpw, self.tp, layout="moe", num_experts=self.mc.num_experts, num_layers=self.mc.num_layers)
            if W > 1:
                dist.barrier(group=self.group)
            C.save_moe_checkpoint(folder, self.mc, cpu(None, None), cpu(self._views(self.master), self.wg), cpu(self._views(self.exp_avg), self.wg_m),
                                  cpu(self._views(self.exp_avg_sq), self.wg_v), st.adam_step, scaler, self.lr_sched.lr(), hyper, world=dpw, rank=dpr,
                                  tp_world=self.tp, tp_rank=self.tp_rank)
            if W > 1:
                dist.barrier(group=self.group)
            return
        if r == 0:
            C.remove_stale_shards(folder, W, 1)
        if W > 1:
            dist.barrier(group=self.group)
        hyper = dict(weight_decay=tc.weight_decay, betas=(tc.adam_beta1, tc.adam_beta2), eps=tc.adam_eps, initial_lr=tc.lr)
        scaler = dict(scale=st.loss_scale, growth_step=st.growth_step, hysteresis_step=st.hysteresis_step)
        cpu = lambda views: {n: t.detach().to("cpu", copy=True) for n, t in self.named_parameters(views)}  # noqa: E731
        C.save_checkpoint(folder, self.mc, cpu(None), cpu(self._views(self.master)), cpu(self._views(self.exp_avg)), cpu(self._views(self.exp_avg_sq)),
                          st.adam_step, scaler, self.lr_sched.lr(), hyper, zero_world=W, zero_ranks=[r], write_model=(r == 0))
        if W > 1:
            dist.barrier(group=self.group)

    def load_checkpoint(self, folder):
        """Resume from InternEvo checkpoint files written by the reference or by save_checkpoint at ANY ZeRO-1 world (the shards are merged)."""
        from . import checkpoint as C

        self._checkpoint_guard()
        ck = C.load_checkpoint(folder, self.mc) if self.dense else C.load_moe_checkpoint(folder, self.mc)
        self._wait_optimizer()
        self.load_named_parameters(ck["params"], sync_master=ck["master"] is None)
        if ck["master"] is None:
            return
        scratch = torch.empty_like(self.wg)   # (the gates are fp32 parameters: their master copy IS the parameter, already loaded)
        for flat, gates, key in ((self.master, scratch, "master"), (self.exp_avg, self.wg_m, "exp_avg"), (self.exp_avg_sq, self.wg_v, "exp_avg_sq")):
            self.load_named_parameters(ck[key], views=self._views(flat), gates=gates)
        st = K.step_state_read(self.state)
        st.loss_scale, st.growth_step, st.hysteresis_step = ck["scaler"]["scale"], ck["scaler"]["growth_step"], ck["scaler"]["hysteresis_step"]
        st.adam_step, st.skip, st.found_inf, st.found_nan = ck["adam_step"], 0, 0, 0
        self.state.copy_(torch.frombuffer(bytearray(bytes(st)), dtype=torch.uint8).to(self.dev))
        self.lr_sched.set_successful_steps(ck["adam_step"])
        self.beta2_sched.set_successful_steps(ck["adam_step"])
        self.step_count = ck["adam_step"]

    def read_state(self):
        st = K.step_state_read(self.state)
        self.lr_sched.set_successful_steps(st.adam_step)
        self.beta2_sched.set_successful_steps(st.adam_step)
        st.group_norms = dict(zip(self.groups, (float(x) for x in self.group_norm.cpu())))
        return st
