"""TEST INFRASTRUCTURE -- CPU oracle of one full training step (scheduler + hybrid-ZeRO optimizer), single rank.

Restates, in plain torch on CPU:
  NonPipelineScheduler._train_one_batch / forward_backward_step   core/scheduler/no_pipeline_scheduler.py:95-239
  HybridZeroOptimizer.backward / step / _step / _unscale_and_clip   solver/optimizer/hybrid_zero_optim.py:592-876
  compute_norm (zero world 1)                                      solver/optimizer/utils.py:265-378
  Engine.step (schedulers stepped only after a successful update)   core/engine.py:105-126
  FineTuneCosineAnnealingWarmupLR via torch's own scheduler classes  solver/schedulers/lr_scheduler.py
Pinned against the real reference by tests/test_oracle_golden.py.
"""
import math

import torch

from . import ops as O
from .model import build_params, forward_logits, micro_loss


class OracleTrainer:
    def __init__(self, path_cfg, dtype=torch.bfloat16, init_fn=None):
        from .model import formula_init

        self.mc, self.tc = path_cfg.model, path_cfg.train
        self.dtype = dtype
        self.params = build_params(self.mc, dtype, init_fn or formula_init)
        self.names = list(self.params.keys())
        # fp32 master / Adam state (hybrid_zero_optim.py:214-233: master = fp16 flat .clone().float())
        self.master = {n: p.detach().clone().float() for n, p in self.params.items()}
        self.m = {n: torch.zeros_like(t) for n, t in self.master.items()}
        self.v = {n: torch.zeros_like(t) for n, t in self.master.items()}
        tc = self.tc
        fp32 = dtype == torch.float32
        # BaseGradScaler: fp32 models use scale 1 (hybrid_zero_optim.py:70-73)
        self.scaler = O.DynamicGradScaler(1.0 if fp32 else tc.initial_scale, tc.growth_factor, tc.backoff_factor, tc.growth_interval,
                                          tc.min_scale, tc.max_scale, tc.hysteresis)
        self.adam_step = 0
        self.k = 0  # successful steps (drives the lr schedule)
        self.beta2_iter = 0
        self.metric = None  # optional oracle.ops.AccPerplexOracle, fed every micro-batch's logits (SchedulerMetricHook)

    # ---- checkpoint hand-over (internevo_amd/checkpoint.py dict form; load_checkpoint / save_checkpoint of the reference format)
    def load_state(self, ck):
        with torch.no_grad():
            for n in self.names:
                self.params[n].copy_(ck["params"][n].to(self.params[n].dtype))
                if ck["master"] is not None:
                    self.master[n].copy_(ck["master"][n])
                    self.m[n].copy_(ck["exp_avg"][n])
                    self.v[n].copy_(ck["exp_avg_sq"][n])
                else:
                    self.master[n].copy_(self.params[n].float())
        if ck["adam_step"] is not None:
            self.adam_step = self.k = self.beta2_iter = int(ck["adam_step"])  # no skipped step so far: successful == total
            sc = ck["scaler"]
            self.scaler.scale, self.scaler.growth_step, self.scaler.hysteresis_step = float(sc["scale"]), int(sc["growth_step"]), int(sc["hysteresis_step"])

    def export_state(self):
        return dict(params={n: self.params[n].detach().clone() for n in self.names}, master={n: t.clone() for n, t in self.master.items()},
                    exp_avg={n: t.clone() for n, t in self.m.items()}, exp_avg_sq={n: t.clone() for n, t in self.v.items()},
                    adam_step=self.adam_step, lr=self._lr(),
                    scaler=dict(scale=self.scaler.scale, growth_step=self.scaler.growth_step, hysteresis_step=self.scaler.hysteresis_step))

    # lr exactly as the reference produces it: torch's scheduler objects driven the same way
    def _lr(self):
        tc = self.tc
        warm = int(tc.total_steps * tc.warmup_ratio) + tc.init_steps
        k = self.k
        if k >= warm:
            tmax = tc.total_steps - warm
            # hand-over quirk of WarmupScheduler + torch's recursive CosineAnnealingLR (see tests/golden lr_trace)
            return tc.eta_min + (tc.lr - tc.eta_min) * (1 + math.cos(math.pi * (k - warm) / tmax)) / (1 + math.cos(math.pi / tmax))
        if k >= tc.init_steps:
            return (k + 1 - tc.init_steps) / int(tc.total_steps * tc.warmup_ratio) * tc.lr
        return 0.0

    def _beta2(self):
        tc = self.tc
        if tc.adam_beta2_c <= 0 or self.beta2_iter == 0:
            return tc.adam_beta2
        return max(tc.adam_beta2, 1 - (1 / self.beta2_iter**tc.adam_beta2_c))

    def eval_batch(self, input_ids, labels, metric=None):
        """forward_only=True step of the evaluation loop (eval/evaluation.py:84-108): rows are whole sequences, micro-batches of
        micro_bsz rows, loss = mean over micro-batches; `metric` (an AccPerplexOracle) sees the logits."""
        tc, mc = self.tc, self.mc
        B, S = input_ids.shape
        M = B // tc.micro_bsz
        total = 0.0
        with torch.no_grad():
            for i in range(M):
                ids = input_ids[i * tc.micro_bsz : (i + 1) * tc.micro_bsz].reshape(-1)
                lab = labels[i * tc.micro_bsz : (i + 1) * tc.micro_bsz].reshape(-1)
                cu = torch.arange(tc.micro_bsz + 1, dtype=torch.int32) * S
                idx = torch.arange(S, dtype=torch.int64).repeat(tc.micro_bsz)
                logits = forward_logits(self.params, mc, ids, idx, cu)
                total += float(O.cross_entropy(logits, lab, tc.label_smoothing)) / M
                if metric is not None:
                    metric.update(logits.float(), lab, None)
        return total

    def train_step(self, batch, labels):
        """batch['input_ids'] [micro_num, T]; returns dict(loss, grad_norm, ok, loss_scale)."""
        tc, mc = self.tc, self.mc
        for p in self.params.values():
            p.grad = None
        M = batch["input_ids"].shape[0]
        total = 0.0
        for i in range(M):
            cu = batch["cu_seqlens"][i] if "cu_seqlens" in batch and batch["cu_seqlens"] is not None else None
            idx = batch["indexes"][i] if "indexes" in batch and batch["indexes"] is not None else None
            if self.metric is None:
                loss = micro_loss(self.params, mc, batch["input_ids"][i], labels[i], idx, cu, tc.label_smoothing)
            else:
                logits = forward_logits(self.params, mc, batch["input_ids"][i], idx, cu)
                loss = O.cross_entropy(logits, labels[i], tc.label_smoothing)
                tid = batch["type_ids"][i] if batch.get("type_ids", None) is not None else None
                self.metric.update(logits.detach().float(), labels[i], tid)
            loss = loss / M                       # `loss /= scale_loss` (no_pipeline_scheduler.py:146)
            total += float(loss.detach())
            (self.scaler.scale * loss).backward()  # HybridZeroOptimizer.backward :592-594
        # compute_norm: sum of squared fp32-cast grads (norm_type 2), inf -> -1, nan -> -2
        sq = 0.0  # calc_lp (utils.py:207-212): `norm += grad_norm ** norm_type` accumulates an fp32 tensor
        for n in self.names:
            gnorm = torch.norm(self.params[n].grad.float(), 2.0)
            sq = sq + gnorm**2.0
        sq = float(sq)
        found_inf, found_nan = math.isinf(sq), math.isnan(sq)
        loss_scale = self.scaler.scale
        if self.dtype != torch.float32:
            self.scaler.update(found_inf)
        if found_inf or found_nan:
            return {"loss": total, "grad_norm": -1.0 if found_inf else -2.0, "ok": False, "loss_scale": self.scaler.scale}
        global_norm = sq**0.5
        inv = 1.0
        if self.dtype != torch.float32 and tc.clip_grad_norm > 0:
            inv = 1.0 / O.unscale_clip_factor(global_norm, loss_scale, tc.clip_grad_norm)
        lr, beta2 = self._lr(), self._beta2()
        self.adam_step += 1
        with torch.no_grad():
            for n in self.names:
                g = self.params[n].grad.float()
                g.mul_(inv)
                O.adamw_step(self.master[n], g, self.m[n], self.v[n], self.adam_step, lr, tc.adam_beta1, beta2, tc.adam_eps, tc.weight_decay)
                self.params[n].copy_(self.master[n])
        self.k += 1
        self.beta2_iter += 1
        return {"loss": total, "grad_norm": global_norm / loss_scale, "ok": True, "loss_scale": self.scaler.scale, "lr": lr}
