"""Host-side scalar schedules of the step (pure Python; no device work).

CosineWarmupLR  = internlm/solver/schedulers/lr_scheduler.py:92-131 FineTuneCosineAnnealingWarmupLR
                  (WarmupScheduler :10-72 around torch's CosineAnnealingLR), in closed form:
                  k = number of successful optimizer steps so far
                    k <  init_steps                       : 0
                    k <  warmup (= init + int(total*ratio)): (k + 1 - init_steps) / int(total*ratio) * lr
                    else                                  : eta_min + (lr - eta_min) * (1 + cos(pi * j / T_max)) / 2,
                                                            j = k - warmup, T_max = total_steps - warmup
                  (torch's CosineAnnealingLR uses the recursive form; both agree to double rounding,
                  pinned by tests/golden/ops.json "lr_trace").
Beta2Scheduler  = internlm/solver/schedulers/beta2_scheduler.py:8-37.
"""
import math


class CosineWarmupLR:
    def __init__(self, base_lr, total_steps, warmup_ratio=0.0, eta_min=0.0, init_steps=0):
        self.base_lr, self.total_steps, self.eta_min, self.init_steps = base_lr, total_steps, eta_min, init_steps
        self.warmup_steps = int(total_steps * warmup_ratio)
        self.warmup_epochs = self.warmup_steps + init_steps
        self.t_max = total_steps - self.warmup_epochs
        self.k = 0

    def lr_at(self, k):
        if k >= self.warmup_epochs:
            j = k - self.warmup_epochs
            # The reference calls after_scheduler.get_lr() OUTSIDE after_scheduler.step() on the hand-over step
            # (lr_scheduler.py:63-68); with torch >= 2.x CosineAnnealingLR then evaluates its recursive formula at
            # last_epoch = 0, so the whole cosine is scaled by 2 / (1 + cos(pi / T_max)) and reaches base_lr at
            # j = 1 instead of j = 0.  Pinned by tests/golden/ops.json["lr_trace"] (generated from the real classes).
            return self.eta_min + (self.base_lr - self.eta_min) * (1 + math.cos(math.pi * j / self.t_max)) / (
                1 + math.cos(math.pi / self.t_max)
            )
        if k >= self.init_steps:
            return (k + 1 - self.init_steps) / self.warmup_steps * self.base_lr
        return 0.0

    def lr(self):
        return self.lr_at(self.k)

    def step(self):
        self.k += 1

    def set_successful_steps(self, k):
        self.k = k


class Beta2Scheduler:
    def __init__(self, init_beta2, c=0.0):
        self.init_beta2, self.c = init_beta2, c
        self.cur_iter = 0

    def beta2(self):
        if self.c <= 0 or self.cur_iter == 0:
            return self.init_beta2
        return max(self.init_beta2, 1 - (1 / self.cur_iter**self.c))

    def step(self):
        self.cur_iter += 1

    def set_successful_steps(self, k):
        self.cur_iter = k
