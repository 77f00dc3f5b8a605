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

    def state_dict(self, n_groups=2):
        """The reference scheduler's state_dict() after self.k steps -- the content of a checkpoint's `schedulder.pt`
        (lr_scheduler.py:28-37: the wrapper's __dict__ plus the after-scheduler's; n_groups = the optimizer's parameter groups,
        "default" and "fp32").  While warming up the wrapper counts the steps, from the hand-over on only the CosineAnnealingLR
        does.  Pinned by tests/golden/sched_state.json (the real class, torch 2.10 key set)."""
        k, w = self.k, self.warmup_epochs
        base = [self.base_lr] * n_groups
        now = [self.lr_at(k)] * n_groups
        return {
            "_init_steps": self.init_steps, "_warmup_steps": self.warmup_steps, "warmup_epochs": w, "finished": k >= w,
            "base_lrs": list(base), "last_epoch": min(k, w), "_step_count": 1 + min(k, w), "_is_initial": False,
            "_get_lr_called_within_step": False, "_last_lr": list(now), "after_scheduler_type": "CosineAnnealingLR",
            "after_scheduler_dict": {"T_max": self.t_max, "eta_min": self.eta_min, "base_lrs": list(base), "last_epoch": max(0, k - w),
                                     "_step_count": 1 + max(0, k - w), "_is_initial": False,
                                     "_get_lr_called_within_step": k == w,  # set by hand on the hand-over step (:124), reset by its next step()
                                     "_last_lr": list(now if k > w else base)},
        }

    def load_state_dict(self, state):
        """Position from a `schedulder.pt`: steps taken = the wrapper's count + the after-scheduler's.  (The reference's
        load_scheduler additionally overwrites the wrapper's last_epoch with step_count + 1, components.py:446 -- inside the
        warm-up that moves a resumed reference run one step ahead of an uninterrupted one; not mirrored.)"""
        for key, mine in (("_init_steps", self.init_steps), ("_warmup_steps", self.warmup_steps)):
            if key in state and state[key] != mine:
                raise ValueError(f"scheduler checkpoint has {key} = {state[key]}, the config gives {mine}")
        after = state.get("after_scheduler_dict", {})
        if "T_max" in after and after["T_max"] != self.t_max:
            raise ValueError(f"scheduler checkpoint has T_max = {after['T_max']}, the config gives {self.t_max}")
        self.k = int(state["last_epoch"]) + int(after.get("last_epoch", 0))


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
