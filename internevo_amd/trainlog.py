"""Per-step bookkeeping of the training loop: the metric the benchmark is quoted in.

Host-side mirror of `record_current_batch_training_metrics` (internlm/train/pipeline.py:464-600), the TGS windows of `TrainState`
(internlm/core/trainer.py:62-78) and `get_megatron_flops` (internlm/utils/common.py:208-238): same quantities, same rounding, same
key order of the log line, so a reference user reads the same numbers.
"""
import collections
import time


def get_megatron_flops(elapsed_time_per_iter, checkpoint=False, seq_len=2048, hidden_size=12, num_layers=32, vocab_size=12, global_batch_size=4,
                       global_world_size=1, mlp_ratio=4, use_swiglu=True):
    """TFLOPS per GPU by the Megatron formula (the factor 4 counts the recomputed forward under activation checkpointing)."""
    fac = 4 if checkpoint else 3
    if use_swiglu:
        mlp_ratio = mlp_ratio * 3 / 2
    per_layer = fac * ((8 + mlp_ratio * 4) * global_batch_size * seq_len * hidden_size**2 + 4 * global_batch_size * seq_len**2 * hidden_size)
    flops = per_layer * num_layers + 6 * global_batch_size * seq_len * hidden_size * vocab_size
    return flops / (elapsed_time_per_iter * global_world_size * (10**12))


class TgsStatistic:
    """tokens / GPU / second over the windows the reference reports: last step, running, mean of steps, 50-step SMA, last 10 / 50."""

    def __init__(self):
        self.sum_step = 0
        self.sum_tg = self.sum_time = 0.0
        self.sum_last_tg_10 = self.sum_last_time_10 = 0.0
        self.sum_last_tg_50 = self.sum_last_time_50 = 0.0
        self.sma_tg = self.sma_time = 0.0
        self.sma_tg_list, self.sma_time_list = collections.deque(), collections.deque()
        self.sum_tgs = 0.0
        self.last_tgs_10 = self.last_tgs_50 = 0

    def update(self, tk_per_gpu, time_cost):
        self.sum_step += 1
        self.sum_tg += tk_per_gpu
        self.sum_time += time_cost
        self.sum_last_tg_10 += tk_per_gpu
        self.sum_last_time_10 += time_cost
        self.sum_last_tg_50 += tk_per_gpu
        self.sum_last_time_50 += time_cost
        self.sma_tg += tk_per_gpu
        self.sma_time += time_cost
        self.sma_tg_list.append(tk_per_gpu)
        self.sma_time_list.append(time_cost)
        if self.sum_step > 50:
            self.sma_tg -= self.sma_tg_list.popleft()
            self.sma_time -= self.sma_time_list.popleft()
        last_tgs_1 = round(tk_per_gpu / time_cost, 2)
        self.sum_tgs += last_tgs_1
        if self.sum_step % 10 == 0:
            self.last_tgs_10 = round(self.sum_last_tg_10 / self.sum_last_time_10, 2)
            self.sum_last_tg_10 = self.sum_last_time_10 = 0.0
        if self.sum_step % 50 == 0:
            self.last_tgs_50 = round(self.sum_last_tg_50 / self.sum_last_time_50, 2)
            self.sum_last_tg_50 = self.sum_last_time_50 = 0.0
        return {
            "tgs/last_tgs_1": last_tgs_1,
            "tgs/tgs_all": round(self.sum_tg / self.sum_time, 2),
            "tgs/tgs_avg": round(self.sum_tgs / self.sum_step, 2),
            "tgs/tgs_SMA": round(self.sma_tg / self.sma_time, 2),
            "tgs/last_tgs_10": self.last_tgs_10,
            "tgs/last_tgs_50": self.last_tgs_50,
        }


def step_infos(*, tflops, step, loss, tk_per_gpu, start_time, tgs: TgsStatistic, lr, loss_scale, grad_norm, batch, labels, num_consumed_tokens,
               inf_nan_skip_batches, adam_beta2, fwd_bwd_time, metric):
    """The ordered key -> value record of one step (pipeline.py:556-590); `line(infos)` renders it like the reference's logger."""
    time_cost = time.time() - start_time
    windows = tgs.update(tk_per_gpu, time_cost)
    cu = batch["cu_seqlens"]
    infos = {"tflops": tflops, "step": step, "loss": loss, "tgs (tokens/gpu/second)": round(tk_per_gpu / (time.time() - start_time), 2)}
    infos.update(windows)
    infos.update({"lr": lr, "loss_scale": loss_scale, "grad_norm": grad_norm, "micro_num": len(labels), "num_consumed_tokens": num_consumed_tokens,
                  "inf_nan_skip_batches": inf_nan_skip_batches, "num_samples_in_batch": sum(len(b) - 1 for b in cu),
                  "largest_length": max(int((b[1:] - b[:-1]).max()) for b in cu), "largest_batch": max(len(b) - 1 for b in cu),
                  "smallest_batch": min(len(b) - 1 for b in cu), "adam_beta2": adam_beta2, "fwd_bwd_time": round(fwd_bwd_time, 2)})
    infos.update(metric)
    return infos


def line(infos):
    return "".join(f"{k}={v} " for k, v in infos.items())
