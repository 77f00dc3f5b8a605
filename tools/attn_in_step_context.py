"""Why is the attention forward 10-15 % slower inside the training step than in a loop of its own?  The bench call (4 x 4096
tokens, 32 / 8 heads, d 128, causal) timed with HIP events around the attention launch only, (a) back to back, (b) each launch behind the
layer's own neighbours in the step (wqkv GEMM before, wo + w1|w3 GEMMs after), (c) the same with a 2 ms idle gap (host sleep) before every
attention launch.  If (b) is slower than (a) and (c) recovers, the chip's clock under the GEMMs' power draw is what the step costs."""
import json
import os
import statistics
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from internevo_amd import kernels as K  # noqa: E402

dev = torch.device("cuda:0")
bf = torch.bfloat16
T, HQ, HKV, D, H, F = 16384, 32, 8, 128, 4096, 14336
g = torch.Generator(device=dev).manual_seed(3)
q = torch.randn(T, HQ, D, device=dev, generator=g).to(bf)
kv = torch.randn(T, 2, HKV, D, device=dev, generator=g).to(bf)
cu = torch.arange(0, T + 1, 4096, dtype=torch.int32, device=dev)
x = torch.randn(T, H, device=dev, generator=g).to(bf)
wqkv = (torch.randn(6144, H, device=dev, generator=g) * 0.02).to(bf)
wo = (torch.randn(H, H, device=dev, generator=g) * 0.02).to(bf)
w13 = (torch.randn(2 * F, H, device=dev, generator=g) * 0.02).to(bf)
y1, y2, y3 = torch.empty(T, 6144, device=dev, dtype=bf), torch.empty(T, H, device=dev, dtype=bf), torch.empty(T, 2 * F, device=dev, dtype=bf)
out, lse = torch.empty(T, HQ, D, device=dev, dtype=bf), torch.empty(HQ, T, device=dev, dtype=torch.float32)


def attn():
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    K.flash_attn_fwd(q, kv[:, 0], kv[:, 1], cu, 4096, None, True, out, lse)
    e.record()
    return s, e


def run(mode, n=24):
    ev = []
    for _ in range(n):
        if mode != "alone":
            K.linear_fwd(x, wqkv, y1)
        if mode == "gemms_then_idle":
            torch.cuda.synchronize()
            time.sleep(0.002)
        ev.append(attn())
        if mode != "alone":
            K.linear_fwd(x, wo, y2)
            K.linear_fwd(x, w13, y3)
    torch.cuda.synchronize()
    t = [s.elapsed_time(e) * 1e3 for s, e in ev][4:]
    return {"mode": mode, "median_us": round(statistics.median(t), 1), "min_us": round(min(t), 1), "max_us": round(max(t), 1)}


for mode in ("alone", "between_gemms", "gemms_then_idle", "alone", "between_gemms"):
    print(json.dumps(run(mode)), flush=True)
