"""What the GEMM epilogue costs: the 7B products at 16 384 rows with the epilogue compiled out / only its global stores skipped
(IE_GEMM_ABLATE 8 / 12, read by the library at first launch -- one process per setting; results are then wrong).
(Round 3 also tried offsetting the first round's blocks by up to 8 ... 64 us so that the CUs' epilogues stop coinciding -- a temporary
ie_tune_gemm_stagger hook, columns stagger1..8 of profiles/r03_gemm_epilogue_probe.jsonl: 3 ... 8 % SLOWER on every product.  Blocks of an XCD running
in lockstep share their operand panels in L2; that is worth more than the exposed epilogue costs.)"""
import json
import os
import statistics
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from internevo_amd import kernels as K  # noqa: E402

dev = torch.device("cuda:0")
bf = torch.bfloat16
T, F = 16384, 14336
units = [0]


def t_once(fn, iters):
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) * 1e-3 / iters


shapes = [("wqkv", 6144, 4096), ("wo", 4096, 4096), ("w13", 2 * F, 4096), ("w2", 4096, F)]
for name, N, Kd in shapes:
    X = torch.randn(T, Kd, device=dev).to(bf)
    W = (torch.randn(N, Kd, device=dev) * 0.02).to(bf)
    DY = torch.randn(T, N, device=dev).to(bf)
    Y = torch.empty(T, N, device=dev, dtype=bf)
    DX = torch.empty(T, Kd, device=dev, dtype=bf)
    DW = torch.zeros(N, Kd, device=dev, dtype=bf)
    fl = 2.0 * T * N * Kd
    kinds = {"fwd": lambda: K.gemm(X, W, False, False, Y), "dgrad": lambda: K.gemm(DY, W, False, True, DX), "wgrad": lambda: K.gemm(DY, X, True, True, DW)}
    for kind, fn in kinds.items():
        times = {u: [] for u in units}
        for _ in range(5):
            for u in units:
                times[u].append(t_once(fn, 5))
        rec = {"gemm": name, "kind": kind, "ablate": int(os.environ.get("IE_GEMM_ABLATE", "0"))}
        for u in units:
            t = statistics.median(times[u])
            rec[f"stagger{u}_us"] = round(t * 1e6, 1)
            rec[f"stagger{u}_tf"] = round(fl / t / 1e12, 1)
        print(json.dumps(rec), flush=True)
