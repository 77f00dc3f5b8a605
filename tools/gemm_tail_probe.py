"""A/B of the automatic GEMM's tail split (ie_tune_gemm_tail_split) on the 7B layer shapes: interleaved rounds, median."""
import json
import os
import statistics
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from internevo_amd import kernels as K  # noqa: E402

dev = torch.device("cuda:0")
T, F, V = 4096, 14336, 92544
bf = torch.bfloat16
MODES = (0, 1, 2)


def t_once(fn, iters):
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) * 1e-3 / iters


shapes = [("wqkv", 6144, 4096), ("wo", 4096, 4096), ("w13", 2 * F, 4096), ("w2", 4096, F)]
if len(sys.argv) > 1:
    shapes = [x for x in shapes if x[0] in sys.argv[1].split(",")]
for name, N, Kd in shapes:
    X = torch.randn(T, Kd, device=dev).to(bf)
    W = torch.randn(N, Kd, device=dev).to(bf)
    DY = torch.randn(T, N, device=dev).to(bf)
    Y = torch.empty(T, N, device=dev, dtype=bf)
    DX = torch.empty(T, Kd, device=dev, dtype=bf)
    DW = torch.zeros(N, Kd, device=dev, dtype=bf)
    fl = 2.0 * T * N * Kd
    kinds = {
        "fwd": (lambda: K.gemm(X, W, False, False, Y, False, -1), Y),
        "dgrad": (lambda: K.gemm(DY, W, False, True, DX, False, -1), DX),
        "wgrad": (lambda: K.gemm(DY, X, True, True, DW, False, -1), DW),
    }
    for kind, (fn, outbuf) in kinds.items():
        ref, diff = None, {}
        for m in MODES:
            K._L().ie_tune_gemm_tail_split(m)
            outbuf.zero_()
            fn()
            torch.cuda.synchronize()
            if ref is None:
                ref = outbuf.clone()
            diff[m] = float((outbuf.float() - ref.float()).abs().max())
        times = {m: [] for m in MODES}
        for _ in range(5):
            for m in MODES:
                K._L().ie_tune_gemm_tail_split(m)
                times[m].append(t_once(fn, 5))
        rec = {"gemm": name, "kind": kind}
        for m in MODES:
            rec[f"mode{m}_TF"] = round(fl / statistics.median(times[m]) / 1e12, 1)
            rec[f"mode{m}_maxdiff"] = diff[m]
        print(json.dumps(rec), flush=True)
    del X, W, DY, Y, DX, DW
K._L().ie_tune_gemm_tail_split(0)
