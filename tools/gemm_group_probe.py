"""A/B of the tile-order group size (ie_tune_gemm_group) on the 7B layer shapes."""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from internevo_amd import kernels as K  # noqa: E402

dev = torch.device("cuda:0"); bf = torch.bfloat16
T, F = 4096, 14336


def t(fn, it=8):
    fn(); torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(it):
        fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) * 1e-3 / it


for name, N, Kd in [("w13", 2 * F, 4096), ("w2", 4096, F), ("head", 92544, 4096)]:
    X = torch.randn(T, Kd, device=dev).to(bf); W = torch.randn(N, Kd, device=dev).to(bf); DY = torch.randn(T, N, device=dev).to(bf)
    Y = torch.empty(T, N, device=dev, dtype=bf); DX = torch.empty(T, Kd, device=dev, dtype=bf); DW = torch.zeros(N, Kd, device=dev, dtype=bf)
    fl = 2.0 * T * N * Kd
    for rnd in range(2):
        for gm in (1, 2, 4, 8, 16):
            K._L().ie_tune_gemm_group(gm)
            r = {"gemm": name, "group": gm,
                 "fwd": fl / t(lambda: K.gemm(X, W, False, False, Y)) / 1e12,
                 "dgrad": fl / t(lambda: K.gemm(DY, W, False, True, DX)) / 1e12,
                 "wgrad": fl / t(lambda: K.gemm(DY, X, True, True, DW, True)) / 1e12}
            if rnd == 1:
                print(json.dumps({k: (round(v) if isinstance(v, float) else v) for k, v in r.items()}), flush=True)
    del X, W, DY, Y, DX, DW
K._L().ie_tune_gemm_group(0)
