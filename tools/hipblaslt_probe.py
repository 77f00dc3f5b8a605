"""Yardstick: hipBLASLt (through torch.matmul / addmm_ / mm) on the 7B layer shapes next to this repo's GEMM (`ie_gemm_bf16`, automatic
schedule) -- same box, same process, bf16, randn data, TFLOP/s from HIP events over `it` launches.  The library is not used by the product.
    python tools/hipblaslt_probe.py [--tokens 16384]"""
import argparse
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

import torch  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--tokens", type=int, default=16384)
args = ap.parse_args()
from internevo_amd import kernels as K  # noqa: E402

dev = torch.device("cuda:0")
bf = torch.bfloat16
T, F, V = args.tokens, 14336, 92544


def t(fn, it=10):
    fn()
    fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(it):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) * 1e-3 / it


for name, N, Kd in [("wqkv", 6144, 4096), ("wo", 4096, 4096), ("w13", 2 * F, 4096), ("w2", 4096, F), ("head", V, 4096)]:
    X = torch.randn(T, Kd, device=dev).to(bf)
    W = torch.randn(N, Kd, device=dev).to(bf)
    DY = torch.randn(T, N, device=dev).to(bf)
    DW = torch.zeros(N, Kd, device=dev, dtype=bf)
    Y = torch.empty(T, N, device=dev, dtype=bf)
    DX = torch.empty(T, Kd, device=dev, dtype=bf)
    fl = 2.0 * T * N * Kd
    r = {"gemm": name, "tokens": T, "N": N, "K": Kd}
    r["fwd_lib_TF"] = fl / t(lambda: torch.matmul(X, W.t(), out=Y)) / 1e12
    r["fwd_ours_TF"] = fl / t(lambda: K.linear_fwd(X, W, Y)) / 1e12
    r["dgrad_lib_TF"] = fl / t(lambda: torch.matmul(DY, W, out=DX)) / 1e12
    r["dgrad_ours_TF"] = fl / t(lambda: K.linear_dgrad(DY, W, DX)) / 1e12
    r["wgrad_lib_TF"] = fl / t(lambda: torch.mm(DY.t(), X, out=DW)) / 1e12
    r["wgrad_ours_TF"] = fl / t(lambda: K.linear_wgrad(DY, X, DW, False)) / 1e12
    print(json.dumps({k: (round(v, 1) if isinstance(v, float) else v) for k, v in r.items()}), flush=True)
    del X, W, DY, DW, Y, DX
