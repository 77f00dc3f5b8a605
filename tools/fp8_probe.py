"""Throughput of ie_gemm_fp8 against ie_gemm_bf16 on forward shapes at 16 384 token rows (HIP events, 10 launches each, same process).
python tools/fp8_probe.py [--tokens 16384]"""
import argparse
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from internevo_amd import kernels as K  # noqa: E402


def time_us(fn, iters=10):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    e1.synchronize()
    return e0.elapsed_time(e1) * 1e3 / iters


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--tokens", type=int, default=16384)
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    g = torch.Generator(device=dev).manual_seed(0)
    for name, N, Kd in (("wo", 4096, 4096), ("w1|w3", 28672, 4096), ("w2", 4096, 14336), ("expert w1|w3 (7B_MoE4)", 10944, 4096), ("expert w2 (7B_MoE4)", 4096, 5504)):
        Kd = (Kd + 127) // 128 * 128
        x = torch.randn(a.tokens, Kd, device=dev, generator=g).to(torch.bfloat16)
        w = (torch.randn(N, Kd, device=dev, generator=g) * 0.02).to(torch.bfloat16)
        qx, dx = K.fp8_quantize(x)
        qw, dw = K.fp8_quantize(w)
        out = torch.empty(a.tokens, N, dtype=torch.bfloat16, device=dev)
        t8 = time_us(lambda: K.gemm_fp8(qx, dx, qw, dw, out=out))
        t16 = time_us(lambda: K.gemm(x, w, out=out))
        tq = time_us(lambda: K.fp8_quantize(x))
        fl = 2.0 * a.tokens * N * Kd
        print(json.dumps({"product": name, "M": a.tokens, "N": N, "K": Kd, "fp8_us": round(t8, 1), "fp8_tflops": round(fl / t8 * 1e-6, 1), "bf16_us": round(t16, 1),
                          "bf16_tflops": round(fl / t16 * 1e-6, 1), "quantize_x_us": round(tq, 1), "quantize_x_GBps": round(3.0 * a.tokens * Kd / tq * 1e-3, 1)}), flush=True)


if __name__ == "__main__":
    main()
