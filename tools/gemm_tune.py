"""A/B the GEMM tile variants on the 7B layer shapes (interleaved rounds in one process, median of rounds;
guide section 5.4 rules 13/24), plus a correctness check of each variant against torch fp32 on the device.

usage: python tools/gemm_tune.py [--json out.json]
"""
import argparse
import json
import os
import statistics
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from internevo_amd import kernels as K  # noqa: E402

ALL_VARIANTS = {4: "dma256", 5: "dma128", 6: "dma256_spread2", 7: "dma256_spread4", 8: "dma128_spread2", 9: "dma256_phased", 10: "dma128_phased",
                11: "buf256_w4", 12: "dma128x256_phased", 13: "buf256_phased2", 14: "buf128x256_phased2"}
VARIANTS = {v: ALL_VARIANTS[v] for v in (5, 8, 9, 10, 11)}


def t_once(fn, iters):
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) * 1e-3 / iters


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--json", default=None)
    ap.add_argument("--rounds", type=int, default=5)
    ap.add_argument("--variants", default=None, help="comma list of tile variant ids (default: 5,6,8,9,10)")
    ap.add_argument("--shapes", default=None, help="comma list out of wqkv,wo,w13,w2,head")
    args = ap.parse_args()
    if args.variants:
        VARIANTS.clear()
        VARIANTS.update({int(v): ALL_VARIANTS[int(v)] for v in args.variants.split(",")})
    dev = torch.device("cuda:0")
    T, F, V = 4096, 14336, 92544
    bf = torch.bfloat16
    shapes = [("wqkv", 6144, 4096), ("wo", 4096, 4096), ("w13", 2 * F, 4096), ("w2", 4096, F), ("head", V, 4096)]
    if args.shapes:
        shapes = [x for x in shapes if x[0] in args.shapes.split(",")]
    out = []
    # correctness of every variant / kind on an awkward shape
    for v in VARIANTS:
        M, N, Kd = (520, 392, 200) if v < 4 else (520, 392, 192)  # ragged M/N edges; the DMA variants need K % 64 == 0
        for akm, bkm in ((False, False), (False, True), (True, True), (True, False)):
            A = torch.randn((Kd, M) if akm else (M, Kd), device=dev).to(bf)
            B = torch.randn((Kd, N) if bkm else (N, Kd), device=dev).to(bf)
            ref = (A.float().t() if akm else A.float()) @ (B.float() if bkm else B.float().t())
            C = K.gemm(A, B, akm, bkm, variant=v)
            err = float((C.float() - ref).abs().max())
            ok = err <= 8e-3 * float(ref.abs().max()) + 0.05
            print(json.dumps({"check": VARIANTS[v], "akm": akm, "bkm": bkm, "max_abs_err": err, "ok": ok}), flush=True)
            assert ok
    for name, N, Kd in shapes:
        X = torch.randn(T, Kd, device=dev).to(bf)
        W = torch.randn(N, Kd, device=dev).to(bf)
        DY = torch.randn(T, N, device=dev).to(bf)
        Y = torch.empty(T, N, device=dev, dtype=bf)
        DX = torch.empty(T, Kd, device=dev, dtype=bf)
        DW = torch.zeros(N, Kd, device=dev, dtype=bf)
        fl = 2.0 * T * N * Kd
        kinds = {
            "fwd": lambda v: K.gemm(X, W, False, False, Y, False, v),
            "dgrad": lambda v: K.gemm(DY, W, False, True, DX, False, v),
            "wgrad": lambda v: K.gemm(DY, X, True, True, DW, True, v),
        }
        for kind, fn in kinds.items():
            times = {v: [] for v in VARIANTS}
            for v in VARIANTS:
                fn(v)
            torch.cuda.synchronize()
            for _ in range(args.rounds):
                for v in VARIANTS:
                    times[v].append(t_once(lambda: fn(v), 5))
            auto = t_once(lambda: fn(-1), 10)
            rec = {"gemm": name, "kind": kind, "auto_TF": fl / auto / 1e12}
            for v in VARIANTS:
                rec[VARIANTS[v] + "_TF"] = fl / statistics.median(times[v]) / 1e12
            print(json.dumps(rec), flush=True)
            out.append(rec)
        ref_t = t_once(lambda: torch.matmul(X, W.t()), 10)
        print(json.dumps({"gemm": name, "kind": "[comparison] torch.matmul fwd", "TF": fl / ref_t / 1e12}), flush=True)
        del X, W, DY, Y, DX, DW
    if args.json:
        with open(args.json, "w") as f:
            json.dump(out, f, indent=1)


if __name__ == "__main__":
    main()
