"""Per-kernel micro-benchmark at the InternLM2-7B / seq-4096 shapes (T = 4096 tokens per micro-batch).

Prints one line per kernel: average launch time (HIP events on the launch stream), achieved GB/s
against the ALGORITHMIC bytes (or TFLOP/s against algorithmic flops), and the fraction of the
MI355X peak (HBM 8 TB/s, bf16 MFMA 2.5 PFLOP/s dense).  torch.matmul (hipBLASLt) is timed beside the
hand-written GEMM purely as a comparison point; it is never on the product path.

usage: python tools/kernel_bench.py [--quick] [--json out.json]
"""
import argparse
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from internevo_amd import kernels as K  # noqa: E402
from internevo_amd._lib import IeScalerConfig  # noqa: E402

HBM_PEAK = 8.0e12
MFMA_PEAK = 2.5e15


def timeit(fn, iters=20, warmup=3):
    for _ in range(warmup):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) * 1e-3 / iters


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--quick", action="store_true")
    ap.add_argument("--json", default=None)
    args = ap.parse_args()
    dev = torch.device("cuda:0")
    T, H, F, V, HQ, HKV, D = 4096, 4096, 14336, 92544, 32, 8, 128
    bf = torch.bfloat16
    res = []

    def rec(name, sec, nbytes=None, flops=None):
        r = {"kernel": name, "us": sec * 1e6}
        if nbytes is not None:
            r["GBps"] = nbytes / sec / 1e9
            r["frac_hbm"] = nbytes / sec / HBM_PEAK
        if flops is not None:
            r["TFLOPs"] = flops / sec / 1e12
            r["frac_mfma"] = flops / sec / MFMA_PEAK
        res.append(r)
        print(json.dumps(r), flush=True)

    x = torch.randn(T, H, device=dev).to(bf)
    x2 = torch.randn(T, H, device=dev).to(bf)
    w = torch.ones(H, device=dev, dtype=bf)
    _, rstd = K.rmsnorm_fwd(x, w, 1e-5)
    rec("rmsnorm_fwd[T,4096]", timeit(lambda: K.rmsnorm_fwd(x, w, 1e-5)), nbytes=T * H * 4)
    r_out = torch.empty_like(x)
    rec("add_rmsnorm_fwd[T,4096]", timeit(lambda: K.add_rmsnorm_fwd(x, x2, w, 1e-5, r_out)), nbytes=T * H * 8)
    ws = torch.empty(K._L().ie_rmsnorm_bwd_partials(T) * H, dtype=torch.float32, device=dev)
    dw = torch.zeros(H, device=dev, dtype=bf)
    rec("rmsnorm_bwd+dres[T,4096]", timeit(lambda: K.rmsnorm_bwd(x2, x, w, rstd, x2, dw, True, ws)), nbytes=T * H * 8)

    qkv = torch.randn(T, (HQ + 2 * HKV) * D, device=dev).to(bf)
    cos = torch.randn(T, D // 2, device=dev).to(bf)
    sin = torch.randn(T, D // 2, device=dev).to(bf)
    pos = torch.arange(T, device=dev)
    q_o = torch.empty(T, HQ, D, device=dev, dtype=bf)
    kv_o = torch.empty(T, 2, HKV, D, device=dev, dtype=bf)
    rec("qkv_rotary_fwd", timeit(lambda: K.qkv_rotary_fwd(qkv, cos, sin, pos, HKV, HQ // HKV, D, True, q_o, kv_o)),
        nbytes=T * (HQ + 2 * HKV) * D * 4)
    dqkv = torch.empty_like(qkv)
    rec("qkv_rotary_bwd", timeit(lambda: K.qkv_rotary_bwd(q_o, kv_o, cos, sin, pos, HKV, HQ // HKV, D, True, dqkv)),
        nbytes=T * (HQ + 2 * HKV) * D * 4)

    w13 = torch.randn(T, 2 * F, device=dev).to(bf)
    act = torch.empty(T, F, device=dev, dtype=bf)
    rec("swiglu_fwd[T,14336]", timeit(lambda: K.swiglu_fwd(w13[:, :F], w13[:, F:], act)), nbytes=T * F * 6)
    dw13 = torch.empty_like(w13)
    rec("swiglu_bwd(+act)[T,14336]", timeit(lambda: K.swiglu_bwd(act, w13[:, :F], w13[:, F:], dw13[:, :F], dw13[:, F:], act)),
        nbytes=T * F * 12)

    logits = torch.randn(T, V, device=dev).to(bf)
    labels = torch.randint(0, V, (T,), device=dev)
    _, lse, loss, count = K.ce_fwd(logits, labels)
    rec("ce_fwd[T,92544] bf16", timeit(lambda: K.ce_fwd(logits, labels), iters=10), nbytes=T * V * 2)
    one = torch.ones(1, device=dev)
    rec("ce_bwd[T,92544] bf16 in-place", timeit(lambda: K.ce_bwd(logits, labels, lse, one, count), iters=10), nbytes=T * V * 4)

    n = 218_112_000  # one InternLM2-7B layer's parameters
    gbuf = torch.randn(n, device=dev).to(bf)
    pws = torch.empty(2048, dtype=torch.float32, device=dev)
    ss = torch.zeros(1, device=dev)
    rec("sumsq[218M bf16]", timeit(lambda: K.sumsq(gbuf, ss, False, pws), iters=10), nbytes=n * 2)
    p32 = torch.randn(n, device=dev)
    m = torch.zeros(n, device=dev)
    v = torch.zeros(n, device=dev)
    p16 = torch.empty(n, device=dev, dtype=bf)
    st = K.step_state_new(dev, 65536.0)
    cfg = IeScalerConfig(2.0, 0.5, 1.0, float(2**24), 1000, 2, 1.0, 1)
    K.step_control(st, K.sumsq(gbuf), cfg)
    rec("adamw[218M] (28 B/param)", timeit(lambda: K.adamw_step(gbuf, p32, m, v, p16, st, 1e-4, 0.9, 0.95, 1e-8, 0.01), iters=10), nbytes=n * 28)
    del p32, m, v, p16, gbuf

    emb = torch.randn(V, H, device=dev).to(bf)
    ids = torch.randint(0, 30, (T,), device=dev)
    eo = torch.empty(T, H, device=dev, dtype=bf)
    rec("embedding_fwd", timeit(lambda: K.embedding_fwd(emb, ids, eo)), nbytes=T * H * 4)
    demb = torch.zeros(V, H, device=dev, dtype=bf)
    pres = torch.empty(V + 1 + T, dtype=torch.int32, device=dev)
    rec("embedding_bwd (30 distinct ids)", timeit(lambda: K.embedding_bwd(eo, ids, demb, True, pres), iters=5), nbytes=T * H * 2)
    del emb, demb

    # ---- GEMMs of one layer (+ head): fwd (NT), dgrad (NN), wgrad (TN)
    shapes = [("wqkv", 6144, 4096), ("wo", 4096, 4096), ("w13", 2 * F, 4096), ("w2", 4096, F)]
    if not args.quick:
        shapes.append(("head", V, 4096))
    for name, N, Kd in shapes:
        X = torch.randn(T, Kd, device=dev).to(bf)
        W = torch.randn(N, Kd, device=dev).to(bf)
        DY = torch.randn(T, N, device=dev).to(bf)
        Y = torch.empty(T, N, device=dev, dtype=bf)
        DX = torch.empty(T, Kd, device=dev, dtype=bf)
        DW = torch.zeros(N, Kd, device=dev, dtype=bf)
        fl = 2.0 * T * N * Kd
        rec(f"gemm fwd {name} [{T}x{N}x{Kd}]", timeit(lambda: K.linear_fwd(X, W, Y), iters=10), flops=fl)
        rec(f"gemm dgrad {name}", timeit(lambda: K.linear_dgrad(DY, W, DX), iters=10), flops=fl)
        rec(f"gemm wgrad {name} (+accumulate)", timeit(lambda: K.linear_wgrad(DY, X, DW, True), iters=10), flops=fl)
        rec(f"  [comparison] torch.matmul fwd {name}", timeit(lambda: torch.matmul(X, W.t()), iters=10), flops=fl)
        del X, W, DY, Y, DX, DW

    # ---- attention, one 4096-token sequence, GQA 32/8, causal
    q = torch.randn(T, HQ, D, device=dev).to(bf)
    kv = torch.randn(T, 2, HKV, D, device=dev).to(bf)
    cu = torch.tensor([0, T], dtype=torch.int32, device=dev)
    out = torch.empty_like(q)
    fl_fwd = 4.0 * T * T * D * HQ / 2
    o, lse_a = K.flash_attn_fwd(q, kv[:, 0], kv[:, 1], cu, T, None, True, out)
    rec("flash_attn_fwd causal [4096, 32/8, 128]", timeit(lambda: K.flash_attn_fwd(q, kv[:, 0], kv[:, 1], cu, T, None, True, out), iters=10),
        flops=fl_fwd)
    do = torch.randn_like(q)
    dq = torch.empty_like(q)
    dkv = torch.empty_like(kv)
    dws = torch.empty(HQ * T, dtype=torch.float32, device=dev)
    rec("flash_attn_bwd causal (algorithmic 2.5x fwd)",
        timeit(lambda: K.flash_attn_bwd(do, q, kv[:, 0], kv[:, 1], o, lse_a, cu, T, None, True, dq, dkv[:, 0], dkv[:, 1], dws), iters=5),
        flops=2.5 * fl_fwd)
    for occ in (1, 2):  # A/B of the dQ kernel's occupancy target (2 = default)
        K._L().ie_tune_flash_dq_occupancy(occ)
        rec(f"flash_attn_bwd causal, dQ kernel compiled for {occ} wave(s)/SIMD",
            timeit(lambda: K.flash_attn_bwd(do, q, kv[:, 0], kv[:, 1], o, lse_a, cu, T, None, True, dq, dkv[:, 0], dkv[:, 1], dws), iters=5),
            flops=2.5 * fl_fwd)
    # packed: 8 sequences of 512
    cu8 = torch.arange(0, T + 1, 512, dtype=torch.int32, device=dev)
    rec("flash_attn_fwd causal packed 8x512", timeit(lambda: K.flash_attn_fwd(q, kv[:, 0], kv[:, 1], cu8, 512, None, True, out), iters=10),
        flops=4.0 * 8 * 512 * 512 * D * HQ / 2)

    if args.json:
        os.makedirs(os.path.dirname(os.path.abspath(args.json)), exist_ok=True)
        with open(args.json, "w") as f:
            json.dump(res, f, indent=1)


if __name__ == "__main__":
    main()
