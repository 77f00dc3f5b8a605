"""Probe: what does running kernels of the training step on two HIP streams at once buy?
  (a) an HBM-bound elementwise kernel (swiglu_bwd, 16 384 x 14 336) under an MFMA-bound GEMM (w13 weight gradient) on another stream
  (b) two GEMMs (w13 input gradient + w13 weight gradient) side by side
against the same kernels run one after the other.  Prints one JSON line per case (µs per iteration)."""
import json
import sys

import torch

sys.path.insert(0, ".")
from internevo_amd import kernels as K  # noqa: E402

dev = torch.device("cuda:0")
T, h, F = 16384, 4096, 14336
bf = torch.bfloat16
x = torch.randn(T, h, device=dev, dtype=bf)
w13 = torch.randn(2 * F, h, device=dev, dtype=bf) * 0.02
dw13 = torch.randn(T, 2 * F, device=dev, dtype=bf)
gw13 = torch.zeros(2 * F, h, device=dev, dtype=bf)
dx = torch.empty(T, h, device=dev, dtype=bf)
dact = torch.randn(T, F, device=dev, dtype=bf)
gu = torch.randn(T, 2 * F, device=dev, dtype=bf)
dgu = torch.empty(T, 2 * F, device=dev, dtype=bf)
act = torch.empty(T, F, device=dev, dtype=bf)
s2 = torch.cuda.Stream()


def wgrad():
    K.linear_wgrad(dw13, x, gw13, False)


def dgrad():
    K.linear_dgrad(dw13, w13, dx)


def swiglu():
    K.swiglu_bwd(dact, gu[:, :F], gu[:, F:], dgu[:, :F], dgu[:, F:], act)


def timed(fn, iters=10):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / iters


def both(a, b):
    def run():
        main = torch.cuda.current_stream()
        s2.wait_stream(main)
        with torch.cuda.stream(s2):
            b()
        a()
        main.wait_stream(s2)
    return run


res = {"wgrad": timed(wgrad), "dgrad": timed(dgrad), "swiglu_bwd": timed(swiglu)}
res["serial_swiglu_wgrad"] = timed(lambda: (swiglu(), wgrad()))
res["overlap_swiglu_wgrad"] = timed(both(swiglu, wgrad))
res["serial_dgrad_wgrad"] = timed(lambda: (dgrad(), wgrad()))
res["overlap_dgrad_wgrad"] = timed(both(dgrad, wgrad))
print(json.dumps({"probe": "two_stream_overlap", "us": {k: round(v, 1) for k, v in res.items()}}))
