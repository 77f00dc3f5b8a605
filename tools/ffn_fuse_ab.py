"""A/B of the FFN products with the SwiGLU arithmetic in their epilogues (ie_gemm_swiglu_fwd / _bwd) against the two-launch path
(product, then swiglu_fwd_k / swiglu_bwd_k) at the 7B shapes: interleaved rounds, median; bit-identity of the results checked."""
import json
import os
import statistics
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from internevo_amd import kernels as K  # noqa: E402

dev = torch.device("cuda:0")
bf = torch.bfloat16
T = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
F, H = 14336, 4096


def t_once(fn, iters):
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) * 1e-3 / iters


g = torch.Generator(device=dev).manual_seed(5)
x = (torch.randn(T, H, device=dev, generator=g)).to(bf)
w13 = (torch.randn(2 * F, H, device=dev, generator=g) * 0.02).to(bf)
w2 = (torch.randn(H, F, device=dev, generator=g) * 0.02).to(bf)
dy = (torch.randn(T, H, device=dev, generator=g)).to(bf)
h13, act = torch.empty(T, 2 * F, device=dev, dtype=bf), torch.empty(T, F, device=dev, dtype=bf)
dh13, dact = torch.empty(T, 2 * F, device=dev, dtype=bf), torch.empty(T, F, device=dev, dtype=bf)
L = K._L()
cases = {
    "fwd": (lambda: K.linear_swiglu_fwd(x, w13, h13, act), (h13, act), 2.0 * T * 2 * F * H, 0),
    "bwd": (lambda: K.linear_dgrad_swiglu_bwd(dy, w2, h13, dh13, dact), (dh13,), 2.0 * T * F * H, 1),
}
for kind, (fn, outs, flops, is_bwd) in cases.items():
    L.ie_tune_ffn_fuse(3)
    fused = int(L.ie_gemm_swiglu_is_fused(is_bwd, T, F, H))
    res = {}
    for mode in (0, 3):
        L.ie_tune_ffn_fuse(mode)
        for o in outs:
            o.zero_()
        fn()
        torch.cuda.synchronize()
        res[mode] = [o.clone() for o in outs]
    same = all(torch.equal(a, b) for a, b in zip(res[0], res[3]))
    times = {0: [], 3: []}
    for _ in range(7):
        for mode in (0, 3):
            L.ie_tune_ffn_fuse(mode)
            times[mode].append(t_once(fn, 5))
    L.ie_tune_ffn_fuse(5)
    t0, t1 = statistics.median(times[0]), statistics.median(times[3])
    print(json.dumps({"product": kind, "rows": T, "F": F, "h": H, "one_launch_for_this_shape": fused, "bit_identical": same, "two_launches_us": round(t0 * 1e6, 1),
                      "fused_us": round(t1 * 1e6, 1), "saved_us": round((t0 - t1) * 1e6, 1), "fused_tflops_of_the_product": round(flops / t1 / 1e12, 1)}), flush=True)
