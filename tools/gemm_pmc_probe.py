"""Launch a few GEMMs of one layer shape per tile variant (+ torch.matmul = hipBLASLt for comparison) -- meant to run
under `rocprofv3 --pmc ... --kernel-trace` so the per-kernel counters can be compared (tools/rocprof_summary.py)."""
import sys

import torch

sys.path.insert(0, __file__.rsplit("/", 2)[0])
from internevo_amd import kernels as K  # noqa: E402

dev = torch.device("cuda:0")
bf = torch.bfloat16
T, N, Kd = 4096, 4096, 14336  # w2 fwd
X = torch.randn(T, Kd, device=dev).to(bf)
W = torch.randn(N, Kd, device=dev).to(bf)
Y = torch.empty(T, N, device=dev, dtype=bf)
variants = [int(v) for v in (sys.argv[1] if len(sys.argv) > 1 else "9,11,15").split(",")]
for _ in range(3):
    for v in variants:
        K.gemm(X, W, False, False, Y, False, v)
    torch.matmul(X, W.t())
torch.cuda.synchronize()
