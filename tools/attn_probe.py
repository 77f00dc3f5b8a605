"""Tiny driver for rocprofv3 --pmc passes: a few launches of the attention kernels (and optionally one GEMM of each
kind) at the 7B shapes.  usage: python tools/attn_probe.py [--gemm]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from internevo_amd import kernels as K  # noqa: E402

dev = torch.device("cuda:0")
T, HQ, HKV, D = 4096, 32, 8, 128
bf = torch.bfloat16
q = torch.randn(T, HQ, D, device=dev).to(bf)
kv = torch.randn(T, 2, HKV, D, device=dev).to(bf)
do = torch.randn(T, HQ, D, device=dev).to(bf)
cu = torch.tensor([0, T], dtype=torch.int32, device=dev)
for _ in range(10):
    o, lse = K.flash_attn_fwd(q, kv[:, 0], kv[:, 1], cu, T, None, True)
    K.flash_attn_bwd(do, q, kv[:, 0], kv[:, 1], o, lse, cu, T, None, True)
if "--gemm" in sys.argv:
    N, Kd = 4096, 14336
    X = torch.randn(T, Kd, device=dev).to(bf)
    W = torch.randn(N, Kd, device=dev).to(bf)
    DY = torch.randn(T, N, device=dev).to(bf)
    for _ in range(3):
        K.linear_fwd(X, W)
        K.linear_dgrad(DY, W)
        K.linear_wgrad(DY, X)
torch.cuda.synchronize()
print("done")
