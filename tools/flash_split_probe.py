"""A/B of the dK/dV head split (ie_tune_flash_dkdv_split) at the 7B attention shape; HIP-event time of the whole backward.
usage: python tools/flash_split_probe.py [nseq]   (nseq sequences of 4096 tokens in one call, default 1; 4 = the merged pass)"""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from internevo_amd import kernels as K  # noqa: E402

dev = torch.device("cuda:0")
S, HQ, HKV, D = 4096, 32, 8, 128
NSEQ = int(sys.argv[1]) if len(sys.argv) > 1 else 1
T = S * NSEQ
bf = torch.bfloat16
q = torch.randn(T, HQ, D, device=dev).to(bf)
kv = torch.randn(T, 2, HKV, D, device=dev).to(bf)
do = torch.randn(T, HQ, D, device=dev).to(bf)
cu = (torch.arange(NSEQ + 1, dtype=torch.int32) * S).to(dev)
o, lse = K.flash_attn_fwd(q, kv[:, 0], kv[:, 1], cu, S, None, True)
ws = torch.empty(K._L().ie_flash_attn_bwd_workspace(T, HQ, HKV, D), dtype=torch.float32, device=dev)
ref = None
for rnd in range(2):
    for split in (1, 2, 4):
        K._L().ie_tune_flash_dkdv_split(split)
        for _ in range(2):
            dq, dk, dv = K.flash_attn_bwd(do, q, kv[:, 0], kv[:, 1], o, lse, cu, S, None, True, delta_ws=ws)
        torch.cuda.synchronize()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(10):
            dq, dk, dv = K.flash_attn_bwd(do, q, kv[:, 0], kv[:, 1], o, lse, cu, S, None, True, delta_ws=ws)
        e.record()
        torch.cuda.synchronize()
        if ref is None:
            ref = (dk.float().clone(), dv.float().clone())
        err = max(float((dk.float() - ref[0]).abs().max()), float((dv.float() - ref[1]).abs().max()))
        print(json.dumps({"nseq": NSEQ, "dkdv_split": split, "bwd_us": s.elapsed_time(e) * 100, "max_abs_diff_vs_split1": err}), flush=True)
K._L().ie_tune_flash_dkdv_split(0)
