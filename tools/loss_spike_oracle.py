"""Oracle trajectory of the first optimizer steps of the benchmark's recipe at the 7B model's WIDTH (hidden 4096, 32 / 8 heads, FFN 14336,
vocabulary 92 544), cut to --layers layers so that the CPU oracle finishes in minutes: lr 1e-4 from step 0 (warmup_ratio 0.01 of 20 steps =
no warm-up), AdamW, dynamic loss scale 2^16, clip 1.0, the synthetic RandomDataset batches of bench.py (fixed_random_dataset_seqlen=True).

Why: the 7B bench run's loss goes 11.4 -> 27.7 (step 3, grad norm 185) -> 0.9 -> 0.004.  This script records what the ORACLE (the
restatement of the reference's torch path, pinned on reference runs) does on the same recipe; tests/test_engine_gpu.py then checks that the
HIP engine retraces it step for step -- if both show the same excursion it is the optimizer's (Adam's first sign-like steps at full
learning rate on a degenerate data set), not a kernel's.

    python tools/loss_spike_oracle.py --layers 2 --steps 8 --out tests/golden/spike_7bwidth_oracle.json
"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--layers", type=int, default=2)
    ap.add_argument("--steps", type=int, default=8)
    ap.add_argument("--micro-num", type=int, default=1)
    ap.add_argument("--seq-len", type=int, default=4096)
    ap.add_argument("--out", default=os.path.join(ROOT, "tests", "golden", "spike_7bwidth_oracle.json"))
    ap.add_argument("--embed-grad-fp32", action="store_true",
                    help="the embedding's weight gradient summed per token in fp32 (torch's accelerator kernel; oracle/ops.py) instead of the CPU kernel's row-by-row bf16 sum")
    args = ap.parse_args()
    from internevo_amd.config import internlm2_7b
    from internevo_amd.data import SyntheticLoader
    from oracle import ops as O
    from oracle.step import OracleTrainer

    if args.embed_grad_fp32:
        O.embedding_grad_in_fp32().__enter__()
    cfg = internlm2_7b(args.seq_len)
    cfg.model.num_layers = args.layers
    cfg.train.micro_num = args.micro_num
    cfg.train.fixed_random_dataset_seqlen = True
    tr = OracleTrainer(cfg, torch.bfloat16)
    loader = iter(SyntheticLoader(args.seq_len, 1, args.micro_num, True, 1_000_000))
    steps = []
    for k in range(args.steps):
        batch, labels = next(loader)
        t0 = time.time()
        r = tr.train_step(batch, labels)
        steps.append({"loss": r["loss"], "grad_norm": r["grad_norm"], "loss_scale": r["loss_scale"], "ok": bool(r["ok"])})
        print(f"step {k}: loss {r['loss']:.5f} grad_norm {r['grad_norm']:.4f} scale {r['loss_scale']} ({time.time() - t0:.1f} s)", flush=True)
    out = {"what": "oracle (CPU, bf16) trajectory of bench.py's recipe at 7B width", "layers": args.layers, "micro_num": args.micro_num,
           "seq_len": args.seq_len, "lr": cfg.train.lr, "total_steps": cfg.train.total_steps, "warmup_ratio": cfg.train.warmup_ratio,
           "init": "oracle.model.formula_init", "embedding_grad": "fp32 sums per token (accelerator kernel)" if args.embed_grad_fp32 else "row-by-row bf16 sum (CPU kernel)", "data": "SyntheticLoader(seq_len, 1, micro_num, fixed_seqlen=True, 1_000_000)", "steps": steps,
           "threads": torch.get_num_threads()}
    with open(args.out, "w") as f:
        json.dump(out, f, indent=1)
    print("wrote", args.out)


if __name__ == "__main__":
    main()
