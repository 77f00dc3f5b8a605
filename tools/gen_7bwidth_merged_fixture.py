"""The CPU oracle's side of tests/test_engine_gpu.py::test_engine_7b_width_merged_benchmark_step_matches_oracle, computed OFFLINE and committed:
two steps of the benchmark recipe (lr 1e-4 from step 0, AdamW, loss scale 2^16, clip 1.0; four micro-batches of one 4096-token sequence each) at the 7B
model's width with one layer, OracleTrainer with the accelerator arithmetic of the embedding gradient (oracle.ops.embedding_grad_in_fp32, which rests on
tests/test_kernels_gpu.py::test_embedding_gradient_of_the_benchmark_batch_against_fp64).  The oracle needs ~75 s per 16 384-token step on the GPU box's
host cores; the GPU suite has a time limit, so the test compares the HIP engine with THIS record instead of re-running the oracle:

    per step: loss, global gradient norm, loss scale; per parameter: the gradient's l2 norm (fp64) and a fixed strided SAMPLE of the (loss-scaled,
    accumulated) gradient -- every `stride`-th element, ~1e5 per tensor -- on which the test computes the relative l2 difference and the sign agreement;
    after step 1: the same sample of the trained bf16 weights.

    python tools/gen_7bwidth_merged_fixture.py            # -> tests/golden/merged_7bwidth_oracle.npz (+ .json), ~3 min on 8 cores, ~40 GB of host memory
    python tools/gen_7bwidth_merged_fixture.py single     # -> tests/golden/single_7bwidth_oracle.*: the oracle's side of
                                                          #    test_engine_7b_shaped_layer_full_size_matches_oracle (ONE step, one 4096-token micro-batch of several
                                                          #    packed sequences, the CPU kernel's embedding-gradient arithmetic as that test always used)
"""
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
SAMPLE = 100_000   # elements kept per tensor (tensors below that size are kept whole)


def stride_of(numel):
    return max(1, numel // SAMPLE)


def main():
    from internevo_amd.config import internlm2_7b
    from internevo_amd.data import SyntheticLoader
    from oracle import ops as O
    from oracle.step import OracleTrainer

    import contextlib

    single = len(sys.argv) > 1 and sys.argv[1] == "single"
    cfg = internlm2_7b(4096)
    cfg.model.num_layers = 1
    cfg.train.micro_num = 1 if single else 4
    if single:
        cfg.train.total_steps = 4
    else:
        cfg.train.fixed_random_dataset_seqlen = True
    tr = OracleTrainer(cfg, torch.bfloat16)
    loader = iter(SyntheticLoader(4096, 1, cfg.train.micro_num, not single, 4000))
    arrays, meta = {}, {"what": __doc__.split("\n\n")[0], "steps": [], "sample": SAMPLE, "threads": torch.get_num_threads(), "lr": cfg.train.lr,
                        "stride": {n: stride_of(p.numel()) for n, p in tr.params.items()}}
    for k in range(1 if single else 2):
        batch, labels = next(loader)
        t0 = time.time()
        with (contextlib.nullcontext() if single else O.embedding_grad_in_fp32()):
            r = tr.train_step(batch, labels)
        rec = {"loss": r["loss"], "grad_norm": r["grad_norm"], "loss_scale": r["loss_scale"], "ok": bool(r["ok"]), "grad_l2": {}}
        for n, p in tr.params.items():
            g = p.grad.reshape(-1)
            rec["grad_l2"][n] = float(g.double().norm())
            arrays[f"g{k}/{n}"] = g[:: stride_of(g.numel())].float().numpy()
        meta["steps"].append(rec)
        print(f"step {k}: loss {r['loss']:.5f} grad_norm {r['grad_norm']:.4f} ({time.time() - t0:.1f} s)", flush=True)
    for n, p in tr.params.items():
        arrays[f"p/{n}"] = p.detach().reshape(-1)[:: stride_of(p.numel())].float().numpy()
    meta["total_steps"], meta["micro_num"] = cfg.train.total_steps, cfg.train.micro_num
    out = os.path.join(ROOT, "tests", "golden", "single_7bwidth_oracle" if single else "merged_7bwidth_oracle")
    np.savez_compressed(out + ".npz", **arrays)
    with open(out + ".json", "w") as f:
        json.dump(meta, f, indent=1)
    print("wrote", out + ".npz", os.path.getsize(out + ".npz") // 1024, "KB")


if __name__ == "__main__":
    main()
