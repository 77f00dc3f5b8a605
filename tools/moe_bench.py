"""Throughput of the INTERNLM_MoE family (BASELINE configs[4], configs/7B_MoE4_sft.py) on ONE MI355X: the whole 32-layer, 4-expert model
(11.9 B parameters: 190 GB of weights, gradients and fp32 optimizer state) fits a single 288 GB GPU, so the expert-parallel all-to-all
is not part of this number.  Synthetic RandomDataset batches (micro_bsz 2 x seq 2048 = 4096 packed tokens, micro_num 4), random-init
weights, Gumbel noise generated on the device.  Prints one JSON line; not the headline bench (that is bench.py on configs[1]).
    python tools/moe_bench.py [--steps 5 --warmup 2 --layers 32]"""
import argparse
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

import torch  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--layers", type=int, default=32)
    ap.add_argument("--expert-fp8", action="store_true", help="OPT-IN: the experts' forward products on e4m3 operands (MoEEngine(expert_fp8=True))")
    args = ap.parse_args()
    from internevo_amd.config import ModelConfig, PathConfig, TrainConfig
    from internevo_amd.data import SyntheticLoader
    from internevo_amd.moe_engine import MoEEngine, ffn_dim

    # configs/7B_MoE4_sft.py
    mc = ModelConfig(vocab_size=103168, hidden_size=4096, num_layers=args.layers, num_attention_heads=32, num_kv_attention_heads=32, mlp_ratio=4 / 3,
                     model_type="INTERNLM_MoE", num_experts=4, moe_capacity_factor=1.0, moe_min_capacity=4, moe_loss_coeff=0.1)
    tc = TrainConfig(seq_len=2048, micro_bsz=2, micro_num=4, total_steps=args.steps + args.warmup, lr=1e-4, fixed_random_dataset_seqlen=True)
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    t0 = time.time()
    eng = MoEEngine(PathConfig(mc, tc), dev, seed=1024, expert_fp8=args.expert_fp8)
    build_s = time.time() - t0
    loader = iter(SyntheticLoader(tc.seq_len, tc.micro_bsz, tc.micro_num, True))
    tokens = tc.seq_len * tc.micro_bsz * tc.micro_num
    for _ in range(args.warmup):
        b, y = next(loader)
        eng.forward_backward(b, y)
        eng.step()
    torch.cuda.synchronize()
    t0 = time.time()
    for _ in range(args.steps):
        b, y = next(loader)
        loss, moe_loss = eng.forward_backward(b, y)
        eng.step()
    torch.cuda.synchronize()
    dt = (time.time() - t0) / args.steps
    st = eng.read_state()
    h, F, L, E, V = mc.hidden_size, ffn_dim(mc), mc.num_layers, mc.num_experts, mc.vocab_size
    # matmul flops per token, forward + backward (x3), top-2 routing: every token passes 2 experts (capacity drops ignored); causal attention exact
    per_token = 3 * 2 * (L * (4 * h * h + 2 * 3 * h * F) + V * h) + 3 * 2 * L * 2 * (tc.seq_len / 2) * h
    print(json.dumps({"bench": "moe_7B_MoE4_sft", "layers": L, "experts": E, "hidden": h, "ffn": F, "params_B": round(eng.params.numel() / 1e9, 2),
                      "tokens_per_step": tokens, "ms_per_step": dt * 1e3, "tokens_per_second": tokens / dt,
                      "tflops_matmul_top2": per_token * tokens / dt * 1e-12, "loss": float(loss), "moe_loss": float(moe_loss),
                      "loss_scale": st.loss_scale, "skipped": st.skipped_total, "hbm_GB": round(torch.cuda.max_memory_allocated() / 2**30, 1),
                      "engine_build_s": round(build_s, 1)}))


if __name__ == "__main__":
    main()
