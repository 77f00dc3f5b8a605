"""Benchmark of the hot path: full training steps of InternLM2-7B (bf16, seq 4096, micro_bsz 1 x micro_num 4,
ZeRO-1 over the data-parallel group) on N MI355X, one process per GPU.

    python bench.py --gpus 1 --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...

A "step" = NonPipelineScheduler.forward_backward_step over micro_num micro-batches + HybridZeroOptimizer.step
(forward, backward, gradient exchange, grad-norm, loss-scale logic, AdamW, parameter exchange) on synthetic
data of the reference's RandomDataset/packed shape, random-init weights.  Rank 0 prints ONE JSON line.

metric/value : whole-job tokens/s (= TGS x n_gpus); `tgs` and the two TFLOPS/GPU figures (reference Megatron
               formula internlm/utils/common.py:208-238, and exact causal-aware GQA count) are in the same line.
roofline     : the dominant kernel (the bf16 MFMA GEMM, ~85 % of step flops): algorithmic flops per launch /
               average launch duration measured with HIP events on the launch stream over the timed region,
               against the 2.5 PFLOP/s dense bf16 MFMA peak.
cpu_baseline : the CPU oracle (oracle/, a restatement of the reference's pure-torch path) timed on this box's
               host cores on a bounded sample, rank 0 / N=1 only.
"""
import argparse
import json
import os
import sys
import time

# the host driver of these boxes supports dmabuf IPC only: without this RCCL's peer mappings fail (hipIpcGetMemHandle: invalid argument).  Exported on the
# GPU boxes already; set here as well, before the HIP runtime loads, so that a launcher with a scrubbed environment still gets a working N > 1 run
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

import torch  # noqa: E402

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

MFMA_PEAK = 2.5e15  # dense bf16, /opt/skills/guides/MI355X_MICROARCH.md


def exact_flops_per_token(mc, seq_len, causal=True):
    """fwd+bwd algorithmic flops per token, GQA-aware (47.37 GFLOP causal / 50.59 dense at 7B, s=4096)."""
    h, f, v = mc.hidden_size, mc.ffn_dim, mc.vocab_size
    lin = 2 * (mc.qkv_dim * h + h * h + 3 * f * h)
    attn = 4 * seq_len * h * (0.5 if causal else 1.0)
    return 3 * ((lin + attn) * mc.num_layers + 2 * v * h)


def host_cores():
    """CPU cores this process may actually use: the cgroup quota when there is one (the GPU boxes of this pool show 256 logical CPUs under a quota of 16
    cores: 128 torch threads on them run the oracle SLOWER than 16), else the affinity mask."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(quota) // int(period)))
    except (OSError, ValueError):
        try:
            q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            if q > 0:
                n = min(n, max(1, q // int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())))
        except (OSError, ValueError):
            pass
    return n


def cpu_baseline(cfg, budget_note=True):
    """Time the CPU oracle on a bounded sample of the same workload: one 4096-token micro-batch through a
    7B-shaped model with 0 and with 1 transformer layer (fwd + bwd + optimizer), extrapolated linearly to
    the full depth.  ~10-40 s of host work."""
    import copy

    from oracle.step import OracleTrainer

    threads = host_cores()
    torch.set_num_threads(threads)   # (one thread per core the box grants; torch's own default follows the logical CPU count)
    tc = cfg.train
    gen = torch.Generator().manual_seed(0)
    ids = torch.randint(0, 30, (1, tc.packed_length), generator=gen)
    labels = torch.cat([ids[:, 1:], torch.full((1, 1), -100)], 1)
    times = {}
    for nl in (0, 1):
        c = copy.deepcopy(cfg)
        c.model.num_layers = nl
        c.train.micro_num = 1
        tr = OracleTrainer(c, torch.bfloat16, init_fn=lambda n, s: torch.empty(s).normal_(0, 0.02, generator=gen) if len(s) > 1 else torch.ones(s))
        batch = {"input_ids": ids, "cu_seqlens": None, "indexes": None}
        t0 = time.time()
        tr.train_step(batch, labels)
        times[nl] = time.time() - t0
        del tr
    t_layer = max(times[1] - times[0], 1e-9)
    full = times[0] + cfg.model.num_layers * t_layer
    return {
        "value": tc.packed_length / full,
        "unit": "tokens/s",
        "cores": threads,
        "kind": "port",
        "sample": f"oracle fwd+bwd+AdamW of one {tc.packed_length}-token micro-batch, 7B-shaped model with 0 layers ({times[0]:.2f} s) and 1 layer "
                  f"({times[1]:.2f} s), linearly extrapolated to {cfg.model.num_layers} layers ({full:.1f} s); torch CPU bf16, {threads} threads",
    }


def _self_launch(n):
    """Re-run this command line as n ranks under torch.distributed.run on 127.0.0.1 with a free port; returns its exit code."""
    import socket
    import subprocess

    have = torch.cuda.device_count()
    if have < (1 if (os.environ.get("IE_BENCH_BACKEND") == "gloo" and "--staged-test" in sys.argv) else n):
        print(f"bench.py --gpus {n}: only {have} GPU(s) visible on this node (one process per GPU over RCCL)", file=sys.stderr)
        return 2
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.run(cmd, env=env).returncode


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=8)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--config", default="7B_internlm2", choices=["7B_internlm2", "7B_llama2", "tiny"])
    ap.add_argument("--seq-len", type=int, default=4096)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--staged-test", action="store_true", help="TEST HOOK, together with IE_BENCH_BACKEND=gloo: every rank on cuda:0, collectives staged through "
                    "the host (exercises the N-rank launch on a one-GPU box; the line it prints says it is not a measurement)")
    ap.add_argument("--no-kernel-timing", action="store_true")
    ap.add_argument("--fwd-variant", type=int, default=-1, help="A/B only: force a GEMM tile variant on the forward products")
    ap.add_argument("--merge-micro", type=int, default=None, help="A/B only: 1 = run the micro-batches of a step as one merged pass, 0 = sequentially")
    ap.add_argument("--no-batch-wgrad", action="store_true", help="A/B only: one weight-gradient GEMM per micro-batch (engine batch_wgrad=False)")
    ap.add_argument("--attn-fwd-variant", type=int, default=None, help="A/B only: ie_tune_flash_fwd_variant (library default when omitted)")
    ap.add_argument("--gemm-persistent", type=int, default=None, help="A/B only: ie_tune_gemm_persistent mode (library default when omitted)")
    ap.add_argument("--gemm-tail-split", type=int, default=None, help="A/B only: ie_tune_gemm_tail_split mode (library default when omitted)")
    ap.add_argument("--checkpoint", type=float, default=0.0, help="model.checkpoint: fraction of layers under activation checkpointing "
                                                                  "(needed for --seq-len 32768 on one GPU); changes the Megatron flops factor to 4")
    ap.add_argument("--micro-num", type=int, default=None, help="override data.micro_num (gradient accumulation steps)")
    ap.add_argument("--num-chunks", type=int, default=1, help="model chunks per pipeline stage (model.num_chunks; > 1 = interleaved 1F1B; needs --pp > 1)")
    ap.add_argument("--pp", type=int, default=1, help="pipeline-parallel size (1F1B, parallel.pipeline=dict(size=pp)); "
                    "N must be a multiple of it; data parallelism + ZeRO-1 run inside a stage")
    ap.add_argument("--tp", type=int, default=1, help="tensor-parallel (Megatron 'mtp') group size, parallel.tensor=dict(size=tp, mode='mtp'); "
                                                       "must divide --gpus; data parallel size = gpus / tp")
    ap.add_argument("--tp-mode", default="mtp", choices=["mtp", "msp", "fsp"], help="parallel.tensor mode with --tp > 1: msp / fsp shard the activations between the "
                    "linears along the sequence (reduce-scatter / all-gather of token rows instead of all-reduces)")
    ap.add_argument("--zero", type=int, default=None, help="parallel.zero1.size: ranks that share one copy of the sharded optimizer state "
                                                            "(hybrid ZeRO when smaller than the data-parallel size; default: the whole data-parallel group)")
    ap.add_argument("--wp", type=int, default=0, help="ISP weight parallelism (parallel.weight = dict(size=wp)): every rank keeps 1/wp of each layer's weights and "
                                                      "gradients, gathered per layer into a two-slot pool (engine weight_parallel=True); 0 = resident weights")
    ap.add_argument("--scale-on-q", action="store_true", help="A/B only: engine scale_on_q=True (softmax scale folded into the rotary kernel's q, folded-softmax "
                                                               "attention forward; moves a bf16 rounding point away from the reference's -- see engine.py)")
    ap.add_argument("--sp", type=int, default=1, help="sequence-parallel (Ulysses / ISP) group size, parallel.tensor=dict(size=sp, mode='isp'); "
                                                       "must divide --gpus; data parallel size = gpus / sp")
    ap.add_argument("--rccl-channels", type=int, default=0, help="N > 1 diagnostics: cap RCCL's channel count (NCCL_MAX_NCHANNELS / NCCL_MIN_NCHANNELS = n, set before the "
                    "communicator is created): fewer channels = fewer CUs held by a collective, for longer (DESIGN.md section 6); 0 = RCCL's default")
    ap.add_argument("--rs-under-w13-only", action="store_true", help="N > 1 diagnostics: launch a layer bucket's gradient reduce-scatter in front of the next layer's "
                    "w1 | w3 backward products instead of right behind its last weight gradient (engine rs_under_w13)")
    ap.add_argument("--hold-cus", default=None, help="DIAGNOSTIC, one GPU: 'n' or 'n,link_GBps' -- where an 8-GPU run would launch a bucket's reduce-scatter / all-gather, n "
                    "idle workgroups hold n CUs for the time the collective would take (a bucket's 1/8 per xGMI link at link_GBps, default 100), on a side stream: the "
                    "price of a collective's CUs for the products beside it (DESIGN.md section 6.2).  The line says so; results are unchanged")
    ap.add_argument("--gemm-persistent-skip-n", type=int, default=None, help="A/B only: products with this many output columns stay on the plain launch")
    ap.add_argument("--qkv-rotary-fuse", type=int, default=None, help="A/B only: ie_tune_qkv_rotary_fuse (1 = split + rotary in the wqkv product's epilogue, 0 = two launches)")
    ap.add_argument("--dgrad-refill-all", type=int, default=None, help="A/B only: ie_tune_gemm_dgrad_refill_all")
    ap.add_argument("--queue-memset", type=int, default=None, help="A/B only: ie_tune_gemm_queue_memset")
    ap.add_argument("--res-in-epilogue", type=int, default=None, choices=[0, 1], help="A/B only: IE_RES_IN_EPILOGUE")
    ap.add_argument("--attn-bwd-rotary-fuse", type=int, default=None, choices=[0, 1], help="A/B only: IE_ATTN_BWD_ROTARY_FUSE")
    ap.add_argument("--flash-bwd-variant", type=int, default=None, help="A/B only: ie_tune_flash_bwd_variant (4 = delta by its own kernel)")
    ap.add_argument("--adamw-cus", type=int, default=None, help="A/B only: the CUs AdamW may occupy beside the next forward (IE_ADAMW_CUS; 0 = whole chip)")
    ap.add_argument("--adamw-full-buckets", type=int, default=None, help="A/B only: the first buckets' AdamW over the whole chip (IE_ADAMW_FULL_BUCKETS)")
    ap.add_argument("--gemm-group", type=int, default=None, help="A/B only: ie_tune_gemm_group (tile rows per group of the XCD-aware tile order; 0 = default 4)")
    ap.add_argument("--wgrad-ksplit", type=int, default=None, choices=[0, 1], help="A/B only: the weight-gradient products' tail k-split (IE_WGRAD_KSPLIT)")
    ap.add_argument("--ffn-fuse", type=int, default=None, help="A/B only: ie_tune_ffn_fuse mode (bit 0 forward gate, bit 1 the w2 input-gradient epilogue)")
    args = ap.parse_args()
    if args.rccl_channels > 0:   # (inherited by the ranks of a self-launched run; read by RCCL when the communicator is created)
        os.environ["NCCL_MAX_NCHANNELS"] = os.environ["NCCL_MIN_NCHANNELS"] = str(args.rccl_channels)

    from internevo_amd import kernels as K
    from internevo_amd.config import internlm2_7b, llama2_7b, tiny
    from internevo_amd.data import SyntheticLoader
    from internevo_amd.engine import InternLM2Engine

    K.LINEAR_FWD_VARIANT = args.fwd_variant
    if args.gemm_tail_split is not None:
        assert K._L().ie_tune_gemm_tail_split(args.gemm_tail_split) == 0
    if args.gemm_persistent is not None:
        assert K._L().ie_tune_gemm_persistent(args.gemm_persistent) == 0
    if args.gemm_persistent_skip_n is not None:
        assert K._L().ie_tune_gemm_persistent_skip_n(args.gemm_persistent_skip_n) == 0
    if args.ffn_fuse is not None:
        assert K._L().ie_tune_ffn_fuse(args.ffn_fuse) == 0
    if args.gemm_group is not None:
        assert K._L().ie_tune_gemm_group(args.gemm_group) == 0
    if args.dgrad_refill_all is not None:
        assert K._L().ie_tune_gemm_dgrad_refill_all(args.dgrad_refill_all) == 0
    if args.flash_bwd_variant is not None:
        assert K._L().ie_tune_flash_bwd_variant(args.flash_bwd_variant) == 0
    if args.queue_memset is not None:
        assert K._L().ie_tune_gemm_queue_memset(args.queue_memset) == 0
    if args.qkv_rotary_fuse is not None:
        assert K._L().ie_tune_qkv_rotary_fuse(args.qkv_rotary_fuse) == 0
    if args.wgrad_ksplit is not None:
        os.environ["IE_WGRAD_KSPLIT"] = str(args.wgrad_ksplit)
    if args.res_in_epilogue is not None:
        os.environ["IE_RES_IN_EPILOGUE"] = str(args.res_in_epilogue)
    if args.attn_bwd_rotary_fuse is not None:
        os.environ["IE_ATTN_BWD_ROTARY_FUSE"] = str(args.attn_bwd_rotary_fuse)
    if args.adamw_cus is not None:
        os.environ["IE_ADAMW_CUS"] = str(args.adamw_cus)
    if args.adamw_full_buckets is not None:
        os.environ["IE_ADAMW_FULL_BUCKETS"] = str(args.adamw_full_buckets)
    if args.hold_cus:
        if args.gpus != 1:
            raise SystemExit("--hold-cus is the one-GPU stand-in for a collective's CUs")
        os.environ["IE_HOLD_CUS"] = args.hold_cus
    if args.attn_fwd_variant is not None:
        assert K._L().ie_tune_flash_fwd_variant(args.attn_fwd_variant) == 0
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # `python bench.py --gpus N` without a launcher: start the N ranks ourselves (one process per GPU over RCCL) and hand their
        # single JSON line through.  Under torch.distributed.run (the driver's N > 1 command) WORLD_SIZE is set and this is skipped.
        raise SystemExit(_self_launch(args.gpus))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: the launcher's --nproc-per-node must equal --gpus")
    # IE_BENCH_BACKEND=gloo is a TEST HOOK for boxes with one GPU (RCCL refuses two ranks on one device): every rank on cuda:0, the
    # collectives staged through the host by comm.StagedGlooBackend -- it exercises the launch, the N-rank engine and this file's
    # reductions end to end; the line it prints says so and is not a measurement.
    if os.environ.get("IE_BENCH_BACKEND") == "gloo" and world > 1 and not args.staged_test:
        raise SystemExit("IE_BENCH_BACKEND=gloo is the one-GPU test hook of this file: it needs --staged-test on the command line as well (a stray "
                         "environment variable must not turn a benchmark into a host-staged run)")
    staged = args.staged_test and os.environ.get("IE_BENCH_BACKEND") == "gloo" and world > 1
    if staged:
        local_rank = 0
    if torch.cuda.device_count() <= local_rank:
        raise SystemExit(f"rank {rank}: local rank {local_rank} has no GPU ({torch.cuda.device_count()} visible); one process per GPU")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        import torch.distributed as dist

        from internevo_amd.comm import backend_for

        if staged:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=dev)  # "nccl" is RCCL on ROCm
        be = backend_for(None)

        def all_reduce(t, op="sum"):   # over all ranks of the job, through the same Backend the engine uses
            if op == "max":            # (max of non-negative values: gather and reduce locally)
                parts = torch.empty(world * t.numel(), dtype=t.dtype, device=t.device)
                be.all_gather(parts, t.contiguous().view(-1), None).wait()
                t.copy_(parts.view(world, -1).max(dim=0).values.view_as(t))
            else:
                be.all_reduce(t, None).wait()
            return t

    cfg = (internlm2_7b(args.seq_len) if args.config == "7B_internlm2" else llama2_7b(args.seq_len) if args.config == "7B_llama2"
           else tiny(seq_len=min(args.seq_len, 256)))
    cfg.train.fixed_random_dataset_seqlen = True  # concrete synthetic input of the metric
    cfg.train.sp_size = args.sp
    cfg.train.tp_size = args.tp
    cfg.train.tp_mode = args.tp_mode
    cfg.train.pp_size = args.pp
    cfg.train.wp_size = max(args.wp, 1)
    cfg.train.num_chunks = args.num_chunks if args.pp > 1 else 1
    cfg.model.checkpoint = args.checkpoint
    if args.micro_num:
        cfg.train.micro_num = args.micro_num
    tc, mc = cfg.train, cfg.model
    eng = InternLM2Engine(cfg, dev, None, world, rank, seed=1024, batch_wgrad=False if args.no_batch_wgrad else None,
                          merge_micro=None if args.merge_micro is None else bool(args.merge_micro), zero_size=args.zero, scale_on_q=args.scale_on_q,
                          weight_parallel=True if args.wp else None, rs_under_w13=args.rs_under_w13_only)
    if world > 1:
        eng.comm.broadcast_params(eng.params)  # over the data-parallel group (the ranks that hold the same shard)
        eng.sync_master_from_params()
    loader = iter(SyntheticLoader(tc.seq_len, tc.micro_bsz, tc.micro_num, tc.fixed_random_dataset_seqlen, 1_000_000 if args.config != "tiny" else 4000,
                                  data_rank=eng.seqpar.data_rank, data_world_size=eng.seqpar.data_world))

    def sync():
        if world > 1:
            torch.distributed.barrier()
        torch.cuda.synchronize()

    # loss / grad norm / loss scale of EVERY step (warm-up included), kept on the device: three 4-byte stream-ordered copies per step, no
    # host synchronisation inside the timed region (grad_norm = IeStepState byte 32, loss_scale = byte 0)
    traj = torch.zeros(args.warmup + args.steps, 3, dtype=torch.float32, device=dev)
    n_done = [0]

    def one_step():
        batch, labels = next(loader)
        loss = eng.forward_backward(batch, labels)
        eng.step()
        row = traj[n_done[0]]
        row[0:1].copy_(loss.view(-1)[0:1])
        row[1:2].copy_(eng.state[32:36].view(torch.float32))
        row[2:3].copy_(eng.state[0:4].view(torch.float32))
        n_done[0] += 1
        return loss

    for _ in range(args.warmup):
        one_step()
    prof = None
    if not args.no_kernel_timing:
        prof = K.KernelProfiler()
        K.GEMM_PROFILER = prof
    from internevo_amd import comm as C_

    waits = C_.WaitTimer()   # exposed communication: HIP events around every Work.wait() of the timed steps (comm.py; nothing is recorded on one rank)
    C_.TIMER = waits
    sync()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        loss = one_step()
    sync()
    dt = time.perf_counter() - t0
    K.GEMM_PROFILER = None
    C_.TIMER = None
    # per rank and step: milliseconds the waiting stream stood still in Work.wait(), by collective kind; over the ranks: max and mean
    kinds = ("reduce_scatter", "all_gather", "all_reduce", "all_to_all", "send_recv", "broadcast", "other")
    ws = waits.summary()
    exposed = torch.tensor([ws.get(k, (0.0, 0))[0] / args.steps for k in kinds] + [float(sum(v[1] for v in ws.values())) / args.steps], device=dev, dtype=torch.float64)
    exp_max, exp_sum = exposed.clone(), exposed.clone()
    if world > 1:
        t = torch.tensor([dt], device=dev, dtype=torch.float64)
        dt = float(all_reduce(t, "max"))
        all_reduce(exp_max, "max")
        all_reduce(exp_sum)
    st = eng.read_state()
    loss_val = float(loss)
    # what every rank sees of the communicator (rank 0 prints it: a job that silently ran as N independent 1-rank jobs would show here)
    comm_info = {"backend": "none (single process)", "rccl_world_size_per_rank": [1]}
    if world > 1:
        seen = torch.zeros(world, dtype=torch.int64, device=dev)
        seen[rank] = torch.distributed.get_world_size()
        all_reduce(seen)   # SUM of one-hot rows: entry r = the world size rank r reports
        comm_info = {"backend": "gloo, all ranks on ONE GPU, host-staged collectives (IE_BENCH_BACKEND test hook: NOT a measurement)" if staged
                     else torch.distributed.get_backend() + " (RCCL over xGMI)", "rccl_world_size_per_rank": [int(x) for x in seen.tolist()]}
    comm_info.update(data_parallel_size=eng.dp_world, zero_shards_per_bucket=eng.world, zero_replicas=eng.comm.n_replica)
    comm_info["exposed_wait_ms_per_step"] = {
        "what": "time the waiting HIP stream stood still inside Work.wait() per step (events on that stream around every wait of the timed steps), by collective "
                "kind: gradient reduce-scatters (+ the hybrid-ZeRO second hop under all_reduce), parameter all-gathers, norm / tensor-parallel all-reduces, "
                "Ulysses / expert all-to-alls, pipeline send-recv; max and mean over the ranks",
        "max_over_ranks": {k: round(float(exp_max[i]), 3) for i, k in enumerate(kinds) if float(exp_max[i]) > 0},
        "mean_over_ranks": {k: round(float(exp_sum[i]) / world, 3) for i, k in enumerate(kinds) if float(exp_sum[i]) > 0},
        "total_max_over_ranks": round(float(exp_max[: len(kinds)].sum()), 3), "total_mean_over_ranks": round(float(exp_sum[: len(kinds)].sum()) / world, 3),
        "waits_per_step_mean": round(float(exp_sum[-1]) / world, 1)}
    comm_info["rccl_channels"] = args.rccl_channels or "default"
    if eng.hold is not None:
        comm_info["DIAGNOSTIC_held_cus"] = (f"{eng.hold[0]} CUs held by idle workgroups where a data-parallel run launches each bucket's reduce-scatter and all-gather, for the "
                                            f"collective's time at {eng.hold[1]:g} GB/s per xGMI link (bench.py --hold-cus): NOT the default benchmark")
    comm_info["gradient_reduce_scatter_launch"] = "in front of the next layer's w1|w3 backward products" if eng.rs_under_w13 else "behind the bucket's last weight gradient"
    if world > 1 and args.tp == 1 and args.pp == 1 and not eng.wp_mode:
        # data-parallel replicas must hold bit-identical parameters after the timed steps (reduce-scatter -> AdamW on the shard -> all-gather):
        # every rank's checksum, gathered; a broken exchange would show here instead of as a plausible-looking throughput
        eng.drain()
        chk = torch.zeros(world, dtype=torch.float64, device=dev)
        chk[rank] = eng.params.double().abs().sum()
        all_reduce(chk)
        comm_info["params_in_sync_across_ranks"] = bool((chk == chk[0]).all())

    tokens_step = tc.packed_length * tc.micro_num * world // (args.sp * args.tp * args.pp)
    sec_step = dt / args.steps
    total_tps = tokens_step / sec_step
    tgs = total_tps / world
    # reference metric (train/pipeline.py:500-556 + utils/common.py:208-238)
    fac = 4 if mc.checkpoint_layers else 3  # get_megatron_flops counts the recomputed forward (utils/common.py:224-226)
    ref_flops_tok = (fac * ((8 + mc.mlp_ratio * 1.5 * 4) * mc.hidden_size**2 + 4 * tc.seq_len * mc.hidden_size) * mc.num_layers
                     + 6 * mc.hidden_size * mc.vocab_size)
    out = {
        "metric": (f"tokens_per_second (TGS x n_gpus), InternLM2-7B bf16 seq{tc.seq_len} training step" if args.config == "7B_internlm2"
                   else f"tokens_per_second (TGS x n_gpus), LLaMA2-7B (configs/7B_llama2.py) bf16 seq{tc.seq_len} training step" if args.config == "7B_llama2"
                   else "tokens_per_second (tiny plumbing config)"),
        "value": total_tps,
        "unit": "tokens/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": sec_step * 1e3,
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": "bf16",
        "data": "synthetic (reference RandomDataset/PackedDatasetWithCut shape, fixed_random_dataset_seqlen=True), random-init weights",
        "config": {"workload": f"configs/7B_internlm2.py (BASELINE.json configs[1]): InternLM2-7B, seq_len {tc.seq_len}, micro_bsz {tc.micro_bsz} x micro_num {tc.micro_num} per GPU, "
                               f"ZeRO-1 over dp{world}, AdamW + dynamic loss scale + grad clip 1.0" if args.config == "7B_internlm2"
                               else "configs/7B_llama2.py (BASELINE.json configs[2]'s model, tensor size 1 as shipped): LLaMA2-7B, vocab 32000" if args.config == "7B_llama2"
                               else "tiny InternLM2 (hidden 512, 2 layers)",
                   "micro_batch_execution": ("merged: the micro_num micro-batches of a step run as one varlen pass" if eng.mm > 1 else
                                             "sequential gradient accumulation" + (", weight gradients batched over the micro-batches" if eng.batch_wgrad else "")),
                   "tokens_per_step": tokens_step, "parallelism": f"dp{world // (args.sp * args.tp * args.pp)}" + (f" x sp{args.sp} (Ulysses/ISP)" if args.sp > 1 else "") + (f" x tp{args.tp} ({args.tp_mode})" if args.tp > 1 else "")
                   + (f" x pp{args.pp} ({'interleaved ' if args.num_chunks > 1 else ''}1F1B{f', {args.num_chunks} chunks' if args.num_chunks > 1 else ''})" if args.pp > 1 else "")
                   + (f", weight parallel wp{eng.world} (layer weights / gradients sharded, two-slot pool)" if eng.wp_mode else "")},
        "tgs": tgs,
        "tflops_per_gpu_reference_formula": ref_flops_tok * tgs / 1e12,
        "tflops_per_gpu_exact_causal": exact_flops_per_token(mc, tc.seq_len) * tgs / 1e12,
        "frac_bf16_mfma_peak": exact_flops_per_token(mc, tc.seq_len) * tgs / MFMA_PEAK,
        "loss_last_step": loss_val,
        # every step of this run in order, warm-up steps first (this rank's loss; the global grad norm; the loss scale after the step)
        "loss_trajectory": [round(float(x), 5) for x in traj[:, 0].tolist()],
        "grad_norm_trajectory": [round(float(x), 5) for x in traj[:, 1].tolist()],
        "loss_scale_trajectory": [float(x) for x in traj[:, 2].tolist()],
        "grad_norm_last_step": st.grad_norm,
        "loss_scale": st.loss_scale,
        "skipped_steps": st.skipped_total,
        "comm": comm_info,
    }
    if prof is not None:
        s = prof.summary()
        ach = s["flops"] / max(s["seconds"], 1e-12)
        # HBM-side bytes per launch from the PMC passes committed under profiles/ (FETCH_SIZE / WRITE_SIZE, collected and
        # corrected as MI355X_MICROARCH.md prescribes): measured ratio to the algorithmic bytes x this run's algorithmic bytes
        traffic, traffic_note = None, None
        try:
            # measured on THIS workload -- two PMC passes of bench.py itself, every GEMM launch of the step as it runs (tools/gemm_traffic_in_step.py) --
            # and only reported if the measurement is of the kernels THIS run launched (ie_gemm_last_kernel after every timed launch): a file of other
            # schedules is refused (traffic: null), never passed on as this run's number
            TRAFFIC_FILE = "r06_gemm_hbm_traffic.json"
            with open(os.path.join(ROOT, "profiles", TRAFFIC_FILE)) as f:
                tj = json.load(f)
            measured = {k.replace("void ", "").strip() for k in tj["kernels"]}
            seen = set(s["kernels"])
            if measured != seen:
                traffic_note = (f"profiles/{TRAFFIC_FILE} is not a measurement of this run's kernels (measured only: {sorted(measured - seen)}; launched only: "
                                f"{sorted(seen - measured)}): no traffic reported -- re-measure with tools/gemm_traffic_in_step.py")
            else:
                # per PRODUCT of this run (a product = one timed launch of this file; the k-split weight gradients are two kernel launches each)
                traffic = float(tj.get("traffic_bytes_total", tj["traffic_bytes_per_launch"] * tj["gemm_launches"])) / (tj.get("steps_profiled", 3) * s["launches"] / args.steps)
                traffic_note = (f"{traffic / (s['bytes'] / max(s['launches'], 1)):.2f}x the algorithmic bytes per product; rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes over "
                                f"bench.py itself ({tj['gemm_launches']} GEMM kernel launches of {tj.get('steps_profiled', 3)} steps, summed and divided by this run's products per step; profiles/{TRAFFIC_FILE}: the same {len(seen)} kernel instantiations this run "
                                "launched, checked by name -- a committed measurement of this workload, not re-measured by this run; fabric-side L2 misses incl. "
                                "Infinity-Cache hits, FETCH_SIZE doubled per MI355X_MICROARCH.md)")
        except (OSError, KeyError, ValueError) as e:
            traffic_note = f"no HBM-traffic measurement available ({type(e).__name__})"
        out["roofline"] = {
            "kernel": "gemm_dma_k<256,256,...> / gemm_p5_k (LDS-DMA bf16 GEMM; fwd: one wave per SIMD, operand-wise refill of two 64-deep stages, v_mfma_f32_16x16x32_bf16, in a persistent frame where it applies (gemm_p5_k; its wave-private epilogue also carries the SwiGLU gate of the w1|w3 product, the SwiGLU backward of the w2 input gradient, the GQA split + rotary of the wqkv product and the residual adds behind wo / w2 -- those launches' extra work is inside their timed duration, their flops are the product's alone), also the long dgrads (B by transposing reads); other dgrads: 8-wave phased k32 ring; wgrad: one-wave-per-SIMD k32 ring, a half-empty last tile round as two half-k products + a fix-up, all inside the product's timed duration; of every linear layer)",
            "bound": "mfma",
            "achieved": ach / 1e12,
            "peak": MFMA_PEAK / 1e12,
            "unit": "TFLOP/s",
            "frac": ach / MFMA_PEAK,
            "traffic": traffic,
            "traffic_note": traffic_note,
            "algorithmic_bytes_per_launch": s["bytes"] / max(s["launches"], 1),
            "launches": s["launches"],
            "kernels": s["kernels"],
            "avg_launch_us": s["avg_us"],
            "algorithmic_flops_per_launch": s["flops"] / max(s["launches"], 1),
            "share_of_step_time": s["seconds"] / dt,
        }
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        try:
            out["cpu_baseline"] = cpu_baseline(cfg)
            out["cpu_baseline"]["machine"] = "this GPU box's host cores"
            # The reference's OWN CPU path (unmodified trainer, torch CPU kernels) cannot run here -- /root/reference does not exist on the GPU
            # box -- so it was timed in the build container at TWO depths of the 7B shape (tests/golden/make_golden.py --run cpu7b_1layer /
            # cpu7b_2layer) and is extrapolated to the model's depth exactly like the port's figure above: t(L) = t(1) + (L - 1) (t(2) - t(1)).
            rj = [json.load(open(os.path.join(ROOT, "profiles", f"r03_reference_cpu_path_{n}layer.json"))) for n in (1, 2)]
            t1, t2 = rj[0]["sec_per_step_timed"], rj[1]["sec_per_step_timed"]
            full = t1 + (cfg.model.num_layers - 1) * (t2 - t1)
            out["cpu_baseline"]["reference"] = {
                "kind": "reference", "value": rj[0]["tokens_per_step"] / full, "unit": "tokens/s", "cores": rj[0]["threads"],
                "machine": f"build container ({rj[0]['host_cores']} host cores; not this box)",
                "sample": f"the unmodified reference training step on a 7B-shaped model with 1 layer ({t1:.2f} s) and 2 layers ({t2:.2f} s) per "
                          f"{rj[0]['tokens_per_step']}-token step, linearly extrapolated to {cfg.model.num_layers} layers ({full:.1f} s); torch CPU bf16, "
                          f"{rj[0]['threads']} threads",
                "source": "profiles/r03_reference_cpu_path_{1,2}layer.json"}
        except Exception as e:  # the baseline must never take the measurement down with it
            out["cpu_baseline"] = {"value": None, "unit": "tokens/s", "cores": torch.get_num_threads(), "kind": "port", "sample": f"failed: {e!r}"}
    if rank == 0:
        print(json.dumps(out), flush=True)
    if world > 1:
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
