#!/bin/bash
# The first run on a node with N > 1 MI355X (nothing with more than one RCCL rank has run yet: the pool this was built on has 1-GPU boxes).  Not run so far.
# What it does, in the order the answers are needed (DESIGN.md sections 6.2 and 7):
#   1. the scaling curve of the default line, N = 1, 2, 4, 8 (the driver's own command);
#   2. at the largest N: the persistent GEMM frame on / off (a collective's kernels hold CUs while it runs: the frame takes its tiles from queues and should lose
#      the same one round as the plain launches, profiles/r05_gemm_persistent_cu_mask.log -- this checks it under real RCCL kernels);
#   3. RCCL's channel count capped (fewer CUs held, for longer) and the reduce-scatter placement under the 7-round w1 | w3 products only;
#   3b. (round 6) how many CUs AdamW takes beside the next forward (--adamw-cus), now that its all-gathers hold CUs too;
#   4. one rocprofv3 kernel trace of rank 0's process group at the largest N: how many workgroups the collectives' kernels run, for how long, beside which products.
# usage: bash tools/first_multi_gpu_run.sh [N_MAX=8] [OUT=multi_gpu_out]
set -u
cd "$(dirname "$0")/.."
NMAX=${1:-8}; OUT=${2:-multi_gpu_out}; mkdir -p "$OUT"
export HSA_ENABLE_IPC_MODE_LEGACY=0 TMPDIR=/tmp
run() {   # run <tag> <n> <bench flags...>
  local tag=$1 n=$2; shift 2
  if [ "$n" = 1 ]; then python bench.py --gpus 1 --steps 10 --warmup 3 --no-cpu-baseline "$@" > "$OUT/$tag.json" 2> "$OUT/$tag.err"
  else python -m torch.distributed.run --nnodes=1 --nproc-per-node "$n" --master-addr 127.0.0.1 --master-port $((29500 + RANDOM % 200)) bench.py --gpus "$n" --steps 10 --warmup 3 --no-cpu-baseline "$@" > "$OUT/$tag.json" 2> "$OUT/$tag.err"; fi
  python3 - "$OUT/$tag.json" "$tag" <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    c = d.get("comm", {})
    print(f"{sys.argv[2]:28s} n={d['n_gpus']} {d['value']:10.0f} tokens/s  {d['ms_per_step']:8.1f} ms/step  per GPU {d['value'] / d['n_gpus']:8.0f}   comm: {json.dumps(c)[:160]}")
except Exception as e:
    print(f"{sys.argv[2]:28s} FAILED ({e}); see the .err file")
PY
}
for n in 1 2 4 8; do [ "$n" -le "$NMAX" ] && run "scale_n$n" "$n"; done
run "persistent_off_n$NMAX" "$NMAX" --gemm-persistent 0
run "persistent_on_n$NMAX" "$NMAX" --gemm-persistent 1
for ch in 8 16 32; do run "rccl_channels_${ch}_n$NMAX" "$NMAX" --rccl-channels $ch; done
run "rs_under_w13_n$NMAX" "$NMAX" --rs-under-w13-only
# round 6: the optimizer's CUs beside the next forward -- on N ranks the update is 1 / N as long, and its all-gathers hold CUs of their own: whole chip (0) against the
# one-GPU default (128) and fewer
for cus in 0 64 128; do run "adamw_cus_${cus}_n$NMAX" "$NMAX" --adamw-cus $cus; done
rm -rf /tmp/prof_mg
rocprofv3 --kernel-trace --stats -d /tmp/prof_mg -o bench -- python -m torch.distributed.run --nnodes=1 --nproc-per-node "$NMAX" --master-addr 127.0.0.1 --master-port 29777 \
  bench.py --gpus "$NMAX" --steps 2 --warmup 1 --no-cpu-baseline --no-kernel-timing > "$OUT/trace_line.json" 2> "$OUT/trace.err"
DB=$(find /tmp/prof_mg -name "*.db" | head -1)
[ -n "$DB" ] && python3 tools/step_sequence.py "$DB" "$OUT/step_sequence_n$NMAX.txt"
[ -n "$DB" ] && python3 tools/rocprof_summary.py "$DB" "$OUT/kernel_stats_n$NMAX.md" "rocprofv3 --kernel-trace --stats -- torchrun --nproc-per-node $NMAX bench.py --steps 2 --warmup 1" | head -24
echo "outputs under $OUT/ ; copy what is to be judged into profiles/"
