#!/bin/bash
# Round 6: ONE parameterised script for every GPU lease (the per-cycle copies of rounds 4 / 5 are gone).
#   usage (on the GPU box, from the repo root):  bash tools/r06_gpu.sh <job> [<tag>] [job arguments...]
# Everything is written under gpurun_out/<tag>/ (merged back by gpurun); what is to be judged is then copied to profiles/ by hand.
# jobs
#   probe      counters the profiler knows for the fabric side, the weight-gradient round-quantisation emulation (kbench), fabric vs DRAM read requests of
#              three GEMM shapes, the bench line, a kernel trace whose launch ORDER is kept (which kernels surround torch's fill kernels)
#   cycle      the validation cycle: GPU suite, smoke, default bench line, the same under rocprofv3 --kernel-trace --stats, two PMC traffic passes
#   abab       in-step A B A B of one bench.py flag under rocprofv3 --kernel-trace:  abab <tag> "<flag> <value A>" "<flag> <value B>" [kernel grep]
#   hold       the step with 0 / 16 / 32 CUs held where the collectives of an 8-GPU run would be, persistent GEMM frame on / off, twice
#   extras     bench.py --seq-len 32768 --checkpoint 1.0 --micro-num 1; tools/moe_bench.py with bf16 and with opt-in fp8 experts
#   sq         SQ counters of the matrix kernels in the step (one PMC pass of bench.py)
#   envab      in-step A B A B of an environment setting:  envab <tag> "VAR=a" "VAR=b"
#   flagab     A B A B of two bench.py flag sets without the profiler:  flagab <tag> "<flags A>" "<flags B>"
#   sweep      one bench.py flag over several values without the profiler, the list twice:  sweep <tag> <flag> <v1> <v2> ...
#   kab        kbench lines:  kab <tag> <kbench arguments ...>
#   tests      a subset of the GPU suite:  tests <tag> <pytest -k expression>
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd "$(dirname "$0")/.."
JOB=${1:-cycle}; TAG=${2:-r06}; shift 2 2>/dev/null
O=gpurun_out/$TAG; mkdir -p "$O"
export TMPDIR=/tmp
K=tools/kbench/kbench
summ() { python3 tools/rocprof_summary.py "$(find "$1" -name '*.db' | head -1)" "$2" "$3" > /dev/null; }

bench_traced() {   # bench_traced <outfile prefix> <bench flags...>: 1 warm-up + 2 timed steps under the kernel trace
  local pre=$1; shift
  rm -rf /tmp/prof_t
  timeout 400 rocprofv3 --kernel-trace --stats -d /tmp/prof_t -o bench -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-kernel-timing "$@" > "${pre}_line.json" 2> "${pre}.err"
  summ /tmp/prof_t "${pre}_kernel_stats.md" "rocprofv3 --kernel-trace --stats -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-kernel-timing $*"
  python3 tools/gpu_idle_from_trace.py "$(find /tmp/prof_t -name '*.db' | head -1)" > "${pre}_idle.md" 2>&1
  python3 tools/step_sequence.py "$(find /tmp/prof_t -name '*.db' | head -1)" "${pre}_sequence.txt" 2>&1 | tail -2
}

case $JOB in
probe)
  ( cd /tmp && timeout 120 rocprofv3 -L 2>/dev/null | grep -i -E "mall|dram|TCC_EA0|TCC_REQ|TCC_HIT|TCC_MISS|HBM|FETCH|WRITE_SIZE|TCC_BUBBLE" ) > "$O/counters_fabric.txt" 2>&1
  wc -l "$O/counters_fabric.txt"
  {   # weight gradients: whole product against (whole rounds) + (the 128 remainder tiles as 256 half-k blocks); wqkv 384 tiles, w2 896 tiles
    for s in "6144 4096 16384" "4096 4096 16384" "4096 4096 8192" "4096 14336 16384" "4096 12288 16384" "28672 4096 16384"; do
      set -- $s
      timeout 120 $K gemm --m $1 --n $2 --k $3 --layout tn --variants 17 --iters 30 2>&1 | grep -o '"M": [0-9]*\|"N": [0-9]*\|"K": [0-9]*\|"us": [0-9.]*\|"tflops": [0-9.]*' | paste -s -d' '
    done
  } > "$O/wgrad_round_emulation.log" 2>&1
  cat "$O/wgrad_round_emulation.log"
  for s in "nt 16384 28672 4096 -1" "tn 28672 4096 16384 17" "nn 16384 14336 4096 -1"; do
    set -- $s
    rm -rf /tmp/prof_c
    ( cd /tmp && timeout 200 rocprofv3 --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_DRAM_sum TCC_EA0_RDREQ_32B_sum TCC_BUBBLE_sum --kernel-trace -d /tmp/prof_c -o r -- "$GRAFT_REPO_ROOT/$K" gemm --m $2 --n $3 --k $4 --layout $1 --variants $5 --iters 5 ) > /dev/null 2> "$O/pmc_$1.err"
    summ /tmp/prof_c "$O/pmc_dram_$1.md" "rocprofv3 --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_DRAM_sum TCC_EA0_RDREQ_32B_sum TCC_BUBBLE_sum -- kbench gemm $*"
    grep "gemm_" "$O/pmc_dram_$1.md" | grep "TCC" | cut -c1-60,100-200
  done
  timeout 600 python bench.py --steps 10 --warmup 3 --no-cpu-baseline > "$O/bench_line.json" 2> "$O/bench.err"; cut -c1-300 "$O/bench_line.json"
  bench_traced "$O/traced"
  python3 - "$(find /tmp/prof_t -name '*.db' | head -1)" > "$O/fill_neighbours.txt" <<'PY'
import sqlite3, sys, collections
c = sqlite3.connect(sys.argv[1])
rows = c.execute("select name, start, end from kernels order by start").fetchall()
seen = collections.Counter()
for i, (n, s, e) in enumerate(rows):
    if "FillFunctor" in n or "bfloat16_copy" in n or "copyBuffer" in n or "fillBuffer" in n:
        prev = rows[i - 1][0][:60] if i else "-"
        nxt = rows[i + 1][0][:60] if i + 1 < len(rows) else "-"
        seen[(n[:90], round((e - s) / 1e3, -1), prev, nxt)] += 1
for (n, us, p, x), k in sorted(seen.items(), key=lambda t: -t[1])[:40]:
    print(k, "x", us, "us", n, "| after:", p, "| before:", x)
PY
  head -30 "$O/fill_neighbours.txt"
  ;;
cycle)
  timeout 1200 python -m pytest tests -x -q -m gpu > "$O/tests.log" 2>&1; echo "tests rc=$?"; tail -3 "$O/tests.log"
  timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > "$O/smoke.log" 2>&1; echo "smoke rc=$?"; tail -2 "$O/smoke.log"
  timeout 900 python bench.py --steps 20 --warmup 3 > "$O/bench_line.json" 2> "$O/bench.err"; echo "bench rc=$?"; cut -c1-300 "$O/bench_line.json"
  rm -rf /tmp/prof_x
  timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/prof_x -o bench -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline > "$O/bench_under_rocprof.json" 2> "$O/rocprof.err"
  summ /tmp/prof_x "$O/kernel_stats.md" "rocprofv3 --kernel-trace --stats -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline"; head -24 "$O/kernel_stats.md"
  rm -rf /tmp/tf /tmp/tw
  timeout 600 rocprofv3 --pmc FETCH_SIZE --kernel-trace -d /tmp/tf -o r -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-kernel-timing > /dev/null 2> "$O/pmc_f.err"
  timeout 600 rocprofv3 --pmc WRITE_SIZE --kernel-trace -d /tmp/tw -o r -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-kernel-timing > /dev/null 2> "$O/pmc_w.err"
  python3 tools/gemm_traffic_in_step.py "$(find /tmp/tf -name '*.db' | head -1)" "$(find /tmp/tw -name '*.db' | head -1)" "$O/gemm_hbm_traffic.json" | tail -5
  ;;
abab)
  A=$1; B=$2; PAT=${3:-gemm_\|flash_\|swiglu\|adamw}
  for rep in 1 2; do
    for arm in A B; do
      [ $arm = A ] && F=$A || F=$B
      bench_traced "$O/${arm}${rep}" $F
      echo "== $arm ($F) rep $rep"; grep -o '"ms_per_step": [0-9.]*\|"loss_last_step": [0-9.]*' "$O/${arm}${rep}_line.json" | tr '\n' ' '; echo
      grep "$PAT" "$O/${arm}${rep}_kernel_stats.md" | cut -c1-72,110-170
    done
  done 2>&1 | tee "$O/abab.log"
  ;;
hold)   # what the CUs a collective holds cost the step: bench.py --hold-cus n (idle workgroups where an 8-GPU run launches its reduce-scatters / all-gathers)
  for rep in 1 2; do
    for pers in 1 0; do
      for h in 0 16 32; do
        [ $h = 0 ] && HF="" || HF="--hold-cus $h,100"
        timeout 300 python bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-kernel-timing --gemm-persistent $pers $HF > "$O/p${pers}_h${h}_${rep}.json" 2> "$O/p${pers}_h${h}_${rep}.err"
        echo "persistent $pers hold $h rep $rep: $(grep -o '"ms_per_step": [0-9.]*' "$O/p${pers}_h${h}_${rep}.json")"
      done
    done
  done 2>&1 | tee "$O/hold.log"
  ;;
extras)   # the long-context data point (seq 32768, every layer checkpointed, one sequence per step) and the MoE family at full size (bf16 experts; opt-in fp8 experts)
  timeout 900 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --seq-len 32768 --checkpoint 1.0 --micro-num 1 > "$O/long_context_line.json" 2> "$O/long_context.err"; cut -c1-400 "$O/long_context_line.json"
  timeout 900 python tools/moe_bench.py --steps 4 --warmup 2 > "$O/moe_bench_line.json" 2> "$O/moe_bench.err"; cat "$O/moe_bench_line.json"
  timeout 900 python tools/moe_bench.py --steps 4 --warmup 2 --expert-fp8 > "$O/moe_bench_fp8_line.json" 2> "$O/moe_bench_fp8.err"; cat "$O/moe_bench_fp8_line.json"
  ;;
sq)   # SQ counters (matrix pipe busy, waits, LDS conflicts) of the matrix kernels in the step: one PMC pass of bench.py itself, counters only
  rm -rf /tmp/sq
  timeout 600 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --kernel-trace -d /tmp/sq -o r -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-kernel-timing > "$O/line.json" 2> "$O/pmc.err"
  python3 tools/sq_counters_in_step.py "$(find /tmp/sq -name '*.db' | head -1)" "$O/sq_counters_in_step.md"
  ;;
envab)   # in-step A B A B of an ENVIRONMENT setting (plain bench lines, 8 timed steps):  envab <tag> "VAR=a" "VAR=b"
  for rep in 1 2; do
    for arm in "$1" "$2"; do
      env $arm timeout 300 python bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-kernel-timing > "$O/line.json" 2> "$O/err.log"
      echo "$arm rep $rep: $(grep -o '"ms_per_step": [0-9.]*' "$O/line.json") $(grep -o 'optimizer stream priority[^)]*)' "$O/err.log" "$O/line.json" 2>/dev/null | head -1)"
    done
  done 2>&1 | tee "$O/envab.log"
  ;;
flagab)   # A B A B of two bench.py flag sets WITHOUT the profiler (plain bench lines, 8 timed steps; the flags are all of the run's extra flags):  flagab <tag> "<flags A>" "<flags B>"
  for rep in 1 2; do
    for arm in A B; do
      [ $arm = A ] && F=$1 || F=$2
      timeout 300 python bench.py --steps 8 --warmup 2 --no-cpu-baseline $F > "$O/${arm}${rep}_line.json" 2> "$O/${arm}${rep}.err"
      echo "$arm ($F) rep $rep: $(grep -o '"ms_per_step": [0-9.]*' "$O/${arm}${rep}_line.json")"
    done
  done 2>&1 | tee "$O/flagab.log"
  ;;
sweep)   # one bench.py flag over several values, without the profiler, the whole list twice:  sweep <tag> <flag> <v1> <v2> ...
  FL=$1; shift
  for rep in 1 2; do
    for v in "$@"; do
      timeout 300 python bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-kernel-timing $FL $v > "$O/v${v}_${rep}_line.json" 2> "$O/v${v}_${rep}.err"
      echo "$FL $v rep $rep: $(grep -o '"ms_per_step": [0-9.]*' "$O/v${v}_${rep}_line.json")"
    done
  done 2>&1 | tee "$O/sweep.log"
  ;;
kab)
  timeout 600 $K "$@" 2>&1 | tee "$O/kbench.log"
  ;;
tests)
  timeout 1200 python -m pytest tests -x -q -m gpu -p no:cacheprovider -k "$1" 2>&1 | tee "$O/tests.log" | tail -15
  ;;
*) echo "unknown job $JOB"; exit 2;;
esac
