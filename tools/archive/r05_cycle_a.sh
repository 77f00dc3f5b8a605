# round-5 GPU cycle A (measurement only, HEAD = the round-4 kernels): the atomic / partial-buffer probe for a fused attention backward, fresh SQ counters
# on the three attention kernels at the bench shape, the in-step GEMM traffic of the kernels HEAD really runs, and the long input-gradient product
# on the 32x32x16 (-4) vs 16x16x32 (-5) refill schedule IN THE STEP under rocprofv3 --kernel-trace (A B A B on one box, kernel microseconds)
cd $GRAFT_REPO_ROOT
O=gpurun_out/r05a; mkdir -p $O
export TMPDIR=/tmp
timeout 120 tools/probes/atomic_rate > $O/atomic_rate.jsonl 2>&1; tail -24 $O/atomic_rate.jsonl | cut -c1-200
timeout 100 tools/kbench/kbench fwd --variants 0,2,3 --iters 20 > $O/kbench_fwd.jsonl 2>&1; cut -c1-250 $O/kbench_fwd.jsonl
timeout 100 tools/kbench/kbench bwd --variants 0,1 --iters 20 > $O/kbench_bwd.jsonl 2>&1; cut -c1-250 $O/kbench_bwd.jsonl
bash tools/kbench/run_pmc.sh r05a/bwd bwd --iters 3 --variants 0
bash tools/kbench/run_pmc.sh r05a/fwd fwd --iters 3 --variants 2,3
for C in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/tr_$C
  timeout 400 rocprofv3 --pmc $C --kernel-trace -d /tmp/tr_$C -o r -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-kernel-timing > $O/bench_under_pmc_$C.json 2> $O/pmc_$C.err
done
python3 tools/gemm_traffic_in_step.py "$(find /tmp/tr_FETCH_SIZE -name '*.db' | head -1)" "$(find /tmp/tr_WRITE_SIZE -name '*.db' | head -1)" $O/gemm_hbm_traffic.json | head -50
cp internevo_amd/csrc/libinternevo_hip.so /tmp/lib_head.so
for rep in 1 2; do
  for arm in head dgrad19; do
    if [ $arm = head ]; then cp /tmp/lib_head.so internevo_amd/csrc/libinternevo_hip.so; else cp tools/kbench/ab/lib_dgrad19.so internevo_amd/csrc/libinternevo_hip.so; fi
    rm -rf /tmp/prof_ab
    timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/prof_ab -o bench -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline > $O/abab_${arm}_${rep}_line.json 2> $O/abab_${arm}_${rep}.err
    DB=$(find /tmp/prof_ab -name "*.db" | head -1)
    python3 tools/rocprof_summary.py "$DB" $O/abab_${arm}_${rep}_kernel_stats.md "rocprofv3 --kernel-trace --stats -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline  [library: $arm]" | grep "gemm_dma_k\|flash" | cut -c1-150
  done
done
cp /tmp/lib_head.so internevo_amd/csrc/libinternevo_hip.so
