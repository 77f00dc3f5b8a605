# round-4 GPU cycle D: the full GPU suite as the driver runs it (without -x, with durations), the default bench line, the same under rocprofv3, the two PMC traffic passes
cd $GRAFT_REPO_ROOT
O=gpurun_out/r04d; mkdir -p $O
export TMPDIR=/tmp
( time timeout 1000 python -m pytest tests -q -m gpu --durations=30 2>&1 | grep -v "amdgpu.ids\|socket.cpp" | tail -150 ) > $O/full.log 2>&1; tail -6 $O/full.log
timeout 400 python bench.py --steps 10 --warmup 3 > $O/bench_line.json 2> $O/bench_err.log; echo "bench rc=$?"; cut -c1-300 $O/bench_line.json
rm -rf /tmp/prof_d
timeout 400 rocprofv3 --kernel-trace --stats -d /tmp/prof_d -o bench -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline > $O/bench_under_rocprof.json 2> $O/rocprof_err.log
DB=$(find /tmp/prof_d -name "*.db" | head -1)
python3 tools/rocprof_summary.py "$DB" $O/kernel_stats.md "rocprofv3 --kernel-trace --stats -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline" | head -12
for C in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/tr_$C
  timeout 400 rocprofv3 --pmc $C --kernel-trace -d /tmp/tr_$C -o r -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-kernel-timing > $O/bench_under_pmc_$C.json 2> $O/pmc_$C.err
done
python3 tools/gemm_traffic_in_step.py "$(find /tmp/tr_FETCH_SIZE -name '*.db' | head -1)" "$(find /tmp/tr_WRITE_SIZE -name '*.db' | head -1)" $O/gemm_hbm_traffic.json | head -40
