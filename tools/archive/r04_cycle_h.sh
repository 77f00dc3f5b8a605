# round-4 GPU cycle H: the forward product on the 16x16x32 schedule in the product path -- the full GPU suite as the driver runs it, the default bench line, the same under rocprofv3
cd $GRAFT_REPO_ROOT
O=gpurun_out/r04h; mkdir -p $O
export TMPDIR=/tmp
( time timeout 500 python -m pytest tests -q -m gpu --durations=15 2>&1 | grep -v "amdgpu.ids\|socket.cpp" | tail -60 ) > $O/full.log 2>&1; tail -6 $O/full.log
timeout 300 python bench.py --steps 10 --warmup 3 > $O/bench_line.json 2> $O/bench_err.log; echo "bench rc=$?"; cut -c1-400 $O/bench_line.json
rm -rf /tmp/prof_h
timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/prof_h -o bench -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline > $O/bench_under_rocprof.json 2> $O/rocprof_err.log
DB=$(find /tmp/prof_h -name "*.db" | head -1)
python3 tools/rocprof_summary.py "$DB" $O/kernel_stats.md "rocprofv3 --kernel-trace --stats -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline" | head -14
