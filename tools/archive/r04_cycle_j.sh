# round-4 GPU cycle J (final): the full GPU suite as the driver runs it, smoke(), the default bench line, the same under rocprofv3
cd $GRAFT_REPO_ROOT
O=gpurun_out/r04j; mkdir -p $O
export TMPDIR=/tmp
( time timeout 500 python -m pytest tests -x -q -m gpu --durations=12 2>&1 | grep -v "amdgpu.ids\|socket.cpp" | tail -40 ) > $O/full.log 2>&1; tail -5 $O/full.log
timeout 120 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
timeout 300 python bench.py --steps 10 --warmup 3 > $O/bench_line.json 2> $O/bench_err.log; echo "bench rc=$?"; cut -c1-330 $O/bench_line.json
rm -rf /tmp/prof_j
timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/prof_j -o bench -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline > $O/bench_under_rocprof.json 2> $O/rocprof_err.log
DB=$(find /tmp/prof_j -name "*.db" | head -1)
python3 tools/rocprof_summary.py "$DB" $O/kernel_stats.md "rocprofv3 --kernel-trace --stats -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline" | head -12
