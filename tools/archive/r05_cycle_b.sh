# round-5 GPU cycle: the full GPU suite as the driver runs it, smoke(), the default bench line, the same under rocprofv3
cd $GRAFT_REPO_ROOT
TAG=${1:-r05b}
O=gpurun_out/$TAG; mkdir -p $O
export TMPDIR=/tmp
( time timeout 900 python -m pytest tests -x -q -m gpu --durations=8 2>&1 | grep -v "amdgpu.ids\|socket.cpp" | tail -30 ) > $O/full.log 2>&1; tail -14 $O/full.log
timeout 120 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
timeout 300 python bench.py --steps 10 --warmup 3 > $O/bench_line.json 2> $O/bench_err.log; echo "bench rc=$?"; cut -c1-330 $O/bench_line.json
rm -rf /tmp/prof_x
timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/prof_x -o bench -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline > $O/bench_under_rocprof.json 2> $O/rocprof_err.log
DB=$(find /tmp/prof_x -name "*.db" | head -1)
python3 tools/rocprof_summary.py "$DB" $O/kernel_stats.md "rocprofv3 --kernel-trace --stats -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline" | head -14
