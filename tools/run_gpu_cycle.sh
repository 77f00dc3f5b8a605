# One full GPU validation cycle: GPU test-suite, default bench line, the same bench under rocprofv3 --kernel-trace --stats.
# usage (on the GPU box, from the repo root): bash tools/run_gpu_cycle.sh <tag>
cd $GRAFT_REPO_ROOT
TAG=${1:-cycle}
mkdir -p gpurun_out/$TAG
export TMPDIR=/tmp
timeout 900 python -m pytest tests -x -q -m gpu > gpurun_out/$TAG/tests.log 2>&1; echo "tests rc=$?"; tail -3 gpurun_out/$TAG/tests.log
timeout 600 python bench.py --steps 10 --warmup 3 > gpurun_out/$TAG/bench_line.json 2> gpurun_out/$TAG/bench_err.log; echo "bench rc=$?"; cut -c1-400 gpurun_out/$TAG/bench_line.json
rm -rf /tmp/prof_$TAG
timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/prof_$TAG -o bench -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/$TAG/bench_under_rocprof.json 2> gpurun_out/$TAG/rocprof_err.log
DB=$(find /tmp/prof_$TAG -name "*.db" | head -1)
python3 tools/rocprof_summary.py "$DB" gpurun_out/$TAG/kernel_stats.md "rocprofv3 --kernel-trace --stats -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline" | head -16
