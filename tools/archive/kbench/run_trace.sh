# usage: run_trace.sh <tag> <kbench args...>  -- per-kernel durations of a kbench invocation -> gpurun_out/<tag>_trace.md
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
TAG=$1; shift
rm -rf /tmp/tr_$TAG
timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/tr_$TAG -o r -- tools/kbench/kbench "$@" > gpurun_out/${TAG}_trace.log 2>&1
DB=$(find /tmp/tr_$TAG -name "*.db" | head -1)
python3 tools/rocprof_summary.py "$DB" gpurun_out/${TAG}_trace.md "rocprofv3 --kernel-trace --stats -- kbench $*" > /dev/null 2>>gpurun_out/${TAG}_trace.log
cut -c1-60,100-200 gpurun_out/${TAG}_trace.md | head -20
