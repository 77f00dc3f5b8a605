cd $GRAFT_REPO_ROOT
O=gpurun_out/r05f; mkdir -p $O
K=tools/kbench/kbench
export TMPDIR=/tmp
for L in sp_same; do
rm -rf /tmp/tr_sp; IE_LIB=tools/kbench/ab/lib_$L.so timeout 120 rocprofv3 --kernel-trace --stats -d /tmp/tr_sp -o r -- $K bwd --variants 2 --iters 10 > $O/trace_$L.log 2>&1
python3 tools/rocprof_summary.py "$(find /tmp/tr_sp -name '*.db' | head -1)" $O/trace_$L.md "rocprofv3 --kernel-trace --stats -- kbench bwd --variants 2 --iters 10 ($L)" | grep flash | cut -c1-160
done
