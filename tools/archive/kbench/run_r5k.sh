cd $GRAFT_REPO_ROOT
O=gpurun_out/r05k; mkdir -p $O
K=tools/kbench/kbench
export TMPDIR=/tmp
for L in nf0 nf1 nf2 sp_nt; do
for v in 2 3; do
rm -rf /tmp/tr_sp; IE_LIB=tools/kbench/ab/lib_$L.so timeout 120 rocprofv3 --kernel-trace --stats -d /tmp/tr_sp -o r -- $K bwd --variants $v --iters 10 > $O/trace_$L.log 2>&1
python3 tools/rocprof_summary.py "$(find /tmp/tr_sp -name '*.db' | head -1)" $O/trace_${L}_v$v.md "rocprofv3 --kernel-trace --stats -- kbench bwd --variants $v --iters 10 ($L)" | grep "true>\|from_ds" | cut -c1-50,110-160 | sed "s/^/$L v$v /"
done; done
