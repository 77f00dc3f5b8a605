cd $GRAFT_REPO_ROOT
O=gpurun_out/r05h; mkdir -p $O
export IE_REF_LIB=tools/kbench/ab/lib_r04.so
K=tools/kbench/kbench
{
timeout 100 $K bwd --variants 0,2,3,0,2,3 --iters 20
echo "== other shapes"
timeout 100 $K bwd --variants 0,2,3 --iters 3 --ragged 1 --seqs 8 --len 3000
timeout 100 $K bwd --variants 0,2 --iters 3 --d 64 --hq 32 --hkv 32
timeout 100 $K bwd --variants 0,2 --iters 3 --causal 0 --len 2048
timeout 100 $K bwd --variants 0,2 --iters 3 --hq 16 --hkv 8 --len 1000 --ragged 1 --seqs 5
timeout 100 $K bwd --variants 0,2,3 --iters 3 --hq 8 --hkv 8 --len 300 --ragged 1 --seqs 7
} > $O/spill.log 2>&1
cut -c1-330 $O/spill.log | sed 's/"T": 16384, "seqs": 4, "ragged": 0, "hq": 32, "hkv": 8, "d": 128, "causal": 1, //'
export TMPDIR=/tmp
rm -rf /tmp/tr_sp; timeout 120 rocprofv3 --kernel-trace --stats -d /tmp/tr_sp -o r -- $K bwd --variants 2 --iters 10 > $O/trace.log 2>&1
python3 tools/rocprof_summary.py "$(find /tmp/tr_sp -name '*.db' | head -1)" $O/spill_kernel_trace.md "rocprofv3 --kernel-trace --stats -- kbench bwd --variants 2 --iters 10" | grep flash | cut -c1-160
