cd $GRAFT_REPO_ROOT
O=gpurun_out/r05v; mkdir -p $O
export IE_REF_LIB=tools/kbench/ab/lib_r04.so
K=tools/kbench/kbench
{
timeout 100 $K fwd --variants 0,2,4,2,4 --iters 30
echo "== other shapes"
timeout 100 $K fwd --variants 0,4 --iters 3 --ragged 1 --seqs 8 --len 3000
timeout 100 $K fwd --variants 0,4 --iters 3 --d 64 --hq 32 --hkv 32
timeout 100 $K fwd --variants 0,4 --iters 3 --causal 0 --len 2048
timeout 100 $K fwd --variants 0,4 --iters 3 --hq 8 --hkv 8 --len 300 --ragged 1 --seqs 7
} > $O/fwd8.log 2>&1
cut -c1-400 $O/fwd8.log | sed 's/"T": 16384, "seqs": 4, "ragged": 0, "hq": 32, "hkv": 8, "d": 128, "causal": 1, //; s/"bench": "flash_fwd", //'
