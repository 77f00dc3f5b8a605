cd $GRAFT_REPO_ROOT
O=gpurun_out/r05aa; mkdir -p $O
export IE_REF_LIB=tools/kbench/ab/lib_r04.so
K=tools/kbench/kbench
{
for rep in 1 2; do
timeout 100 $K fwd --variants 2,4,6 --iters 30
done
echo "== full attention 4 x 2048"; timeout 100 $K fwd --variants 4,6 --iters 30 --len 2048 --causal 0
echo "== ragged"; timeout 100 $K fwd --variants 0,4,6 --iters 30 --seqs 8 --len 3000 --ragged 1
echo "== long loop"; timeout 100 $K fwd --variants 2,4,6 --iters 3000
} > $O/fwd8p.log 2>&1
cut -c1-400 $O/fwd8p.log | sed 's/"T": 16384, "seqs": 4, "ragged": 0, "hq": 32, "hkv": 8, "d": 128, "causal": 1, //; s/"bench": "flash_fwd", //'
timeout 600 python -m pytest tests/test_kernels_gpu.py -q -m gpu -k "flash" -x -p no:cacheprovider 2>&1 | tail -5
