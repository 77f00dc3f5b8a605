cd $GRAFT_REPO_ROOT
O=gpurun_out/r05z; mkdir -p $O
export IE_REF_LIB=tools/kbench/ab/lib_r04.so
K=tools/kbench/kbench
{
for rep in 1 2; do
echo "== base"; timeout 100 $K fwd --variants 2,4 --iters 30
echo "== alt"; IE_LIB=tools/kbench/ab/lib_f8alt.so timeout 100 $K fwd --variants 4 --iters 30
done
} > $O/fwd8.log 2>&1
cut -c1-400 $O/fwd8.log | sed 's/"T": 16384, "seqs": 4, "ragged": 0, "hq": 32, "hkv": 8, "d": 128, "causal": 1, //; s/"bench": "flash_fwd", //'
