cd $GRAFT_REPO_ROOT
O=gpurun_out/r05ab; mkdir -p $O
export IE_REF_LIB=tools/kbench/ab/lib_r04.so
K=tools/kbench/kbench
{
for rep in 1 2; do
echo "== base"; timeout 100 $K fwd --variants 4 --iters 30
for a in 8 16 32 48 64 65; do echo "== abl $a"; IE_LIB=tools/kbench/ab/lib_f8abl$a.so timeout 100 $K fwd --variants 4 --iters 30; done
done
} > $O/fwd8abl.log 2>&1
cut -c1-100 $O/fwd8abl.log | sed 's/"T": 16384, "seqs": 4, "ragged": 0, "hq": 32, "hkv": 8, "d": 128, "causal": 1, //; s/"bench": "flash_fwd", //'
