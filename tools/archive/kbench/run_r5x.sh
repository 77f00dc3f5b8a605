cd $GRAFT_REPO_ROOT
O=gpurun_out/r05x; mkdir -p $O
K=tools/kbench/kbench
{
echo "== full"; timeout 100 $K fwd --variants 4 --iters 30
for a in 1 2 3; do echo "== abl $a"; IE_LIB=tools/kbench/ab/lib_f8abl$a.so timeout 100 $K fwd --variants 4 --iters 30; done
} > $O/abl.log 2>&1
cut -c1-330 $O/abl.log | sed 's/"T": 16384, "seqs": 4, "ragged": 0, "hq": 32, "hkv": 8, "d": 128, "causal": 1, //; s/"bench": "flash_fwd", //; s/"o_max_abs.*//'
