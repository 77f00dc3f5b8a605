cd $GRAFT_REPO_ROOT
O=gpurun_out/r05u; mkdir -p $O
K=tools/kbench/kbench
{
for rep in 1 2; do
echo "== base"; timeout 100 $K fwd --variants 0,2 --iters 30
for L in fwd_prio1 fwd_prio3; do echo "== $L"; IE_LIB=tools/kbench/ab/lib_$L.so timeout 100 $K fwd --variants 0 --iters 30; done
done
} > $O/prio.log 2>&1
cut -c1-330 $O/prio.log | sed 's/"T": 16384, "seqs": 4, "ragged": 0, "hq": 32, "hkv": 8, "d": 128, "causal": 1, //; s/"o_max_abs.*//'
