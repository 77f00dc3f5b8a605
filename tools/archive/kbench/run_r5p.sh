cd $GRAFT_REPO_ROOT
O=gpurun_out/r05p; mkdir -p $O
K=tools/kbench/kbench
{
for fl in 0 1; do
echo "== flush $fl"
timeout 100 $K fwd --variants 0,2,3 --iters 30 --flush $fl
timeout 100 $K bwd --variants 0,3 --iters 20 --flush $fl
timeout 100 $K gemm --m 16384 --n 4096 --k 14336 --layout nt --variants -1 --iters 30 --flush $fl
timeout 100 $K gemm --m 4096 --n 14336 --k 16384 --layout tn --variants -1 --iters 30 --flush $fl
done
} > $O/flush.log 2>&1
cut -c1-330 $O/flush.log | sed 's/"T": 16384, "seqs": 4, "ragged": 0, "hq": 32, "hkv": 8, "d": 128, "causal": 1, //; s/"o_max_abs.*//; s/"dq_max.*//; s/"max_scaled.*//'
