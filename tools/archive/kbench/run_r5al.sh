cd $GRAFT_REPO_ROOT
O=gpurun_out/r05al; mkdir -p $O
K=tools/kbench/kbench
{
for rep in 1 2; do
timeout 60 $K gemm --m 16384 --n 4096 --k 4096 --layout nn --variants 15,20,22 --iters 30
timeout 60 $K gemm --m 16384 --n 4096 --k 6144 --layout nn --variants 15,20,22 --iters 30
done
} > $O/p5.log 2>&1
grep -o '"variant": [-0-9]*\|"N": [0-9]*, "K": [0-9]*\|"us": [0-9.]*' $O/p5.log | paste - - - - - - - - -
