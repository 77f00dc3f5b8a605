cd $GRAFT_REPO_ROOT
O=gpurun_out/r05am; mkdir -p $O
K=tools/kbench/kbench
{
for rep in 1 2; do
timeout 60 $K gemm --m 16384 --n 4096 --k 28672 --layout nn --variants 20,22 --iters 20
timeout 60 $K gemm --m 16384 --n 4096 --k 6144 --layout nt --variants 20,22 --iters 30
timeout 60 $K gemm --m 16384 --n 4096 --k 8192 --layout nn --variants 20,22 --iters 30
timeout 60 $K gemm --m 16384 --n 6144 --k 4096 --layout nn --variants 20,22 --iters 30
done
} > $O/p5.log 2>&1
grep -o '"variant": [-0-9]*\|"layout": "[a-z]*"\|"N": [0-9]*, "K": [0-9]*\|"us": [0-9.]*' $O/p5.log | paste - - - - - - - -
