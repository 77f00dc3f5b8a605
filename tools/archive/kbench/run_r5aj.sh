cd $GRAFT_REPO_ROOT
O=gpurun_out/r05aj; mkdir -p $O
K=tools/kbench/kbench
{
for rep in 1 2; do
for k in 2048 4096 8192; do timeout 60 $K gemm --m 16384 --n 4096 --k $k --layout nt --variants -1,22 --iters 30; done
timeout 60 $K gemm --m 16384 --n 6144 --k 4096 --layout nt --variants -1,22 --iters 30
timeout 60 $K gemm --m 16384 --n 4096 --k 14336 --layout nt --variants -1,22 --iters 30
timeout 60 $K gemm --m 16384 --n 14336 --k 4096 --layout nn --variants -1,22 --iters 30
timeout 60 $K gemm --m 16384 --n 28672 --k 4096 --layout nt --variants -1,22 --iters 20
done
} > $O/p5.log 2>&1
grep -o '"variant": [-0-9]*\|"layout": "[a-z]*"\|"N": [0-9]*, "K": [0-9]*\|"us": [0-9.]*\|"max_scaled_err": [0-9.e-]*' $O/p5.log | paste - - - - - - - - - - | head -30
