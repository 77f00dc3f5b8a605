cd $GRAFT_REPO_ROOT
O=gpurun_out/r05af; mkdir -p $O
K=tools/kbench/kbench
{
for rep in 1 2; do
for k in 2048 4096 8192 16384; do timeout 60 $K gemm --m 16384 --n 4096 --k $k --layout nt --variants -1 --iters 30; done
for k in 2048 4096 8192; do timeout 60 $K gemm --m 16384 --n 6144 --k $k --layout nt --variants -1 --iters 30; done
done
} > $O/ksweep.log 2>&1
grep -o '"N": [0-9]*, "K": [0-9]*\|"us": [0-9.]*\|"tflops": [0-9.]*' $O/ksweep.log | paste - - -
