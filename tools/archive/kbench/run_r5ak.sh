cd $GRAFT_REPO_ROOT
O=gpurun_out/r05ak; mkdir -p $O
K=tools/kbench/kbench
{
for rep in 1 2; do
for abl in 0 4; do
echo "== abl $abl"
for k in 2048 4096 8192; do IE_GEMM_ABLATE=$abl timeout 60 $K gemm --m 16384 --n 4096 --k $k --layout nt --variants -1,22 --iters 30; done
IE_GEMM_ABLATE=$abl timeout 60 $K gemm --m 16384 --n 28672 --k 4096 --layout nt --variants -1,22 --iters 20
done
done
} > $O/p5.log 2>&1
grep -o '^== abl [0-9]*\|"variant": [-0-9]*\|"N": [0-9]*, "K": [0-9]*\|"us": [0-9.]*\|"max_scaled_err": [0-9.e-]*' $O/p5.log | paste - - - - - - - - - - - - - - - - - | head
