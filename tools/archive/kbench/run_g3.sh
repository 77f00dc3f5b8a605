cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
K=tools/kbench/kbench
L=${1:-g3}
{
for rep in 1 2; do
for abl in 0 1 2 3 4; do
echo "== IE_GEMM_ABLATE=$abl (1 no barrier, 2 no vmcnt wait, 4 barrier after every k-step)"
IE_GEMM_ABLATE=$abl timeout 100 $K gemm --m 16384 --n 4096 --k 4096 --layout nt --variants 11 --iters 10
IE_GEMM_ABLATE=$abl timeout 100 $K gemm --m 16384 --n 28672 --k 4096 --layout nt --variants 11 --iters 10
IE_GEMM_ABLATE=$abl timeout 100 $K gemm --m 16384 --n 4096 --k 14336 --layout nt --variants 11 --iters 10
done
done
} > gpurun_out/$L.log 2>&1
grep -E "==|us|rror|fail" gpurun_out/$L.log | sed 's/"bench": "gemm", //' | cut -c1-150
