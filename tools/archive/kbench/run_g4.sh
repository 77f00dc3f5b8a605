cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
K=tools/kbench/kbench
L=${1:-g4}
{
timeout 60 $K gemm --m 1000 --n 520 --k 64 --layout nt --variants 18 --iters 2
timeout 60 $K gemm --m 1000 --n 520 --k 320 --layout nt --variants 18 --iters 2
for rep in 1 2; do
for abl in 0 3; do
echo "== IE_GEMM_ABLATE=$abl"
IE_GEMM_ABLATE=$abl timeout 100 $K gemm --m 16384 --n 4096 --k 4096 --layout nt --variants 11,18,16 --iters 10
IE_GEMM_ABLATE=$abl timeout 100 $K gemm --m 16384 --n 28672 --k 4096 --layout nt --variants 11,18,16 --iters 10
IE_GEMM_ABLATE=$abl timeout 100 $K gemm --m 16384 --n 4096 --k 14336 --layout nt --variants 11,18,16 --iters 10
done
done
} > gpurun_out/$L.log 2>&1
grep -E "==|us|rror|fail" gpurun_out/$L.log | sed 's/"bench": "gemm", //' | cut -c1-150
