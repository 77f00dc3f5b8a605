cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
K=tools/kbench/kbench
{
for i in 1 2; do
echo "== lib builtin-dma (r1 bwd kernels)"; timeout 200 $K bwd --variants 0 --iters 10; timeout 100 $K fwd --variants 0,2 --iters 10
echo "== lib asm-dma + explicit-register dkdv"; IE_LIB=tools/kbench/ab/lib_asm_dma.so timeout 200 $K bwd --variants 0 --iters 10; IE_LIB=tools/kbench/ab/lib_asm_dma.so timeout 100 $K fwd --variants 0,2 --iters 10
done
} > gpurun_out/k8.log 2>&1
grep -E "==|us" gpurun_out/k8.log | sed 's/"T".*"us"/"us"/' | cut -c1-120
