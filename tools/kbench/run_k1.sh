cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
K=tools/kbench/kbench
{
timeout 200 $K bwd --variants 0,1 --iters 10
timeout 100 $K fwd --variants 0,2 --iters 10
} > gpurun_out/k7.log 2>&1
cut -c1-330 gpurun_out/k7.log
timeout 400 python -m pytest tests/test_kernels_gpu.py -x -q -m gpu -k "flash" 2>&1 | tail -5
