cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
L=${1:-f1}
{
timeout 300 python -m pytest tests/test_kernels_gpu.py -q -x -k "flash or attn" 2>&1 | tail -3
for rep in 1 2; do timeout 60 tools/kbench/kbench bwd --variants 0 --iters 10; done
timeout 60 tools/kbench/kbench bwd --variants 0 --iters 10 --d 64 --hq 32 --hkv 32 2>&1 | tail -1
} > gpurun_out/$L.log 2>&1
