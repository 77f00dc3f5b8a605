cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
K=tools/kbench/kbench
{
timeout 250 $K fwd --variants 2,48 --iters 10
} > gpurun_out/k6.log 2>&1
cut -c1-260 gpurun_out/k6.log
