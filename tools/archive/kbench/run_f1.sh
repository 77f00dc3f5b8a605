cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
L=${1:-f1}
{
timeout 300 python -m pytest tests/test_kernels_gpu.py -q -x -k "flash or attn" 2>&1 | tail -3
for v in 0 1; do timeout 60 tools/kbench/kbench bwd --variants $v --iters 10 | cut -c1-330; done
timeout 60 tools/kbench/kbench bwd --variants 0,1 --iters 10 --d 64 --hq 32 --hkv 32 2>&1 | tail -2 | cut -c1-330
for v in $(ls tools/kbench/ab/lib_*.so 2>/dev/null); do
  echo "# lib $v"
  for va in 0 1; do IE_LIB=$v timeout 60 tools/kbench/kbench bwd --variants $va --iters 2 2>&1 | tail -2 | cut -c1-330; done
done
} > gpurun_out/$L.log 2>&1
