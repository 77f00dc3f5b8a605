cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
L=${1:-f4}
{
for v in $(ls tools/kbench/ab/lib_*.so); do
  echo "# lib $v"
  for va in 0 1; do IE_LIB=$v timeout 60 tools/kbench/kbench bwd --variants $va --iters 2 2>&1 | tail -2 | cut -c1-320; done
done
} > gpurun_out/$L.log 2>&1
