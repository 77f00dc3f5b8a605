cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
L=${1:-f2}
{
for v in $(ls tools/kbench/ab/lib_*.so); do
  for va in ${VARS:-0 1}; do
  echo "# lib $v variant $va"
  IE_LIB=$v timeout 60 tools/kbench/kbench bwd --variants $va --iters 10 2>&1 | tail -1 | cut -c1-320
  done
done
} > gpurun_out/$L.log 2>&1
