cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
K=tools/kbench/kbench
L=${1:-g8}
{
echo "== edge shapes"
for k in 64 128 192 320 640; do
  timeout 60 $K gemm --m 1000 --n 520 --k $k --layout nt --variants 20 --iters 2
  timeout 60 $K gemm --m 1000 --n 520 --k $k --layout nn --variants 20 --iters 2
  timeout 60 $K gemm --m 1000 --n 520 --k $k --layout tn --variants 20 --iters 2
done
for rep in 1 2; do
echo "== 7B shapes, 16384 tokens (rep $rep)"
for shape in "6144 4096" "4096 4096" "28672 4096" "4096 14336"; do
  set -- $shape
  timeout 100 $K gemm --m 16384 --n $1 --k $2 --layout nt --variants 19,20 --iters 10
  timeout 100 $K gemm --m 16384 --n $2 --k $1 --layout nn --variants 19,20 --iters 10
done
done
} > gpurun_out/$L.log 2>&1
grep -E "==|us|rror|fail" gpurun_out/$L.log | sed 's/"bench": "gemm", //' | cut -c1-170
