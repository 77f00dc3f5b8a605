# round 4: the weight-gradient product (both operands k-major) on the 16x16x32 ring (variant 21) against the production ring (variant 17)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
K=tools/kbench/kbench
{
for shape in "512 512 64" "512 512 128" "777 1000 192" "520 392 320" "4096 4096 16384" "6144 4096 16384" "28672 4096 16384" "4096 14336 16384"; do
  set -- $shape
  echo "== wgrad M=$1 N=$2 K=$3"
  timeout 100 $K gemm --m $1 --n $2 --k $3 --layout tn --variants 17,21,17,21 --iters 20
done
} > gpurun_out/r4d.log 2>&1
grep -E "==|us|rror|fail" gpurun_out/r4d.log | sed 's/"bench": "gemm", //' | cut -c1-200
