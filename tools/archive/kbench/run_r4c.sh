# round 4: the input-gradient product (A k-contiguous, B k-major) on the 16x16x32 refill schedule (variant 20) against variants 19 (32x32x16 refill) and 15 (8-wave ring)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
K=tools/kbench/kbench
{
for shape in "512 512 64" "512 512 128" "777 1000 192" "16384 4096 6144" "16384 4096 4096" "16384 4096 28672" "16384 14336 4096"; do
  set -- $shape
  echo "== dgrad M=$1 N=$2 K=$3"
  timeout 100 $K gemm --m $1 --n $2 --k $3 --layout nn --variants 15,19,20,19,20 --iters 20
done
} > gpurun_out/r4c.log 2>&1
grep -E "==|us|rror|fail" gpurun_out/r4c.log | sed 's/"bench": "gemm", //' | cut -c1-200
