# round 4: the refill schedule on v_mfma_f32_16x16x32_bf16 (variant 20) against the production forward kernel (variant 19): correctness on small / ragged
# shapes (k-tiles 1, 2, 3; ragged M and N), then the three forward shapes of the benchmark layer
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
K=tools/kbench/kbench
{
for shape in "512 512 64" "512 512 128" "777 1000 192" "16384 4096 4096" "16384 28672 4096" "16384 4096 14336" "16384 6144 4096"; do
  set -- $shape
  echo "== fwd M=$1 N=$2 K=$3"
  timeout 100 $K gemm --m $1 --n $2 --k $3 --layout nt --variants 19,20,19,20 --iters 20
done
} > gpurun_out/r4b.log 2>&1
grep -E "==|us|rror|fail" gpurun_out/r4b.log | sed 's/"bench": "gemm", //' | cut -c1-200
