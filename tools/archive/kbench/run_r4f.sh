# round 4: DMA pieces of the 16x16x32 refill schedule one every third MFMA pair (production build) against the bursts inherited from SPREAD -4 (lib_rf5_dma0.so)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
K=tools/kbench/kbench
{
for shape in "nt 512 512 64" "nt 512 512 128" "nt 777 1000 192" "nn 520 392 320" "nt 16384 4096 4096" "nt 16384 28672 4096" "nt 16384 4096 14336" "nn 16384 4096 28672" "nn 16384 14336 4096"; do
  set -- $shape
  echo "== $1 M=$2 N=$3 K=$4: spread"
  timeout 100 $K gemm --m $2 --n $3 --k $4 --layout $1 --variants 20,20 --iters 20
  echo "== $1 M=$2 N=$3 K=$4: bursts"
  IE_LIB=tools/kbench/ab/lib_rf5_dma0.so timeout 100 $K gemm --m $2 --n $3 --k $4 --layout $1 --variants 20,20 --iters 20
done
} > gpurun_out/r4f.log 2>&1
grep -E "==|us|rror|fail" gpurun_out/r4f.log | sed 's/"bench": "gemm", //' | cut -c1-160
