# round 4: where the forward product's time goes on the 16x16x32 refill schedule (variant 20): timing ablations built with -DIE_REFILL_ABL=n (results wrong):
# 8 no DMA pieces, 16 no fragment reads, 24 neither (MFMAs + barriers + waits), 31 MFMAs only
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
K=tools/kbench/kbench
{
for shape in "4096 4096" "28672 4096"; do
  set -- $shape
  echo "== fwd N=$1 K=$2, production"
  timeout 100 $K gemm --m 16384 --n $1 --k $2 --layout nt --variants 20 --iters 10
  for a in 8 16 24 31; do
    echo "== fwd N=$1 K=$2, IE_REFILL_ABL=$a"
    IE_LIB=tools/kbench/ab/lib_rf5_abl$a.so timeout 100 $K gemm --m 16384 --n $1 --k $2 --layout nt --variants 20 --iters 10
  done
done
} > gpurun_out/r4e.log 2>&1
grep -E "==|us|rror|fail" gpurun_out/r4e.log | sed 's/"bench": "gemm", //' | cut -c1-160
