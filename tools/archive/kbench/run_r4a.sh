# round 4: where the forward product's time goes on the refill schedule (variant 19): timing ablations built with -DIE_REFILL_ABL=n (results wrong):
# 1 no barriers, 2 no landing waits, 3 neither, 7 neither and no LDS waits
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
K=tools/kbench/kbench
{
for shape in "4096 4096" "28672 4096" "4096 14336"; do
  set -- $shape
  echo "== fwd N=$1 K=$2, production"
  timeout 100 $K gemm --m 16384 --n $1 --k $2 --layout nt --variants 19 --iters 10
  for a in 1 2 3 7; do
    echo "== fwd N=$1 K=$2, IE_REFILL_ABL=$a"
    IE_LIB=tools/kbench/ab/lib_rf_abl$a.so timeout 100 $K gemm --m 16384 --n $1 --k $2 --layout nt --variants 19 --iters 10
  done
done
} > gpurun_out/r4a.log 2>&1
grep -E "==|us|rror|fail" gpurun_out/r4a.log | sed 's/"bench": "gemm", //' | cut -c1-200
