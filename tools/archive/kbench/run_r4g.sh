# round 4: the output tile stored non-temporally (lib_nt_store.so) against plain stores (production build), automatic dispatch (variant -1)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
K=tools/kbench/kbench
{
for shape in "nt 16384 4096 4096" "nt 16384 6144 4096" "nt 16384 28672 4096" "nt 16384 4096 14336" "nn 16384 14336 4096" "nn 16384 4096 6144" "tn 28672 4096 16384" "tn 4096 4096 16384"; do
  set -- $shape
  echo "== $1 M=$2 N=$3 K=$4: plain"
  timeout 100 $K gemm --m $2 --n $3 --k $4 --layout $1 --variants -1,-1 --iters 20
  echo "== $1 M=$2 N=$3 K=$4: nt"
  IE_LIB=tools/kbench/ab/lib_nt_store.so timeout 100 $K gemm --m $2 --n $3 --k $4 --layout $1 --variants -1,-1 --iters 20
done
} > gpurun_out/r4g.log 2>&1
grep -E "==|us|rror|fail" gpurun_out/r4g.log | sed 's/"bench": "gemm", //' | cut -c1-160
