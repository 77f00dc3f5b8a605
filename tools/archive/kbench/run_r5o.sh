# round 5: what the 256x256-tile products lose when some CUs are not theirs (a collective's kernels hold them on a multi-GPU run): the same launches
# with the process restricted to fewer CUs (ROC_GLOBAL_CU_MASK / HSA_CU_MASK), 20 iterations each
cd $GRAFT_REPO_ROOT
O=gpurun_out/r05o; mkdir -p $O
K=tools/kbench/kbench
mask() {  # n CUs of 256 enabled, spread evenly over the 8 XCDs (32 CUs each): per-XCD mask of n/8 low bits
  python3 - "$1" <<'PY'
import sys
n=int(sys.argv[1]); per=n//8
m=0
for x in range(8): m |= ((1<<per)-1) << (32*x)
print(hex(m))
PY
}
{
for n in 256 248 240 224 192; do
  M=$(mask $n)
  for shape in "nt 16384 4096 4096" "nt 16384 4096 14336" "nn 16384 14336 4096" "tn 4096 14336 16384"; do
    set -- $shape
    echo "== cus $n $1 M=$2 N=$3 K=$4"
    ROC_GLOBAL_CU_MASK=$M HSA_CU_MASK=0:$M timeout 100 $K gemm --m $2 --n $3 --k $4 --layout $1 --variants -1 --iters 20 2>&1 | tail -1 | cut -c1-200
  done
done
} > $O/cu_mask.log 2>&1
cat $O/cu_mask.log | paste - - | awk '{print $2,$3,$4,$5,$6,$7, $0}' | sed 's/{"bench.*"us": /us /; s/, "tflops": / tf /; s/, "max_scaled.*//' | cut -c1-120
