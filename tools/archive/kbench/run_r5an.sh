# the persistent frame with its per-XCD tile queues vs the plain launch, on a free chip and with CUs taken away (ROC_GLOBAL_CU_MASK)
cd $GRAFT_REPO_ROOT
O=gpurun_out/r05an; mkdir -p $O
K=tools/kbench/kbench
timeout 300 python -m pytest tests/test_kernels_gpu.py -q -m gpu -x -p no:cacheprovider -k "persistent or ffn_products" 2>&1 | tail -3
mask() {
  python3 - "$1" <<'PY'
import sys
n=int(sys.argv[1]); per=n//8
m=0
for x in range(8): m |= ((1<<per)-1) << (32*x)
print(hex(m))
PY
}
{
for n in 256 248 240 224; do
  M=$(mask $n)
  for shape in "nt 16384 4096 4096" "nt 16384 6144 4096" "nn 16384 14336 4096"; do
    set -- $shape
    echo "== cus $n $1 M=$2 N=$3 K=$4"
    ROC_GLOBAL_CU_MASK=$M HSA_CU_MASK=0:$M timeout 100 $K gemm --m $2 --n $3 --k $4 --layout $1 --variants 20,22 --iters 20 2>&1 | grep -o '"variant": [0-9]*\|"us": [0-9.]*\|"max_scaled_err": [0-9.e-]*' | paste - - - - - -
  done
done
} > $O/cu_mask.log 2>&1
cat $O/cu_mask.log | paste - - | cut -c1-200
