cd $GRAFT_REPO_ROOT
O=gpurun_out/r05ad; mkdir -p $O
export IE_REF_LIB=tools/kbench/ab/lib_r04.so
K=tools/kbench/kbench
{
for rep in 1 2; do
echo "== base"; timeout 100 $K fwd --variants 4 --iters 30
echo "== defer"; IE_LIB=tools/kbench/ab/lib_f8defer.so timeout 100 $K fwd --variants 4 --iters 30
echo "== abl 49"; IE_LIB=tools/kbench/ab/lib_f8abl49.so timeout 100 $K fwd --variants 4 --iters 30
echo "== defer49"; IE_LIB=tools/kbench/ab/lib_f8defer49.so timeout 100 $K fwd --variants 4 --iters 30
done
} > $O/fwd8defer.log 2>&1
grep -o '^== .*\|"variant": [0-9]*\|"us": [0-9.]*\|"o_rms_rel": [0-9.e-]*' $O/fwd8defer.log | paste - - - - | head -40
