cd $GRAFT_REPO_ROOT
O=gpurun_out/r05ac; mkdir -p $O
export IE_REF_LIB=tools/kbench/ab/lib_r04.so
K=tools/kbench/kbench
{
for rep in 1 2; do
echo "== base"; timeout 100 $K fwd --variants 4,6 --iters 30
for a in 1 17 33 49 16 32 48 128 129; do echo "== abl $a"; IE_LIB=tools/kbench/ab/lib_f8abl$a.so timeout 100 $K fwd --variants 4,6 --iters 30; done
done
} > $O/fwd8abl.log 2>&1
grep -o '^== .*\|"variant": [0-9]*\|"us": [0-9.]*' $O/fwd8abl.log | paste - - - - - | head -40
