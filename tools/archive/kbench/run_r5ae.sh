cd $GRAFT_REPO_ROOT
O=gpurun_out/r05ae; mkdir -p $O
export IE_REF_LIB=tools/kbench/ab/lib_r04.so
K=tools/kbench/kbench
{
for rep in 1 2; do
echo "== base"; timeout 100 $K fwd --variants 4,6 --iters 30
for a in 1 49 128; do echo "== abl $a"; IE_LIB=tools/kbench/ab/lib_f8abl$a.so timeout 100 $K fwd --variants 4,6 --iters 30; done
done
echo "== ragged"; timeout 100 $K fwd --variants 4,6 --iters 30 --seqs 8 --len 3000 --ragged 1
echo "== full"; timeout 100 $K fwd --variants 4,6 --iters 30 --len 2048 --causal 0
} > $O/fwd8ring.log 2>&1
grep -o '^== .*\|"variant": [0-9]*\|"us": [0-9.]*\|"o_rms_rel": [0-9.e-]*' $O/fwd8ring.log | paste - - - - - - - | head -40
