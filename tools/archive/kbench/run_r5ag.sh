cd $GRAFT_REPO_ROOT
O=gpurun_out/r05ag; mkdir -p $O
K=tools/kbench/kbench
{
for rep in 1 2; do
for abl in 0 8 12; do
echo "== abl $abl"
for k in 2048 4096 8192; do IE_GEMM_ABLATE=$abl timeout 60 $K gemm --m 16384 --n 4096 --k $k --layout nt --variants -1 --iters 30; done
done
done
} > $O/frame.log 2>&1
grep -o '^== abl [0-9]*\|"K": [0-9]*\|"us": [0-9.]*' $O/frame.log | paste - - - - - - -
