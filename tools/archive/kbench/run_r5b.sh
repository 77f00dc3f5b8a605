# round 5: widened epilogue stores (T21) -- the library under test against the round-4 library (tools/kbench/ab/lib_r04.so) as the reference
cd $GRAFT_REPO_ROOT
O=gpurun_out/r05b; mkdir -p $O
export IE_REF_LIB=tools/kbench/ab/lib_r04.so
K=tools/kbench/kbench
{
for rep in 1 2; do
  echo "== fwd r04"; IE_LIB=tools/kbench/ab/lib_r04.so timeout 100 $K fwd --variants 0,2,3 --iters 20
  echo "== fwd new"; timeout 100 $K fwd --variants 0,2,3 --iters 20
  echo "== bwd r04"; IE_LIB=tools/kbench/ab/lib_r04.so timeout 100 $K bwd --variants 0,1 --iters 20
  echo "== bwd new"; timeout 100 $K bwd --variants 0,1 --iters 20
done
echo "== correctness on other shapes (new vs r04 reference)"
timeout 100 $K fwd --variants 0,2 --iters 3 --ragged 1 --seqs 8 --len 3000
timeout 100 $K bwd --variants 0,1 --iters 3 --ragged 1 --seqs 8 --len 3000
timeout 100 $K fwd --variants 0,2 --iters 3 --d 64 --hq 32 --hkv 32
timeout 100 $K bwd --variants 0,1 --iters 3 --d 64 --hq 32 --hkv 32
timeout 100 $K fwd --variants 0,2 --iters 3 --causal 0 --len 2048
timeout 100 $K bwd --variants 0,1 --iters 3 --causal 0 --len 2048
} > $O/t21.log 2>&1
cut -c1-300 $O/t21.log | sed 's/"T": 16384, "seqs": 4, "ragged": 0, "hq": 32, "hkv": 8, "d": 128, "causal": 1, //'
