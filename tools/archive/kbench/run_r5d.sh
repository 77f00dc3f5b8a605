cd $GRAFT_REPO_ROOT
O=gpurun_out/r05d; mkdir -p $O
export IE_REF_LIB=tools/kbench/ab/lib_r04.so
K=tools/kbench/kbench
{
echo "== phase stamps: variant 0 then 2"
IE_LIB=tools/kbench/ab/lib_sp_timing.so timeout 100 $K bwd --variants 0 --iters 1
IE_LIB=tools/kbench/ab/lib_sp_timing.so timeout 100 $K bwd --variants 2 --iters 1
for L in sp_nt sp_sc; do echo "== $L"; IE_LIB=tools/kbench/ab/lib_$L.so timeout 100 $K bwd --variants 0,2 --iters 20; done
} > $O/spill2.log 2>&1
cut -c1-330 $O/spill2.log | grep -v '"variant": 0, .*"us": 18' | sed 's/"T": 16384, "seqs": 4, "ragged": 0, "hq": 32, "hkv": 8, "d": 128, "causal": 1, //'
