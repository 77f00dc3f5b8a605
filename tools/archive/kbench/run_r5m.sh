cd $GRAFT_REPO_ROOT
O=gpurun_out/r05m; mkdir -p $O
export IE_REF_LIB=tools/kbench/ab/lib_r04.so
K=tools/kbench/kbench
{
timeout 100 $K bwd --variants 0,1,2,3 --iters 20
echo "== other shapes"
timeout 100 $K bwd --variants 0,2,3 --iters 3 --ragged 1 --seqs 8 --len 3000
timeout 100 $K bwd --variants 0,2 --iters 3 --d 64 --hq 32 --hkv 32
timeout 100 $K bwd --variants 0,2 --iters 3 --causal 0 --len 2048
timeout 100 $K bwd --variants 0,2,3 --iters 3 --hq 8 --hkv 8 --len 300 --ragged 1 --seqs 7
} > $O/spill.log 2>&1
cut -c1-330 $O/spill.log | sed 's/"T": 16384, "seqs": 4, "ragged": 0, "hq": 32, "hkv": 8, "d": 128, "causal": 1, //; s/"bench": "flash_bwd", //; s/"tflops_algorithmic"/"tf"/'
ls /sys/class/drm/card*/device/hwmon/hwmon*/ 2>&1 | head -30
{
unset IE_REF_LIB
bash tools/power_sample.sh idle sleep 2
bash tools/power_sample.sh fwd_v2 $K fwd --variants 2 --iters 4000
bash tools/power_sample.sh fwd_v3 $K fwd --variants 3 --iters 4000
bash tools/power_sample.sh bwd_v0 $K bwd --variants 0 --iters 1500
bash tools/power_sample.sh bwd_v3 $K bwd --variants 3 --iters 1500
bash tools/power_sample.sh gemm_fwd $K gemm --m 16384 --n 4096 --k 4096 --layout nt --variants -1 --iters 8000
bash tools/power_sample.sh gemm_wgrad $K gemm --m 4096 --n 14336 --k 16384 --layout tn --variants -1 --iters 2000
bash tools/power_sample.sh mfma_rate tools/probes/mfma_rate
} > $O/power.log 2>&1
cat $O/power.log
