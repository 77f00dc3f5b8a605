# round-5 yardstick: the plain forward products of the training step (wqkv, wo, w2, head: 97 per step) by this repo's kernel vs by hipBLASLt (torch.mm),
# INSIDE the step, under rocprofv3 --kernel-trace (kernel microseconds), A B A B on one box.  The fused w1 | w3 product stays this repo's in both arms.
cd $GRAFT_REPO_ROOT
TAG=${1:-r05h}
O=gpurun_out/$TAG; mkdir -p $O
export TMPDIR=/tmp
for rep in 1 2; do
  for arm in ours lib; do
    V=-1; [ $arm = lib ] && V=-100
    rm -rf /tmp/prof_ab
    timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/prof_ab -o bench -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-kernel-timing --fwd-variant $V > $O/${arm}_${rep}_line.json 2> $O/${arm}_${rep}.err
    DB=$(find /tmp/prof_ab -name "*.db" | head -1)
    python3 tools/rocprof_summary.py "$DB" $O/${arm}_${rep}_kernel_stats.md "rocprofv3 --kernel-trace --stats -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-kernel-timing --fwd-variant $V" > /dev/null
    echo "== $arm $rep"; grep -o '"ms_per_step": [0-9.]*' $O/${arm}_${rep}_line.json; grep "gemm_dma_k\|Cijk\|flash_fwd8" $O/${arm}_${rep}_kernel_stats.md | cut -c1-60,110-170
  done
done
