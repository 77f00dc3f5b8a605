# in-step A/B of the persistent GEMM frame (ie_tune_gemm_persistent) under rocprofv3 --kernel-trace, A B A B on one box
cd $GRAFT_REPO_ROOT
TAG=${1:-r05k}
O=gpurun_out/$TAG; mkdir -p $O
export TMPDIR=/tmp
for rep in 1 2; do
  for arm in 0 1; do
    rm -rf /tmp/prof_ab
    timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/prof_ab -o bench -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-kernel-timing --gemm-persistent $arm > $O/p${arm}_${rep}_line.json 2> $O/p${arm}_${rep}.err
    DB=$(find /tmp/prof_ab -name "*.db" | head -1)
    python3 tools/rocprof_summary.py "$DB" $O/p${arm}_${rep}_kernel_stats.md "rocprofv3 --kernel-trace --stats -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-kernel-timing --gemm-persistent $arm" > /dev/null
    echo "== persistent $arm rep $rep"; grep -o '"ms_per_step": [0-9.]*\|"loss_last_step": [0-9.]*' $O/p${arm}_${rep}_line.json | tr '\n' ' '; echo; grep "gemm_dma_k\|gemm_p5_k\|flash_fwd8" $O/p${arm}_${rep}_kernel_stats.md | cut -c1-70,110-170
  done
done
