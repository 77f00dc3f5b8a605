# in-step A/B of the attention forward variants beside the persistent GEMMs (rocprofv3 --kernel-trace)
cd $GRAFT_REPO_ROOT
TAG=${1:-r05q}
O=gpurun_out/$TAG; mkdir -p $O
export TMPDIR=/tmp
for arm in 4 6 2 4 6; do
    rm -rf /tmp/prof_ab
    timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/prof_ab -o bench -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-kernel-timing --attn-fwd-variant $arm > $O/v${arm}_line.json 2> $O/v${arm}.err
    DB=$(find /tmp/prof_ab -name "*.db" | head -1)
    python3 tools/rocprof_summary.py "$DB" $O/v${arm}_kernel_stats.md "rocprofv3 --kernel-trace --stats -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-kernel-timing --attn-fwd-variant $arm" > /dev/null
    echo "== fwd variant $arm  $(grep -o '"ms_per_step": [0-9.]*' $O/v${arm}_line.json)"; grep "flash_fwd\|gemm_p5_k<false, 0>\|gemm_p5_k<false, 1>" $O/v${arm}_kernel_stats.md | cut -c1-60,110-170
done
