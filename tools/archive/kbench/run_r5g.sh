cd $GRAFT_REPO_ROOT
O=gpurun_out/r05g; mkdir -p $O
K=tools/kbench/kbench
export TMPDIR=/tmp
for C in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum" "TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum" "TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum"; do
  T=$(echo $C | tr ' ' '_')
  rm -rf /tmp/pm; timeout 120 rocprofv3 --pmc $C --kernel-trace -d /tmp/pm -o r -- $K bwd --variants 0,2 --iters 2 > $O/pmc_$T.log 2>&1
  python3 tools/rocprof_summary.py "$(find /tmp/pm -name '*.db' | head -1)" $O/pmc_$T.md "rocprofv3 --pmc $C --kernel-trace -- kbench bwd --variants 0,2 --iters 2" > /dev/null 2>&1
  grep "flash_d" $O/pmc_$T.md | grep -v "| [0-9]* | [0-9.]* | [0-9.]* | [0-9.]* |$" | cut -c1-60,150-260
done
