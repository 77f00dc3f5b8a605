# usage: run_pmc.sh <tag> <kbench args...>   -- two SQ counter passes over a kbench invocation; summaries -> gpurun_out/<tag>_pmc{1,2}.md
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
TAG=$1; shift
P1="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS"
P2="SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_MISC"
i=1
for P in "$P1" "$P2"; do
  rm -rf /tmp/pmc_$i
  timeout 300 rocprofv3 --pmc $P --kernel-trace -d /tmp/pmc_$i -o r -- tools/kbench/kbench "$@" > gpurun_out/${TAG}_pmc$i.log 2>&1
  DB=$(find /tmp/pmc_$i -name "*.db" | head -1)
  python3 tools/rocprof_summary.py "$DB" gpurun_out/${TAG}_pmc$i.md "rocprofv3 --pmc $P --kernel-trace -- kbench $*" > /dev/null 2>>gpurun_out/${TAG}_pmc$i.log
  i=$((i+1))
done
grep -h "flash\|gemm" gpurun_out/${TAG}_pmc1.md gpurun_out/${TAG}_pmc2.md | cut -c1-200
