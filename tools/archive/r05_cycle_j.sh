# SQ counters of the GEMM kernels of the step with the persistent frame (one PMC pass, kernel trace only)
cd $GRAFT_REPO_ROOT
TAG=${1:-r05v}
O=gpurun_out/$TAG; mkdir -p $O
export TMPDIR=/tmp
CNT="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_ANY"
rm -rf /tmp/tr_f
timeout 400 rocprofv3 --pmc $CNT --kernel-trace -d /tmp/tr_f -o r -- python bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-kernel-timing > $O/line.json 2> $O/err.log
python3 - "$(find /tmp/tr_f -name '*.db' | head -1)" <<'PY' | tee $O/sq.txt
import sqlite3, sys
c = sqlite3.connect(sys.argv[1])
rows = c.execute("select kernel_name, counter_name, sum(value), count(distinct dispatch_id) from counters_collection group by kernel_name, counter_name").fetchall()
d = {}
for n, cn, v, k in rows:
    if "gemm_dma_k" in n or "gemm_p5_k" in n or "flash_" in n:
        d.setdefault(n[:90], {})[cn] = v / k
for n, m in sorted(d.items()):
    busy = m.get("SQ_VALU_MFMA_BUSY_CYCLES", 0) / (32 * max(m.get("SQ_BUSY_CYCLES", 1), 1))
    print("%-92s mfma busy %.3f  SQ_BUSY_CYCLES %12.0f  wait_inst_any/wave_cycles %.3f" % (n, busy, m.get("SQ_BUSY_CYCLES", 0), m.get("SQ_WAIT_INST_ANY", 0) / max(m.get("SQ_WAVE_CYCLES", 1), 1)))
PY
