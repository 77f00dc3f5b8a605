# SQ counters of the plain forward products inside the step: this repo's kernel vs the hipBLASLt yardstick (one PMC pass per arm, kernel trace only)
cd $GRAFT_REPO_ROOT
TAG=${1:-r05j}
O=gpurun_out/$TAG; mkdir -p $O
export TMPDIR=/tmp
CNT="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_ANY"
for arm in lib ours; do
  V=-1; [ $arm = lib ] && V=-100
  rm -rf /tmp/tr_f
  timeout 400 rocprofv3 --pmc $CNT --kernel-trace -d /tmp/tr_f -o r -- python bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-kernel-timing --fwd-variant $V > $O/${arm}_line.json 2> $O/${arm}.err
  python3 - "$(find /tmp/tr_f -name '*.db' | head -1)" <<'PY' | tee $O/${arm}_sq.txt
import sqlite3, sys
c = sqlite3.connect(sys.argv[1])
rows = c.execute("select kernel_name, counter_name, sum(value), count(distinct dispatch_id) from counters_collection group by kernel_name, counter_name").fetchall()
d = {}
for n, cn, v, k in rows:
    if "gemm_dma_k" in n or "Cijk" in n:
        d.setdefault(n[:100], {})[cn] = v / k
for n, m in d.items():
    busy = m.get("SQ_VALU_MFMA_BUSY_CYCLES", 0) / (32 * max(m.get("SQ_BUSY_CYCLES", 1), 1))
    print(n)
    print("   ", {k: round(v) for k, v in sorted(m.items())}, " mfma busy %.3f  lds conflict/active %.3f  wait_inst_any/wave_cycles %.3f" % (
        busy, m.get("SQ_LDS_BANK_CONFLICT", 0) / max(m.get("SQ_LDS_IDX_ACTIVE", 1), 1), m.get("SQ_WAIT_INST_ANY", 0) / max(m.get("SQ_WAVE_CYCLES", 1), 1)))
PY
done
