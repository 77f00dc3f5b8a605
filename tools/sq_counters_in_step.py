"""Per-kernel SQ counters of the GEMM / attention kernels AS THEY RUN IN THE BENCHMARK STEP, from one rocprofv3 PMC pass of bench.py itself:

    rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE \
        --kernel-trace -d /tmp/sq -o r -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-kernel-timing
    python tools/sq_counters_in_step.py <.db> <out.md>

(counters in their own pass, kernel trace only, as MI355X_MICROARCH.md prescribes).  Per kernel instantiation: dispatches, average duration, the summed counters and
the derived ratios -- matrix pipe busy = SQ_VALU_MFMA_BUSY_CYCLES / (32 x SQ_BUSY_CYCLES) (the figure of profiles/r05_step_forward_gemm_sq_counters_vs_hipblaslt.txt),
parked share = SQ_WAIT_ANY / SQ_WAVE_CYCLES, issue-stall share = SQ_WAIT_INST_ANY / SQ_WAVE_CYCLES, LDS conflict share = SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE."""
import sqlite3
import sys


def main():
    db, out = sys.argv[1], sys.argv[2]
    c = sqlite3.connect(db)
    rows = c.execute("select kernel_name, counter_name, sum(value), count(distinct dispatch_id) from counters_collection group by kernel_name, counter_name").fetchall()
    dur = dict(c.execute("select name, avg(end - start) / 1000.0 from kernels group by name").fetchall())
    per = {}
    for name, cn, v, n in rows:
        if not any(k in name for k in ("gemm_", "flash_")):
            continue
        per.setdefault(name, {"n": n})[cn] = v
    lines = ["# SQ counters of the matrix kernels in the benchmark step", "",
             "`rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE "
             "--kernel-trace -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-kernel-timing` (one pass, counters only; durations under the profiler)", "",
             "| kernel | dispatches | avg us | matrix pipe busy | parked (WAIT_ANY / WAVE_CYCLES) | issue stall (WAIT_INST_ANY / WAVE_CYCLES) | LDS conflict / active |", "|---|---:|---:|---:|---:|---:|---:|"]
    for name in sorted(per, key=lambda k: -per[k].get("SQ_BUSY_CYCLES", 0)):
        p = per[name]
        short = name.replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0]
        g = lambda k: float(p.get(k, 0.0))  # noqa: E731
        busy = g("SQ_VALU_MFMA_BUSY_CYCLES") / max(32.0 * g("SQ_BUSY_CYCLES"), 1.0)
        lines.append(f"| `{short}` | {p['n']} | {dur.get(name, 0.0):.1f} | {busy:.3f} | {g('SQ_WAIT_ANY') / max(g('SQ_WAVE_CYCLES'), 1):.3f} | "
                     f"{g('SQ_WAIT_INST_ANY') / max(g('SQ_WAVE_CYCLES'), 1):.3f} | {g('SQ_LDS_BANK_CONFLICT') / max(g('SQ_LDS_IDX_ACTIVE'), 1):.3f} |")
    open(out, "w").write("\n".join(lines) + "\n")
    print("\n".join(lines))


if __name__ == "__main__":
    main()
