"""Summarise a rocprofv3 (ROCm 7.2) rocpd SQLite database into a small text table for profiles/.

    python tools/rocprof_summary.py gpurun_out/prof_r3/bench_results.db profiles/r01_bench_kernel_stats.md "<command line that was profiled>"

rocprofv3 --kernel-trace --stats writes a .db by default on this image; `top_kernels` is its per-kernel
aggregate (calls, total / average duration in ns-derived microseconds, share of GPU time).  When the database
also holds PMC samples (`--pmc ...` run) the per-kernel average of every counter is appended.
"""
import sqlite3
import sys


def short(name, n=110):
    name = name.replace("(anonymous namespace)::", "")
    return name if len(name) <= n else name[: n - 3] + "..."


def main():
    db, out, cmd = sys.argv[1], sys.argv[2], (sys.argv[3] if len(sys.argv) > 3 else "")
    c = sqlite3.connect(db)
    rows = c.execute("select name, total_calls, total_duration, average, percentage from top_kernels order by total_duration desc").fetchall()
    lines = ["# rocprofv3 kernel summary", "", f"command: `{cmd}`", f"source: `{db}` (rocprofv3 --kernel-trace --stats, ROCm 7.2, MI355X gfx950)", "",
             "| kernel | calls | total ms | avg us | % GPU time |", "|---|---:|---:|---:|---:|"]
    for name, calls, tot, avg, pct in rows[:40]:
        lines.append(f"| `{short(name)}` | {calls} | {tot / 1e3:.2f} | {avg:.1f} | {pct:.2f} |")
    total = sum(r[2] for r in rows)
    lines += ["", f"total GPU kernel time: {total / 1e3:.1f} ms over {sum(r[1] for r in rows)} launches"]
    try:
        pmc = c.execute(
            "select kernel_name, counter_name, sum(value) / count(distinct dispatch_id), count(distinct dispatch_id) from counters_collection "
            "group by kernel_name, counter_name order by kernel_name").fetchall()
    except sqlite3.Error:
        pmc = []
    if pmc:
        lines += ["", "## PMC counters (average per dispatch)", "", "| kernel | counter | avg value | dispatches |", "|---|---|---:|---:|"]
        for name, cn, v, n in pmc:
            lines.append(f"| `{short(name, 80)}` | {cn} | {v:.1f} | {n} |")
    open(out, "w").write("\n".join(lines) + "\n")
    print("\n".join(lines[:24]))


if __name__ == "__main__":
    main()
