"""Effective shader clock per kernel from a rocprofv3 --pmc GRBM_GUI_ACTIVE --kernel-trace database: GRBM_GUI_ACTIVE counts the cycles the graphics engine was
busy during a dispatch, so (counter / dispatch duration) is the clock the kernel ran at (MI355X_MICROARCH.md, DVFS give-back).  usage: kernel_clock.py <db> <out.md> <command>"""
import sqlite3
import sys


def main():
    db, out, cmd = sys.argv[1], sys.argv[2], sys.argv[3]
    c = sqlite3.connect(db)
    tables = [r[0] for r in c.execute("select name from sqlite_master where type in ('table','view')")]
    if "counters_collection" not in tables:
        raise SystemExit(f"no counters_collection view in {db}: {tables[:20]}")
    cols = [r[1] for r in c.execute("pragma table_info(counters_collection)")]
    # per dispatch: the counter (one row per dimension instance: the MAX over instances is the engine's busy-cycle count), start, end
    rows = c.execute("select kernel_name, dispatch_id, max(value), min(start), max(end) from counters_collection where counter_name = 'GRBM_GUI_ACTIVE' "
                     "group by kernel_name, dispatch_id").fetchall() if "start" in cols and "end" in cols else []
    if not rows:
        raise SystemExit(f"counters_collection has no start / end columns: {cols}")
    agg = {}
    for name, _, cyc, st, en in rows:
        a = agg.setdefault(name, [0, 0.0, 0.0])
        a[0] += 1
        a[1] += cyc
        a[2] += (en - st)
    lines = ["# effective shader clock per kernel", "", f"command: `{cmd}`", "", "| kernel | dispatches | avg us | GRBM_GUI_ACTIVE per dispatch | clock GHz |", "|---|---:|---:|---:|---:|"]
    for name, (n, cyc, ns) in sorted(agg.items(), key=lambda kv: -kv[1][2]):
        if ns <= 0:
            continue
        lines.append(f"| `{name[:100]}` | {n} | {ns / n / 1e3:.1f} | {cyc / n:.0f} | {cyc / ns:.3f} |")
    open(out, "w").write("\n".join(lines) + "\n")
    print("\n".join(lines[:24]))


if __name__ == "__main__":
    main()
