"""Per-launch view of one kernel in a rocprofv3 (rocpd) database: duration of every launch in start order, and how much of it ran while a
second kernel (another stream) was on the chip.  Answers "does kernel X slow down in the step because Y shares the chip with it?".

    python tools/rocprof_timeline.py <bench_results.db> <substring of X> <substring of Y> [out.md]
"""
import sqlite3
import statistics
import sys


def main():
    db, xname, yname = sys.argv[1], sys.argv[2], sys.argv[3]
    out = sys.argv[4] if len(sys.argv) > 4 else None
    c = sqlite3.connect(db)
    rows = c.execute("select name, start, end from kernels order by start").fetchall()
    ys = [(s, e) for n, s, e in rows if yname in n]
    xs = [(s, e) for n, s, e in rows if xname in n]
    lines = [f"# `{xname}` launches vs `{yname}` on the chip", "", f"source: `{db}`; {len(xs)} launches of X, {len(ys)} of Y", ""]
    if not xs:
        print("no launches of", xname)
        return
    j = 0
    recs = []
    for s, e in xs:
        while j < len(ys) and ys[j][1] <= s:
            j += 1
        ov, k = 0, j
        while k < len(ys) and ys[k][0] < e:
            ov += max(0, min(e, ys[k][1]) - max(s, ys[k][0]))
            k += 1
        recs.append(((e - s) / 1e3, ov / (e - s)))
    alone = [d for d, f in recs if f < 0.05]
    shared = [d for d, f in recs if f >= 0.5]
    lines += ["| launches | count | median us | mean us | min | max |", "|---|---:|---:|---:|---:|---:|"]
    for label, v in (("all", [d for d, _ in recs]), (f"no `{yname}` on the chip (< 5 % of the launch)", alone), (f"`{yname}` on the chip for >= 50 % of the launch", shared)):
        if v:
            lines.append(f"| {label} | {len(v)} | {statistics.median(v):.1f} | {statistics.mean(v):.1f} | {min(v):.1f} | {max(v):.1f} |")
    lines += ["", "per launch (start order): duration us / share of the launch with Y on the chip", ""]
    lines.append(" ".join(f"{d:.0f}/{f:.2f}" for d, f in recs))
    text = "\n".join(lines) + "\n"
    if out:
        open(out, "w").write(text)
    print(text[:3000])


if __name__ == "__main__":
    main()
