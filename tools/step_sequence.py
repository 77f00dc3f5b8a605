"""The kernels of the LAST benchmark step of a rocprofv3 kernel trace, in start order: short name, duration, the gap to the previous end on the same queue,
the queue -- to see what sits between the products (fills, copies, reductions) and what it costs in stream time.

    python tools/step_sequence.py <bench_results.db> [out.txt]

Runs of the same (name, queue) are folded ("x n", durations averaged).  The step = from the last embedding_fwd_k launch to the end of the trace."""
import sqlite3
import sys


def main():
    db = sys.argv[1]
    out = open(sys.argv[2], "w") if len(sys.argv) > 2 else sys.stdout
    c = sqlite3.connect(db)
    try:
        rows = c.execute("select name, start, end, queue_id from kernels order by start").fetchall()
    except sqlite3.OperationalError:
        rows = [(n, s, e, 0) for n, s, e in c.execute("select name, start, end from kernels order by start").fetchall()]
    emb = [i for i, r in enumerate(rows) if "embedding_fwd_k" in r[0]]
    if not emb:
        raise SystemExit("no embedding_fwd_k launch in the trace")
    lo = max(0, emb[-1] - 120)     # (the optimizer's kernels of the step before sit in front of the embedding)
    win = rows[lo:]
    short = lambda n: n.replace("void ", "").replace("(anonymous namespace)::", "").replace("at::native::", "").split("(")[0][:70]  # noqa: E731
    last_end, folded = {}, []
    for n, s, e, q in win:
        gap = (s - last_end[q]) / 1e3 if q in last_end else 0.0
        last_end[q] = e
        k = (short(n), q)
        if folded and folded[-1][0] == k:
            folded[-1][1] += 1; folded[-1][2] += (e - s) / 1e3; folded[-1][3] += gap
        else:
            folded.append([k, 1, (e - s) / 1e3, gap])
    print(f"# {len(win)} launches from {win[0][1]} (ns); columns: queue | kernel | launches | avg us | avg gap to the previous end on this queue, us", file=out)
    for (n, q), k, d, g in folded:
        print(f"{q} | {n} | {k} | {d / k:.1f} | {g / k:.1f}", file=out)


if __name__ == "__main__":
    main()
