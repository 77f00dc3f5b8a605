"""How much of a benchmark step the GPU spends with NO kernel running, from a rocprofv3 kernel trace.

    rocprofv3 --kernel-trace --stats -d /tmp/prof -o bench -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline
    python tools/gpu_idle_from_trace.py /tmp/prof/.../bench_results.db            # -> a markdown summary on stdout (copy it to profiles/)

Takes the window between the first and the last GEMM launch of the trace's final third (the timed steps; engine construction and warm-up in front of
it are dropped), merges the kernels' [start, end] intervals over all streams and reports the busy time, the idle time and the distribution of the gaps
between consecutive busy intervals -- the number a HIP-graph capture of the step could win at most."""
import sys


def main(db):
    import sqlite3

    rows = [(int(s), int(e), n) for n, s, e in sqlite3.connect(db).execute("select name, start, end from kernels order by start").fetchall()]
    rows.sort()
    gemm = [i for i, r in enumerate(rows) if "gemm_dma_k" in r[2]]
    if not gemm:
        raise SystemExit("no gemm_dma_k launches in the trace")
    lo = gemm[len(gemm) * 2 // 3]
    win = rows[lo : gemm[-1] + 1]
    t0, t1 = win[0][0], max(r[1] for r in win)
    busy, gaps, cur_end = 0, [], win[0][0]
    short = lambda n: n.replace("void ", "").replace("(anonymous namespace)::", "").replace("at::native::", "").split("(")[0][:48]  # noqa: E731
    by_pair, last = {}, short(win[0][2])
    for s, e, name in win:
        if s > cur_end:
            gaps.append(s - cur_end)
            k = (last, short(name))
            by_pair[k] = (by_pair.get(k, (0, 0))[0] + 1, by_pair.get(k, (0, 0))[1] + s - cur_end)
            busy_from = s
        else:
            busy_from = cur_end
        if e > cur_end:
            busy += e - busy_from
            cur_end = e
            last = short(name)
    span = t1 - t0
    idle = span - busy
    gaps.sort()
    n = len(gaps)
    pct = lambda q: gaps[min(n - 1, int(q * n))] / 1e3 if n else 0.0  # noqa: E731
    print("# GPU idle time inside the timed steps (rocprofv3 kernel trace)\n")
    print(f"window: {span / 1e6:.1f} ms, {len(win)} kernel launches; busy (union over streams) {busy / 1e6:.1f} ms, idle {idle / 1e6:.2f} ms = {100 * idle / span:.2f} % of the window\n")
    print(f"gaps between busy intervals: {n}; median {pct(0.5):.1f} us, p90 {pct(0.9):.1f} us, p99 {pct(0.99):.1f} us, max {gaps[-1] / 1e3 if n else 0:.1f} us; "
          f"sum of gaps <= 20 us: {sum(g for g in gaps if g <= 20000) / 1e6:.2f} ms, > 20 us: {sum(g for g in gaps if g > 20000) / 1e6:.2f} ms")
    tiny = [(e - s, n) for s, e, n in win if e - s < 15000]
    print(f"\nkernels shorter than 15 us in the window: {len(tiny)}, {sum(t for t, _ in tiny) / 1e6:.2f} ms in all\n")
    print("| idle between (kernel that ended last -> kernel that starts) | gaps | total ms |\n|---|---:|---:|")
    for (a, b), (k, t) in sorted(by_pair.items(), key=lambda kv: -kv[1][1])[:14]:
        print(f"| `{a}` -> `{b}` | {k} | {t / 1e6:.2f} |")


if __name__ == "__main__":
    main(sys.argv[1] if len(sys.argv) > 1 else ".")
