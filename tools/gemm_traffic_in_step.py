"""HBM-side (fabric) traffic of the GEMM kernels AS THEY RUN IN THE BENCHMARK STEP, from two rocprofv3 PMC passes of bench.py itself:

    rocprofv3 --pmc FETCH_SIZE --kernel-trace -d /tmp/tf -o r -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-kernel-timing
    rocprofv3 --pmc WRITE_SIZE --kernel-trace -d /tmp/tw -o r -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-kernel-timing
    python tools/gemm_traffic_in_step.py <fetch .db> <write .db> profiles/r04_gemm_hbm_traffic.json

(separate passes, kernel trace only, as MI355X_MICROARCH.md prescribes).  Per gemm_dma_k instantiation: dispatches, average FETCH_SIZE / WRITE_SIZE in KB,
corrected bytes (fetch bytes = 2 x FETCH_SIZE x 1024 for wide coalesced streaming reads on gfx950 -- 16 B per lane, `buffer_load ... lds`; WRITE_SIZE x 1024
as is), and over ALL GEMM launches of the run the traffic per launch -- the number bench.py reports as `roofline.traffic` (against the algorithmic bytes per
launch it computes itself)."""
import json
import sqlite3
import sys


def per_kernel(db, counter):
    c = sqlite3.connect(db)
    rows = c.execute("select kernel_name, sum(value), count(distinct dispatch_id) from counters_collection where counter_name = ? group by kernel_name", (counter,)).fetchall()
    return {n: (v, k) for n, v, k in rows if "gemm_dma_k" in n or "gemm_bf16_k" in n or "gemm_p5_k" in n or "ksplit_fixup_k" in n}


def main():
    fdb, wdb, out = sys.argv[1], sys.argv[2], sys.argv[3]
    f, w = per_kernel(fdb, "FETCH_SIZE"), per_kernel(wdb, "WRITE_SIZE")
    kernels, extra, tot_bytes, tot_launch = {}, {}, 0.0, 0
    for name in sorted(f):
        fv, fk = f[name]
        wv, wk = w.get(name, (0.0, fk))
        fetch_b, write_b = 2.0 * fv * 1024 / fk, wv * 1024 / max(wk, 1)
        short = name.replace("(anonymous namespace)::", "").split("(")[0]
        if "ksplit_fixup_k" in name:   # part of a k-split product, not a product kernel of its own: its bytes count, its launches do not (bench.py checks `kernels` by name)
            extra[short] = {"dispatches": fk, "traffic_bytes_per_launch": round(fetch_b + write_b)}
            tot_bytes += (fetch_b + write_b) * fk
            continue
        kernels[short] = {"dispatches": fk, "FETCH_SIZE_KB_avg": round(fv / fk, 1), "WRITE_SIZE_KB_avg": round(wv / max(wk, 1), 1),
                          "traffic_bytes_per_launch": round(fetch_b + write_b)}
        tot_bytes += (fetch_b + write_b) * fk
        tot_launch += fk
    res = {"source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes, --kernel-trace only) -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline "
                     "--no-kernel-timing: every GEMM launch of three benchmark steps (InternLM2-7B, merged 16 384-row pass), the kernels the step really runs "
                     "(incl. the w1|w3 forward product with the SwiGLU gate in its epilogue, EPI = 1)",
           "correction": "MI355X_MICROARCH.md section HBM: fetch bytes = 2 x FETCH_SIZE x 1024 (wide coalesced streaming reads report 1/2 on gfx950), WRITE_SIZE x 1024 as "
                         "is; FETCH counts fabric-side L2 misses, Infinity-Cache hits included",
           "kernels": kernels, "fixups_of_k_split_products": extra, "gemm_launches": tot_launch, "traffic_bytes_per_launch": round(tot_bytes / max(tot_launch, 1)),
           # (a PRODUCT can be more than one kernel launch since round 6 -- the weight gradients' tail k-split runs the whole rounds and the two half-k remainders as
           # two launches: bench.py divides the run's total by ITS products, three steps' worth of them)
           "steps_profiled": 3, "traffic_bytes_total": round(tot_bytes)}
    json.dump(res, open(out, "w"), indent=1)
    print(json.dumps(res, indent=1)[:3000])


if __name__ == "__main__":
    main()
