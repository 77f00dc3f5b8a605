# fabric traffic of the plain forward products inside the step: this repo's kernel vs the hipBLASLt yardstick (rocprofv3 --pmc FETCH_SIZE, kernel trace only)
cd $GRAFT_REPO_ROOT
TAG=${1:-r05i}
O=gpurun_out/$TAG; mkdir -p $O
export TMPDIR=/tmp
for arm in lib ours; do
  V=-1; [ $arm = lib ] && V=-100
  rm -rf /tmp/tr_f
  timeout 400 rocprofv3 --pmc FETCH_SIZE --kernel-trace -d /tmp/tr_f -o r -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-kernel-timing --fwd-variant $V > $O/${arm}_line.json 2> $O/${arm}.err
  python3 - "$(find /tmp/tr_f -name '*.db' | head -1)" <<'PY' | tee $O/${arm}_fetch.txt
import sqlite3, sys
c = sqlite3.connect(sys.argv[1])
rows = c.execute("select kernel_name, sum(value), count(distinct dispatch_id) from counters_collection where counter_name = 'FETCH_SIZE' group by kernel_name order by 2 desc").fetchall()
for n, v, k in rows[:8]:
    print(f"{n[:90]:90s} dispatches {k:5d}  FETCH_SIZE avg {v / k:12.1f} KB  -> fabric read bytes per launch (x2 x1024) {2 * v * 1024 / k / 1e9:8.3f} GB")
PY
done
