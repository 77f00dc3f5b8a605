cd $GRAFT_REPO_ROOT
O=gpurun_out/r05j; mkdir -p $O
K=tools/kbench/kbench
export TMPDIR=/tmp
for C in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum"; do
  T=$(echo $C | tr ' ' '_')
  rm -rf /tmp/pm; timeout 120 rocprofv3 --pmc $C --kernel-trace -d /tmp/pm -o r -- $K bwd --variants 0,2,3 --iters 2 > $O/pmc_$T.log 2>&1
  python3 tools/rocprof_summary.py "$(find /tmp/pm -name '*.db' | head -1)" $O/pmc_$T.md "rocprofv3 --pmc $C --kernel-trace -- kbench bwd --variants 0,2,3 --iters 2" > /dev/null 2>&1
done
python3 - <<'PY'
import re,glob
for f in sorted(glob.glob('gpurun_out/r05j/pmc_*.md')):
    for line in open(f):
        m=re.match(r"\| `(?:void )?(\w+)<([^>]*)>.*?` \| (\w+) \| ([\d.e+]+) \| (\d+) \|",line)
        if m and m.group(1).startswith('flash_d') and 'delta' not in m.group(1):
            print(f"{m.group(1)+'<'+m.group(2)+'>':40s} {m.group(3):24s} {float(m.group(4))/int(m.group(5)):16.0f} per launch")
PY
