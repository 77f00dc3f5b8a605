cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r05w
bash tools/kbench/run_pmc.sh r05w/fwd8 fwd --iters 3 --variants 4 > /dev/null 2>&1
python3 - <<'PY'
import re,collections
d=collections.defaultdict(dict)
for i in (1,2):
    for line in open(f'gpurun_out/r05w/fwd8_pmc{i}.md'):
        m=re.match(r"\| `(?:void )?(\w+)<([^>]*)>.*?` \| (\w+) \| ([\d.]+) \| (\d+) \|",line)
        if m: d[m.group(1)+'<'+m.group(2)+'>'][m.group(3)]=(float(m.group(4)),int(m.group(5)))
for k,v in d.items():
    if 'fwd8' not in k: continue
    wc=v['SQ_WAVE_CYCLES'][0]
    print(k)
    for c,(val,n) in sorted(v.items()):
        print(f'   {c:34s} {val/n:14.0f}   {val/wc*100 if c.startswith("SQ_W") or c.startswith("SQ_A") else 0:6.1f}%')
    print('   MFMA busy %.1f%%' % (v['SQ_VALU_MFMA_BUSY_CYCLES'][0]/v['SQ_VALU_MFMA_BUSY_CYCLES'][1]/ (v['SQ_BUSY_CYCLES'][0]/v['SQ_BUSY_CYCLES'][1]) /32*100))
PY
