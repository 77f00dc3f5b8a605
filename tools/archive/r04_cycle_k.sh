# round-4 GPU cycle K: the training step with the long input-gradient products on variant 20 (production) against variant 19 (tools/kbench/ab/lib_dgrad19.so copied over the library ON THE BOX), A B A B
cd $GRAFT_REPO_ROOT
O=gpurun_out/r04k; mkdir -p $O
L=internevo_amd/csrc/libinternevo_hip.so
cp $L /tmp/lib_prod.so
for r in 1 2; do
  cp /tmp/lib_prod.so $L
  timeout 200 python bench.py --steps 8 --warmup 3 --no-cpu-baseline > $O/dgrad20_$r.json 2>> $O/err.log
  cp tools/kbench/ab/lib_dgrad19.so $L
  timeout 200 python bench.py --steps 8 --warmup 3 --no-cpu-baseline > $O/dgrad19_$r.json 2>> $O/err.log
done
cp /tmp/lib_prod.so $L
python3 - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r04k/*.json')):
    try:
        d=json.load(open(f)); print(f, round(d['value'],1), round(d['ms_per_step'],2), d['roofline']['frac'])
    except Exception as e: print(f, 'ERR', e)
PY
