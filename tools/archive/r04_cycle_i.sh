# round-4 GPU cycle I: the training step with the GEMM output tiles stored non-temporally (tools/kbench/ab/lib_nt_store.so copied over the library ON THE BOX) against plain stores, A B A B
cd $GRAFT_REPO_ROOT
O=gpurun_out/r04i; mkdir -p $O
L=internevo_amd/csrc/libinternevo_hip.so
cp $L /tmp/lib_plain.so
for r in 1 2; do
  cp /tmp/lib_plain.so $L
  timeout 200 python bench.py --steps 8 --warmup 3 --no-cpu-baseline > $O/plain_$r.json 2>> $O/err.log
  cp tools/kbench/ab/lib_nt_store.so $L
  timeout 200 python bench.py --steps 8 --warmup 3 --no-cpu-baseline > $O/nt_$r.json 2>> $O/err.log
done
cp /tmp/lib_plain.so $L
python3 - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r04i/*.json')):
    try:
        d=json.load(open(f)); print(f, round(d['value'],1), round(d['ms_per_step'],2), d['roofline']['frac'])
    except Exception as e: print(f, 'ERR', e)
PY
