# round 5: AdamW on a CU-masked stream beside the next step's forward (IE_ADAMW_CUS) -- A/B over bench.py, one box
cd $GRAFT_REPO_ROOT
O=gpurun_out/r05q; mkdir -p $O
for rep in 1 2; do
for n in 0 32 64 96; do
  IE_ADAMW_CUS=$n timeout 300 python bench.py --steps 12 --warmup 3 --no-cpu-baseline > $O/cus_${n}_$rep.json 2> $O/cus_${n}_$rep.err
  python3 -c "
import json
d=json.loads(open('$O/cus_${n}_$rep.json').read().strip().splitlines()[-1]); print('IE_ADAMW_CUS=$n', round(d['value'],1), 'tok/s', round(d['ms_per_step'],2), 'ms', 'loss', d['loss_last_step'])" 2>&1 | tail -1
done; done
