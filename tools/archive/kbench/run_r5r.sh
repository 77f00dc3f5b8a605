# round 5: the five-product attention backward IN THE STEP (IE_ATTN_BWD_SPILL=1), A B A B over bench.py on one box
cd $GRAFT_REPO_ROOT
O=gpurun_out/r05r; mkdir -p $O
for rep in 1 2; do
for n in 0 1; do
  IE_ATTN_BWD_SPILL=$n timeout 300 python bench.py --steps 12 --warmup 3 --no-cpu-baseline > $O/spill_${n}_$rep.json 2> $O/spill_${n}_$rep.err
  python3 -c "
import json
d=json.loads(open('$O/spill_${n}_$rep.json').read().strip().splitlines()[-1]); print('IE_ATTN_BWD_SPILL=$n run $rep:', round(d['value'],1), 'tokens/s', round(d['ms_per_step'],2), 'ms per step, loss', d['loss_last_step'], 'gn', d['grad_norm_last_step'])" 2>&1 | tail -1
done; done
