cd $GRAFT_REPO_ROOT; O=gpurun_out/r06y_hold; mkdir -p $O
for rep in 1 2; do
  for cfg in "0 0" "0 16" "0 32" "128 0" "128 8" "128 16" "128 64"; do
    set -- $cfg
    [ $2 = 0 ] && HF="" || HF="--hold-cus $2,100"
    timeout 300 python bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-kernel-timing --adamw-cus $1 $HF > $O/a$1_h$2_$rep.json 2> $O/a$1_h$2_$rep.err
    echo "adamw-cus $1 hold $2 rep $rep: $(grep -o '"ms_per_step": [0-9.]*' $O/a$1_h$2_$rep.json)"
  done
done 2>&1 | tee $O/hold2.log
