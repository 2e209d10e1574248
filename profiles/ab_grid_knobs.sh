for cfg in "BSP_X=0" "BSP_NO_LOOKAHEAD=1" "BSP_DUE_STREAM=0" "BSP_BULK_YIELD=0" "BSP_EARLY_FORK=0" "BSP_BULK_KERNEL=0"; do
  for rep in 1 2; do
    env $cfg python bench.py --workload grid82 --no-extras --no-cpu-baseline --no-profile --steps 20 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('grid82', '$cfg', d['ms_per_step'], d['residual_probe'])"
  done
done
