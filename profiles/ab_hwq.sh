# hardware queues: GPU_MAX_HW_QUEUES (HIP default 4) against the number of streams a factor uses
for w in grid82 bal871; do
  for cfg in "BSP_LOOKAHEAD_MIN_GF=0" "BSP_LOOKAHEAD_MIN_GF=0 GPU_MAX_HW_QUEUES=8" "BSP_LOOKAHEAD_MIN_GF=0 GPU_MAX_HW_QUEUES=2" "BSP_LOOKAHEAD_MIN_GF=0 BSP_DUE_STREAM=0"; do
    for rep in 1 2; do
    env $cfg python bench.py --workload $w --no-extras --no-cpu-baseline --no-profile --steps 10 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$w', '$cfg', d['ms_per_step'])"
    done
  done
done
