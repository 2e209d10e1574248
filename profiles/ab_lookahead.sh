# lookahead schedule (side streams) on / off, per workload
for w in grid82 bal-small flat50k bal871; do
  for cfg in "BSP_X=0" "BSP_NO_LOOKAHEAD=1"; do
    for rep in 1 2; do
    env $cfg python bench.py --workload $w --no-extras --no-cpu-baseline --no-profile --steps 10 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$w', '$cfg', d['ms_per_step'], d['plan'])"
    done
  done
done
for cfg in "BSP_X=0" "BSP_NO_LOOKAHEAD=1"; do
  env $cfg python bench.py --workload grid82 --batch 64 --no-extras --no-cpu-baseline --no-profile 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('g64', '$cfg', d['ms_per_step'])"
  env $cfg python bench.py --workload bal871 --no-cpu-baseline --no-profile 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('c5', '$cfg', d['c5'])"
done
