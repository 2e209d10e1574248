# A/B of environment switches on the batched C4 workload: bash profiles/ab_env_g64.sh "VAR=1" "VAR=2 OTHER=x" ...
for cfg in "$@"; do
  for rep in 1 2; do
    env $cfg python bench.py --workload grid82 --batch 64 --no-extras --no-cpu-baseline --no-profile 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('g64', '$cfg', d['ms_per_step'], d['residual_probe'])"
  done
done
