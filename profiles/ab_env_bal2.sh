# A/B of environment settings on the headline workload, alternating:
#   bash profiles/ab_env_bal2.sh reps "VAR=1 VAR2=2" "VAR=3" ...     ("" = defaults, always included)
reps=${1:-3}; shift
for rep in $(seq $reps); do
  for envs in "" "$@"; do
    env $envs python bench.py --no-extras --no-cpu-baseline --no-profile --steps 10 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('bal871', '[${envs:-default}]', d['ms_per_step'], d['residual_probe'])"
  done
done
