# A/B of one environment setting against the defaults over the other workloads, alternating:
#   bash profiles/ab_env_multi.sh reps "VAR=1 VAR2=2"
reps=${1:-2}; shift
for wl in "flat50k:--workload flat50k --steps 5" "grid82:--workload grid82 --steps 50" "grid82x64:--workload grid82 --batch 64 --steps 10" "bal-small:--workload bal-small --steps 20" "tridiag:--workload tridiag --steps 50"; do
  name=${wl%%:*}; args=${wl#*:}
  for rep in $(seq $reps); do
    for envs in "" "$@"; do
      env $envs python bench.py $args --no-extras --no-cpu-baseline --no-profile 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$name', '[${envs:-default}]', d['ms_per_step'], d['residual_probe'])"
    done
  done
done
