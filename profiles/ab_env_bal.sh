# A/B of environment switches on the headline workload, alternating: bash profiles/ab_env_bal.sh "VAR=1" ...
for rep in 1 2 3; do
for cfg in "BSP_X=0" "$@"; do
    env $cfg python bench.py --no-extras --no-cpu-baseline --steps 10 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('bal871', '$cfg', d['ms_per_step'], d['residual_probe'], d['kernel_ms']['elim_factor'][0], d['kernel_ms']['elim_update'][0])"
done
done
