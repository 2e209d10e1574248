# A/B of library variants on the headline workload, alternating: bash profiles/ab_lib_bal.sh varA ...
for rep in 1 2 3 4; do
for lib in "" $*; do
    if [ -n "$lib" ]; then export BSP_LIB_PATH=$GRAFT_REPO_ROOT/baspacho_amd/lib$lib.so; else unset BSP_LIB_PATH; fi
    python bench.py --no-extras --no-cpu-baseline --no-profile --steps 10 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('bal871', '${lib:-base}', d['ms_per_step'], d['residual_probe'])"
done
done
