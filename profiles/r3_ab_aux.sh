cd $GRAFT_REPO_ROOT
for rep in 1 2 3; do
for lib in "" aux2 aux16 aux1; do
    if [ -n "$lib" ]; then export BSP_LIB_PATH=$GRAFT_REPO_ROOT/baspacho_amd/lib$lib.so; else unset BSP_LIB_PATH; fi
    python bench.py --no-extras --no-cpu-baseline --steps 10 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('bal871', '${lib:-base}', d['ms_per_step'], d['kernel_ms']['update'][0], d['kernel_ms_isolated']['update'][0], d['residual_probe'])"
done
done
