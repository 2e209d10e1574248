# A/B of library variants: elimination kernels' in-situ times on the headline workload
for rep in 1 2; do
for lib in "" $*; do
    if [ -n "$lib" ]; then export BSP_LIB_PATH=$GRAFT_REPO_ROOT/baspacho_amd/lib$lib.so; else unset BSP_LIB_PATH; fi
    python bench.py --no-extras --no-cpu-baseline --steps 10 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('bal871', '${lib:-base}', d['ms_per_step'], d['residual_probe'], d['kernel_ms']['elim_factor'][0], d['kernel_ms']['elim_update'][0])"
done
done
