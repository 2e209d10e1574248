# A/B of library variants on the headline workload with the per-kernel event times, alternating:
#   bash profiles/ab_lib_bal_k.sh reps lib1 [lib2 ...]
reps=${1:-3}; shift
for rep in $(seq $reps); do
  for lib in "" $*; do
    if [ -n "$lib" ]; then export BSP_LIB_PATH=$GRAFT_REPO_ROOT/baspacho_amd/lib$lib.so; else unset BSP_LIB_PATH; fi
    python bench.py --no-extras --no-cpu-baseline --steps 10 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); k=d.get('kernel_ms',{}); print('bal871', '${lib:-new}', d['ms_per_step'], d['residual_probe'], 'elim_factor', k.get('elim_factor'), 'elim_update', k.get('elim_update'))"
  done
done
