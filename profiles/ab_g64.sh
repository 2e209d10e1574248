# A/B of library variants (built with BSP_OUT / BSP_EXTRA_DEFS, see build.sh) on the batched C4 workload
for lib in "" $*; do
  for rep in 1 2; do
    if [ -n "$lib" ]; then export BSP_LIB_PATH=$GRAFT_REPO_ROOT/baspacho_amd/lib$lib.so; else unset BSP_LIB_PATH; fi
    python bench.py --workload grid82 --batch 64 --no-extras --no-cpu-baseline --no-profile 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('g64', '${lib:-base}', d['ms_per_step'], d['residual_probe'])"
  done
done
