cd $GRAFT_REPO_ROOT
for rep in 1 2; do
for ty in 0 1; do
  export BSP_TILE_YIELD=$ty
  python bench.py --no-extras --no-cpu-baseline --steps 10 --warmup 3 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('bal871 tile_yield=$ty %.3f ms probe %.1e' % (d['ms_per_step'], d['residual_probe']), d['kernel_ms']['update'], d['kernel_ms']['chain_update'])"
done
done
for ty in 0 1; do
  export BSP_TILE_YIELD=$ty
  python bench.py --workload flat50k --no-extras --no-cpu-baseline --no-profile --steps 5 --warmup 2 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('flat50k tile_yield=$ty %.3f ms probe %.1e' % (d['ms_per_step'], d['residual_probe']))"
done
