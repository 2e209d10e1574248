cd $GRAFT_REPO_ROOT
for cfg in "BSP_DUE_STREAM=1" "BSP_DUE_STREAM=0" "BSP_BULK_AHEAD=0.3" "BSP_BULK_AHEAD=1.0" "BSP_BULK_AHEAD=2.0" "BSP_BULK_EXTRA_LDS=0" "BSP_DUE_STREAM=1"; do
  env $cfg python bench.py --workload flat50k --no-extras --no-cpu-baseline --steps 5 --warmup 2 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('%-28s %.3f ms  %s %s' % ('$cfg', d['ms_per_step'], {k: v[0] for k, v in d['kernel_ms'].items() if k in ('update','chain_update')}, d['plan']['num_atomic_upd_tasks']))"
done
