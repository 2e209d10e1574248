cd $GRAFT_REPO_ROOT
python -m pytest tests/test_factor_gpu.py -x -q -m gpu 2>&1 | tail -2
for thr in -1 2048 4096 100000000; do
  export BSP_UPD_PREFETCH_WGS=$thr
  echo "== prefetch for launches <= $thr workgroups"
  for w in "grid82 --batch 64" "grid82 --batch 8" "grid82" "flat50k" "bal-small"; do
    python bench.py --workload $w --no-extras --no-cpu-baseline --no-profile --steps 8 --warmup 2 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('%-20s %.3f ms  probe %.1e' % ('$w', d['ms_per_step'], d['residual_probe']))"
  done
done
