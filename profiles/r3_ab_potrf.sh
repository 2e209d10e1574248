cd $GRAFT_REPO_ROOT
for rep in 1 2; do
for lib in "" _oldpotrf; do
  export BSP_LIB_PATH=$GRAFT_REPO_ROOT/baspacho_amd/libbaspacho_amd$lib.so
  python bench.py --no-extras --no-cpu-baseline --steps 10 --warmup 3 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('bal871 ${lib:-blocked} %.3f ms' % d['ms_per_step'], d['kernel_ms'])"
done
done
for lib in "" _oldpotrf; do
  export BSP_LIB_PATH=$GRAFT_REPO_ROOT/baspacho_amd/libbaspacho_amd$lib.so
  for w in "grid82 --batch 64" "grid82 --batch 8" "grid82" "flat50k"; do
    python bench.py --workload $w --no-extras --no-cpu-baseline --no-profile --steps 8 --warmup 2 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('%-20s ${lib:-blocked} %.3f ms' % ('$w', d['ms_per_step']))"
  done
done
