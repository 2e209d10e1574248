cd $GRAFT_REPO_ROOT
python -m pytest tests/test_factor_gpu.py tests/test_stress_gpu.py tests/test_full_size_gpu.py -x -q -m gpu 2>&1 | tail -3
for w in "grid82 --batch 64" "grid82 --batch 8" "grid82" "bal-small" "flat50k" "tridiag"; do
  python bench.py --workload $w --no-extras --no-cpu-baseline --no-profile --steps 5 --warmup 2 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('%-20s %.3f ms  probe %.1e' % ('$w', d['ms_per_step'], d['residual_probe']))"
done
python bench.py --no-extras --no-cpu-baseline --steps 8 --warmup 2 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('bal871 %.3f ms' % d['ms_per_step'], d['solve1_ms'])"
BSP_LIB_PATH=$GRAFT_REPO_ROOT/baspacho_amd/libbaspacho_amd_trace.so timeout 300 python tools/trace_potrf.py 2>&1 | tail -7
