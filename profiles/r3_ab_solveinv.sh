cd $GRAFT_REPO_ROOT
python -m pytest tests/test_solve_gpu.py tests/test_solve_perop_gpu.py tests/test_full_size_gpu.py tests/test_stress_gpu.py tests/test_bal_gpu.py tests/test_examples_gpu.py -x -q -m gpu 2>&1 | tail -4
for v in 0 1 0 1; do
BSP_SOLVE_INV=$v python bench.py --no-cpu-baseline --steps 3 --warmup 1 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('SOLVE_INV=$v solve1_ms', d['solve1_ms'], 'c5', d['c5']['factor_f32_ms'], d['c5']['refine_ms'], d['c5']['iterations'], d['c5']['final_rel_residual'])"
done
BSP_SOLVE_INV=1 python tools/solve_time.py 2>&1 | tail -3
