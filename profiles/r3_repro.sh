cd $GRAFT_REPO_ROOT
python -m pytest tests/test_factor_gpu.py tests/test_solve_gpu.py -x -q -m gpu 2>&1 | tail -2
python bench.py --no-extras --no-cpu-baseline --steps 10 --warmup 3 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('bal871 %.3f ms' % d['ms_per_step'], d['roofline']['traffic'] is not None)"
