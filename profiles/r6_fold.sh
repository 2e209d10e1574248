# round 6: potrf folded into the trsm launch of small multi-panel levels (trsmPanelPotrf)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_factor_gpu.py tests/test_stress_gpu.py tests/test_tail_gpu.py tests/test_solve_gpu.py -q -m gpu -x 2>&1 | tail -4
for rep in 1 2; do
python tools/ab_suite.py --reps=15 "--filter=^1|^2|^3|^4|grid82" "BSP_POTRF_IN_TRSM=0" - 2>&1 | grep -v "Warning\|amdgpu.ids"
done
run() { python bench.py --workload grid82 --batch $1 --no-extras --no-cpu-baseline --no-profile --steps 20 --warmup 3 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('%.4f ms' % d['ms_per_step'], d.get('residual_probe'))"; }
for rep in 1 2; do for b in 8 64; do
  echo "batch $b folded:   $(run $b)"
  echo "batch $b unfolded: $(BSP_POTRF_IN_TRSM=0 run $b)"
done; done
