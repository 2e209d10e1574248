# factor() as a captured hipGraph (BSP_GRAPH=1) against plain launches (=0), every workload
for w in grid82 tridiag bal-small bal871 flat50k; do
  for g in 0 1 0 1; do
    BSP_GRAPH=$g python bench.py --workload $w --no-extras --no-cpu-baseline --no-profile --steps 10 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$w', 'graph=$g', d['ms_per_step'], d['residual_probe'])"
  done
done
for g in 0 1 0 1; do
  BSP_GRAPH=$g python bench.py --workload grid82 --batch 64 --no-extras --no-cpu-baseline --no-profile 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('g64', 'graph=$g', d['ms_per_step'], d['residual_probe'])"
done
