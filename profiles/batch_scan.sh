for b in 64 32 16 8 4 1; do
  python bench.py --workload grid82 --batch $b --no-extras --no-cpu-baseline --no-profile --steps 10 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('grid82 batch', $b, d['ms_per_step'], 'ms  per-matrix %.3f ms' % (d['ms_per_step']/$b))"
done
