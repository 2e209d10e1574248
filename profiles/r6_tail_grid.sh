# round 6, item 1b: the persistent tail on the root lump of GRID 82x82 (990 columns = 4 outer blocks)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
run() { python bench.py --workload grid82 --batch $1 --no-extras --no-cpu-baseline --no-profile --steps 200 --warmup 20 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('%.4f ms' % d['ms_per_step'], d.get('residual_probe'), (d.get('plan') or {}).get('num_tail_panels'))"; }
for rep in 1 2; do
for b in 1 8 64; do
  echo "batch $b product:            $(run $b)"
  echo "batch $b tail 3 of 4 blocks: $(BSP_TAIL_MIN_BLOCKS=4 BSP_TAIL_BLOCKS=3 run $b)"
  echo "batch $b tail 2 of 4 blocks: $(BSP_TAIL_MIN_BLOCKS=4 BSP_TAIL_BLOCKS=2 run $b)"
done
done
