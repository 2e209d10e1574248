cd $GRAFT_REPO_ROOT
timeout 600 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -3
timeout 600 python bench.py --steps 3 --warmup 1 --no-extras 2>/dev/null | tail -1 | cut -c1-400
