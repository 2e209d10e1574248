cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for v in "" _ut1 _ut2 _ut3; do
  export BSP_LIB_PATH=$GRAFT_REPO_ROOT/baspacho_amd/libbaspacho_amd$v.so
  rm -rf gpurun_out/gv_stats
  timeout 300 rocprofv3 --kernel-trace -d gpurun_out/gv_stats -o s -- python bench.py --workload grid82 --batch 64 --no-extras --no-cpu-baseline --no-profile --steps 1 --warmup 1 > gpurun_out/gv$v.log 2>&1
  echo "== variant '$v'"; python profiles/kdispatch.py gpurun_out/gv_stats/s_results.db updateTileId 22 | awk '{print $2, $3, $5}' | tr '\n' ';'; echo
done > gpurun_out/g64_dispatch_var.txt
rm -rf gpurun_out/gv_stats
