cd $GRAFT_REPO_ROOT
export BSP_LIB_PATH=$GRAFT_REPO_ROOT/baspacho_amd/libbaspacho_amd_trace_upd.so
timeout 300 python tools/trace_upd.py 8 > gpurun_out/trace_upd_b8.txt 2>&1; echo "b8 rc=$?"
timeout 300 python tools/trace_upd.py 64 > gpurun_out/trace_upd_b64.txt 2>&1; echo "b64 rc=$?"
timeout 300 python tools/trace_upd.py 1 > gpurun_out/trace_upd_b1.txt 2>&1; echo "b1 rc=$?"
