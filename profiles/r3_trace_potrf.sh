cd $GRAFT_REPO_ROOT
BSP_LIB_PATH=$GRAFT_REPO_ROOT/baspacho_amd/libbaspacho_amd_trace.so timeout 300 python tools/trace_potrf.py 2>&1 | tail -8
