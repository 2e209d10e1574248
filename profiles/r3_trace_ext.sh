cd $GRAFT_REPO_ROOT
BSP_LIB_PATH=baspacho_amd/libbaspacho_amd_trace_tile.so timeout 300 python tools/trace_extents.py 2>&1 | tail -24
