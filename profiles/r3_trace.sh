cd $GRAFT_REPO_ROOT
BSP_LIB_PATH=baspacho_amd/libbaspacho_amd_trace.so python tools/trace_potrf.py 2>&1 | tail -8 | tee gpurun_out/r3_trace_potrf.txt
BSP_LIB_PATH=baspacho_amd/libbaspacho_amd_trace_tile.so python tools/trace_extents.py 2>&1 | tail -30 | tee gpurun_out/r3_trace_extents.txt
BSP_LIB_PATH=baspacho_amd/libbaspacho_amd_trace_tile.so python tools/trace_tiles.py 2>&1 | tail -12 | tee gpurun_out/r3_trace_tiles.txt
