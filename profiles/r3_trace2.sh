cd $GRAFT_REPO_ROOT
for v in 0 1; do
echo "=== BSP_NOW_SPLIT=$v"
BSP_NOW_SPLIT=$v BSP_LIB_PATH=baspacho_amd/libbaspacho_amd_trace_tile.so python tools/trace_extents.py 2>&1 | tail -26
done
