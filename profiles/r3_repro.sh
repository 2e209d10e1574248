cd $GRAFT_REPO_ROOT
timeout 600 python tools/debug_dense.py 320 576 832 1600 2>&1 | grep -v amdgpu.ids | tail -8
timeout 600 python tools/stress.py 9426 1 | tail -1
python -m pytest tests/test_factor_gpu.py -x -q -m gpu 2>&1 | tail -3
BSP_FUSE_POTRF=0 python -m pytest tests/test_factor_gpu.py -x -q -m gpu -k "last_lump" 2>&1 | tail -2
