cd $GRAFT_REPO_ROOT
timeout 1200 python -m pytest tests/test_gpu_fullsize.py tests/test_gpu_parity.py tests/test_gpu_edge_paths.py -x -q 2>&1 | tail -3
bash tools/dbg/ab_libs.sh
