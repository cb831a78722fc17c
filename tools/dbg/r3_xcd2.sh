cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests/test_gpu_fullsize.py tests/test_gpu_parity.py tests/test_gpu_edge_paths.py tests/test_gpu_graph_step.py tests/test_gpu_inner_step.py -x -q 2>&1 | tail -3
for rep in 1 2 3; do for v in base xcd2; do echo -n "$v: "; GGS_LIB_PATH=$PWD/gaussian-garments_amd/csrc/variants/$v.so python tools/profile_graph_step.py 128 2>&1 | tail -1; done; done
for v in base xcd2; do echo -n "$v loop: "; GGS_LIB_PATH=$PWD/gaussian-garments_amd/csrc/variants/$v.so python tools/profile_loop.py 2>&1 | grep "per-view loop"; done
