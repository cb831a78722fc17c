cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_stylegan_ops.py tests/test_gpu_stylenet.py -x -q 2>&1 | tail -3
python tools/bench_stylegan_ops.py 2>/dev/null > gpurun_out/stylegan_ops.md; grep "upfirdn2d" gpurun_out/stylegan_ops.md
