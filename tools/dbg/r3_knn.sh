cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_knn.py -x -q 2>&1 | tail -4
python tools/bench_next_rows.py 2>/dev/null | grep "f2"
