cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_loss.py tests/test_gpu_graph_step.py tests/test_gpu_inner_step.py -x -q 2>&1 | tail -5
for i in 1 2; do python tools/profile_graph_step.py 128 2>&1 | tail -1; done
python tools/bench_next_rows.py 2>/dev/null | head -30
