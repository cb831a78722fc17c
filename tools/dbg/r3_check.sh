# round-3 checkpoint: GPU suite, default bench line, graph-step and per-view-loop timings
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 2400 python -m pytest tests -m gpu -x -q -s 2>&1 | tail -60 > gpurun_out/pytest_gpu.txt
timeout 900 python bench.py > gpurun_out/bench_default.json 2> gpurun_out/bench_default.err
timeout 300 python tools/profile_graph_step.py 128 > gpurun_out/graph_step.txt 2>&1
timeout 300 python tools/profile_loop.py > gpurun_out/profile_loop.txt 2>&1
tail -5 gpurun_out/pytest_gpu.txt; tail -c 600 gpurun_out/bench_default.json; tail -2 gpurun_out/graph_step.txt; head -3 gpurun_out/profile_loop.txt
