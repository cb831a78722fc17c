cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 1200 python -m pytest tests/test_gpu_fullsize_steps.py tests/test_gpu_bench_ranks.py tests/test_gpu_stylenet.py -q -s 2>&1 | grep -E "^\[|^  [_a-z]|relative L1|passed|failed|StyleUNetLite" > gpurun_out/r03_test_numbers.txt
bash tools/profile_round.sh r03 > gpurun_out/profile_round.log 2>&1
tail -3 gpurun_out/profile_round.log
