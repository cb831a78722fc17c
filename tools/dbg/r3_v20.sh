cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_gpu_inner_step.py tests/test_gpu_bench_ranks.py -x -q 2>&1 | tail -2
for i in 1 2 3; do python bench.py --views 20 --chunk 20 --steps 100 --warmup 5 --cpu-views 0 --loop-views 0 --extra-configs 0 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('20 views/step:', d['value'], d['ms_per_step'])"; done
python bench.py --steps 50 --cpu-views 0 --loop-views 0 --extra-configs 0 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('160 views/step:', d['value'], d['ms_per_step'])"
