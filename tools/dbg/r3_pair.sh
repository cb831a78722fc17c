cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py tests/test_gpu_edge_paths.py tests/test_gpu_inner_step.py tests/test_gpu_graph_step.py -q 2>&1 | tail -15
for rep in 1 2; do for pm in 0 1; do echo -n "GGS_PAIR=$pm: "; GGS_PAIR=$pm python bench.py --steps 30 --warmup 3 --views 1 --chunk 1 --no-graph --cpu-views 0 --loop-views 0 --extra-configs 0 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); k=d['roofline']['kernel_ms_per_launch']; print(d['value'], {a: round(b*1e3,1) for a,b in k.items()})"; done; done
for pm in 0 1; do echo -n "GGS_PAIR=$pm "; GGS_PAIR=$pm python tools/profile_graph_step.py 128 2>&1 | tail -1; done
