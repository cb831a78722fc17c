cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py -x -q 2>&1 | tail -3
for W in 8192 4096 2048; do
  for D in 0 3; do
    echo "== GGS_BWD_WAVES=$W sh=$D"; GGS_BWD_WAVES=$W python bench.py --steps 10 --warmup 2 --cpu-views 0 --loop-views 0 --sh-degree $D 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['roofline']['kernel_ms_per_launch'])"
  done
done
echo "== c5"; for W in 8192 2048; do GGS_BWD_WAVES=$W python bench.py --steps 5 --warmup 1 --cpu-views 0 --loop-views 0 --sh-degree 3 --n-around 500 --n-rows 500 --width 3840 --height 2160 --chunk 16 --views 32 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['roofline']['kernel_ms_per_launch'])"; done
