cd $GRAFT_REPO_ROOT
for rep in 1 2; do
python bench.py --steps 30 --warmup 3 --cpu-views 0 --loop-views 0 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); k=d['roofline']['kernel_ms_per_launch']; print('cfg2', d['value'], 'fwd', k['render_fwd'], 'bwd', k['render_bwd'])"
python bench.py --steps 5 --warmup 1 --cpu-views 0 --loop-views 0 --sh-degree 3 --n-around 500 --n-rows 500 --width 3840 --height 2160 --chunk 16 --views 32 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); k=d['roofline']['kernel_ms_per_launch']; print('c5  ', d['value'], 'fwd', k['render_fwd'], 'bwd', k['render_bwd'])"
done
