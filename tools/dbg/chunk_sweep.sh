cd $GRAFT_REPO_ROOT
for c in 16 32 40 80 160; do
python bench.py --steps 20 --warmup 3 --cpu-views 0 --loop-views 0 --extra-configs 0 --chunk $c 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); k=d['roofline']['kernel_ms_per_launch']; print('chunk', $c, d['value'], 'fwd/view', round(k['render_fwd']/$c*1e3,2), 'bwd/view', round(k['render_bwd']/$c*1e3,2))"
done
