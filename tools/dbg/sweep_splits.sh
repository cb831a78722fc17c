cd $GRAFT_REPO_ROOT
for D in 0 3; do
  for S in 2 3 4 5 6 8 10; do
    W=$((1564 * S))
    echo -n "sh=$D splits=$S: "; GGS_BWD_WAVES=$W python bench.py --steps 10 --warmup 2 --cpu-views 0 --loop-views 0 --extra-configs 0 --sh-degree $D 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); k=d['roofline']['kernel_ms_per_launch']; print(d['value'], round(k['preprocess_bwd']/d['roofline']['launch_views']*1e3,2))"
  done
done
