# A/B timing of library builds kept under csrc/variants/*.so: the default bench, three runs each, interleaved
cd $GRAFT_REPO_ROOT
for rep in 1 2 3; do
  for f in gaussian-garments_amd/csrc/variants/*.so; do
    echo -n "$(basename $f) rep $rep: "
    GGS_LIB_PATH=$PWD/$f python bench.py --steps 30 --warmup 3 --cpu-views 0 --loop-views 0 --extra-configs 0 $BENCH_ARGS 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); k=d['roofline']['kernel_ms_per_launch']; print(d['value'], {a: round(b / d['roofline']['launch_views'] * 1e3, 2) for a, b in k.items()})"
  done
done
