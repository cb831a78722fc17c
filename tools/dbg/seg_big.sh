cd $GRAFT_REPO_ROOT
for L in 0 1024 512 256; do
  echo "== c5 GGS_SEG_BIG_LEN=$L"; GGS_SEG_BIG_LEN=$L python bench.py --steps 5 --warmup 1 --cpu-views 0 --loop-views 0 --sh-degree 3 --n-around 500 --n-rows 500 --width 3840 --height 2160 --chunk 16 --views 32 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['roofline']['kernel_ms_per_launch'])"
done
for L in 0 512 256; do
  echo "== config2 GGS_SEG_BIG_LEN=$L"; GGS_SEG_BIG_LEN=$L python bench.py --steps 10 --warmup 2 --cpu-views 0 --loop-views 0 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['roofline']['kernel_ms_per_launch'])"
done
