cd $GRAFT_REPO_ROOT
for a in "--steps 20 --warmup 5" "--steps 200 --warmup 5" "--steps 20 --warmup 5" "--steps 20 --warmup 30" "--steps 5 --warmup 1"; do
  echo -n "$a : "
  python bench.py $a --cpu-views 0 --loop-views 0 --extra-configs 0 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])"
done
