#!/bin/bash
# A/B of the forward preprocess with / without the SH rows staged in LDS (GGS_PRE_SH_LDS): per-kernel times at K = 16 and config 5
cd "${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}"
for rep in 1 2; do for v in 0 1; do
  for cfg in "--sh-degree 3 --chunk 32 --views 64" "--sh-degree 3 --n-around 500 --n-rows 500 --width 3840 --height 2160 --chunk 16 --views 32"; do
    echo -n "GGS_PRE_SH_LDS=$v [$cfg] rep $rep: "
    GGS_PRE_SH_LDS=$v timeout 600 python bench.py --steps 6 --warmup 2 --cpu-views 0 --loop-views 0 --extra-configs 0 --pipeline 0 $cfg 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); k=d['roofline']['kernel_ms_per_launch']; print(d['value'], 'whole', d['roofline']['whole_path']['frac'], {a: round(b / d['roofline']['launch_views'] * 1e3, 2) for a, b in k.items() if a in ('preprocess','preprocess_bwd','scatter')})"
  done; done; done
