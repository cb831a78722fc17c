#!/bin/bash
# launch-set size / pipelining sweep of the two secondary workloads of bench.py (K = 16 at config 2; stress config 5)
cd "${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}"
run() { python bench.py --cpu-views 0 --loop-views 0 --extra-configs 0 --timing-only "$@" 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])"; }
for rep in 1 2; do
for c in 32 40 64; do for p in 0 1; do echo -n "K16 views 64 chunk $c pipeline $p graph: "; run --sh-degree 3 --views 64 --chunk $c --pipeline $p --steps 12 --warmup 3; done; done
for c in 8 16 32; do for p in 0 1; do echo -n "config5 views 32 chunk $c pipeline $p graph: "; run --sh-degree 3 --n-around 500 --n-rows 500 --width 3840 --height 2160 --views 32 --chunk $c --pipeline $p --steps 6 --warmup 2; done; done
done
