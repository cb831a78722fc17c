# L2 hit rate and fabric fetch of the two render kernels with and without the XCD-aware work order (variants base / xcd)
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
for v in base xcd; do
  export GGS_LIB_PATH=$R/gaussian-garments_amd/csrc/variants/$v.so
  timeout 600 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum --kernel-trace -d $R/gpurun_out/l2_$v -o f -- python $R/bench.py --cpu-views 0 --loop-views 0 --extra-configs 0 --chunk 32 --steps 1 --warmup 0 --views 32 > $R/gpurun_out/l2_$v.log 2>&1
  python - <<PY
import sqlite3, glob, re
db = glob.glob("$R/gpurun_out/l2_$v/**/*.db", recursive=True)
if not db: print("$v: no db"); raise SystemExit
cur = sqlite3.connect(db[0]).cursor()
rows = {}
for name, counter, n, total in cur.execute("select kernel_name, counter_name, count(*), sum(value) from counters_collection group by kernel_name, counter_name"):
    m = re.match(r"(ggs_k_\w+)", name)
    if m: rows.setdefault(m.group(1), {})[counter] = total / n
for k in ("ggs_k_render_fwd", "ggs_k_render_bwd"):
    c = rows.get(k, {})
    h, m_ = c.get("TCC_HIT_sum", 0), c.get("TCC_MISS_sum", 0)
    print("$v", k, "L2 hits / launch %.3e  misses %.3e  hit rate %.1f %%" % (h, m_, 100 * h / max(h + m_, 1)))
PY
  rm -rf $R/gpurun_out/l2_$v
done
