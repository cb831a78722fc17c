R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
for v in base xcd; do
  export GGS_LIB_PATH=$R/gaussian-garments_amd/csrc/variants/$v.so
  timeout 600 rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $R/gpurun_out/f_$v -o f -- python $R/bench.py --cpu-views 0 --loop-views 0 --extra-configs 0 --chunk 32 --steps 1 --warmup 0 --views 32 > $R/gpurun_out/f_$v.log 2>&1
  python - <<PY
import sqlite3, glob, re
db = glob.glob("$R/gpurun_out/f_$v/**/*.db", recursive=True)[0]
cur = sqlite3.connect(db).cursor()
rows = {}
for name, n, total in cur.execute("select kernel_name, count(*), sum(value) from counters_collection where counter_name='FETCH_SIZE' group by kernel_name"):
    m = re.match(r"(ggs_k_\w+)", name)
    if m: rows[m.group(1)] = (n, total)
for k in ("ggs_k_render_fwd", "ggs_k_render_bwd"):
    n, t = rows[k]
    # FETCH_SIZE arrives in KB summed over 8 XCDs? use the same convention as tools/hbm_summary.py: value is in kilobytes
    print("$v", k, "launches", n, "FETCH_SIZE per launch (raw units)", t / n)
PY
  rm -rf $R/gpurun_out/f_$v
done
