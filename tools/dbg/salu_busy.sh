#!/bin/bash
# SALU / VALU busy of every kernel of the default bench workload (rocprofv3 derived metrics, one --pmc pass each, kernel-trace only)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for m in "SALUBusy" "VALUBusy" "SQ_INSTS_SALU SQ_INST_CYCLES_SALU SQ_ACTIVE_INST_SCA SQ_BUSY_CYCLES GRBM_GUI_ACTIVE"; do
  n=$(echo $m | cut -d' ' -f1)
  timeout 300 rocprofv3 --pmc $m --kernel-trace -d $R/gpurun_out/salu_$n -o s -- python $R/bench.py --cpu-views 0 --loop-views 0 --extra-configs 0 --steps 1 --warmup 0 --views 32 --chunk 32 > $R/gpurun_out/salu_$n.log 2>&1
  python - <<PY
import glob, sqlite3, re
f = glob.glob("$R/gpurun_out/salu_$n/**/*.db", recursive=True)
if not f: print("no db for $n"); raise SystemExit
cur = sqlite3.connect(f[0]).cursor()
rows = {}
for name, counter, n, total, mx in cur.execute("select kernel_name, counter_name, count(*), sum(value), max(value) from counters_collection group by kernel_name, counter_name"):
    m = re.match(r"(ggs_k_\w+)", name)
    if m: rows.setdefault(m.group(1), {})[counter] = (n, total, mx)
for k, c in sorted(rows.items()):
    print(k, {a: (round(b[1] / b[0], 2), round(b[2], 2)) for a, b in c.items()})
PY
  rm -rf $R/gpurun_out/salu_$n
done
