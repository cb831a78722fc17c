#!/usr/bin/env python
"""gpurun_out/TAG_{trace,sq,fetch,write}/**/*.db (tools/profile_all.sh) -> profiles/PREFIX_{kernel_stats.md, sq_counters.md,
hbm_traffic.json, hbm_traffic.md, bench.json}.
Usage: python tools/profile_summary.py TAG profiles/r02_sh0 [P W H sh_degree]"""
import glob
import json
import os
import re
import shutil
import sqlite3
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import hbm_summary  # noqa: E402
import rocpd_summary  # noqa: E402


def db(tag, kind):
    f = glob.glob(os.path.join(ROOT, "gpurun_out", f"{tag}_{kind}", "**", "*.db"), recursive=True)
    return f[0] if f else None


def sq_table(path, out, header):
    cur = sqlite3.connect(path).cursor()
    rows = {}
    for name, counter, n, total in cur.execute(
            "select kernel_name, counter_name, count(*), sum(value) from counters_collection group by kernel_name, counter_name"):
        m = re.match(r"(ggs_k_\w+)", name)
        if m:
            rows.setdefault(m.group(1), {})[counter] = (n, total)
    dur = {}
    for name, avg in cur.execute("select name, avg(end-start) from kernels group by name"):
        m = re.match(r"(ggs_k_\w+)", name)
        if m:
            dur[m.group(1)] = avg / 1e3
    with open(out, "w") as f:
        f.write(header)
        f.write("| kernel | waves / launch | VALU / wave | SALU / wave | LDS / wave | wave-cycles / wave (quad) | WAIT_ANY % | "
                "WAIT_INST_ANY % | ACTIVE_INST_VALU % | quad-cycles / VALU | kernel us |\n|---|---|---|---|---|---|---|---|---|---|---|\n")
        for k, c in sorted(rows.items()):
            g = lambda n: c.get(n, (1, 0.0))[1]
            launches = c["SQ_WAVES"][0]
            waves = max(g("SQ_WAVES"), 1.0)
            wc = max(g("SQ_WAVE_CYCLES"), 1.0)
            f.write(f"| {k} | {waves / launches:.0f} | {g('SQ_INSTS_VALU') / waves:.1f} | {g('SQ_INSTS_SALU') / waves:.1f} | "
                    f"{g('SQ_INSTS_LDS') / waves:.1f} | {wc / waves:.0f} | {100 * g('SQ_WAIT_ANY') / wc:.1f} | "
                    f"{100 * g('SQ_WAIT_INST_ANY') / wc:.1f} | {100 * g('SQ_ACTIVE_INST_VALU') / wc:.1f} | "
                    f"{g('SQ_ACTIVE_INST_VALU') / max(g('SQ_INSTS_VALU'), 1):.2f} | {dur.get(k, 0):.0f} |\n")


def main(tag, prefix, workload):
    g = os.path.join(ROOT, "gpurun_out")
    bid = open(os.path.join(g, f"{tag}_build_id.txt")).read().strip()
    bargs = open(os.path.join(g, f"{tag}_args.txt")).read().strip()
    t = db(tag, "trace")
    if t:
        old = sys.stdout
        with open(prefix + "_kernel_stats.md", "w") as f:
            sys.stdout = f
            rocpd_summary.main(t)
            sys.stdout = old
            f.write(f"\nSource: `rocprofv3 --kernel-trace --stats -- python bench.py --cpu-views 0 --loop-views 0 --extra-configs 0 {bargs} --steps 3 "
                    f"--warmup 1` on one MI355X, library build {bid}.\n")
    s = db(tag, "sq")
    if s:
        sq_table(s, prefix + "_sq_counters.md",
                 f"# SQ counters (rocprofv3 --pmc, one pass, kernel-trace only), build {bid}\n\n`python bench.py --cpu-views 0 "
                 f"--loop-views 0 --extra-configs 0 {bargs} --steps 1 --warmup 0 --views 32`; wave-cycle counters are in quad-cycles.\n\n")
    fe, wr = db(tag, "fetch"), db(tag, "write")
    if fe and wr:
        hbm_summary.main(fe, wr, prefix + "_hbm_traffic", int(os.environ.get("PMC_VIEWS", "32")), bid, workload)
    b = os.path.join(g, f"{tag}_bench.json")
    if os.path.exists(b) and os.path.getsize(b) > 2:
        shutil.copy(b, prefix + "_bench_profiled.json")
    print("wrote", sorted(glob.glob(prefix + "_*")))


if __name__ == "__main__":
    wl = dict(zip(("P", "W", "H", "sh_degree"), map(int, sys.argv[3:7]))) if len(sys.argv) > 6 else None
    main(sys.argv[1], sys.argv[2], wl)
