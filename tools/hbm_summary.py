#!/usr/bin/env python
"""Turn the two rocprofv3 PMC passes of tools/hbm_traffic.sh (FETCH_SIZE, WRITE_SIZE; rocpd .db) into
profiles/rNN_hbm_traffic.json / .md: HBM bytes per launch and per view for every ggs_k_* kernel.
Usage: python tools/hbm_summary.py gpurun_out/hbm_X_fetch/f_results.db gpurun_out/hbm_X_write/w_results.db profiles/r01_hbm_traffic [views]
FETCH_SIZE / WRITE_SIZE are reported in KB; FETCH_SIZE on gfx950 counts 64 B per 128-B request for wide coalesced
reads (MI355X_MICROARCH.md, HBM / rocprofv3 section) -> raw and x2 are both kept, traffic = fetch_x2 + write."""
import json
import re
import sqlite3
import sys


def per_kernel(path, counter):
    cur = sqlite3.connect(path).cursor()
    out = {}
    for name, n, avg in cur.execute(
            "select kernel_name, count(*), avg(value) from counters_collection where counter_name = ? "
            "group by kernel_name", (counter,)):
        m = re.match(r"(ggs_k_\w+)", name)
        if m:
            out[m.group(1)] = (n, avg * 1024.0)
    return out


def main(fetch_db, write_db, out_prefix, views=32, build_id="", workload=None):
    f, w = per_kernel(fetch_db, "FETCH_SIZE"), per_kernel(write_db, "WRITE_SIZE")
    src = ("rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) --kernel-trace -- python bench.py "
           f"--steps 1 --warmup 0 --cpu-views 0 --loop-views 0 --views {views}")
    note = (f"bytes per LAUNCH ({views} views), averaged over the dispatches of each kernel. fetch_x2 applies the gfx950 "
            "correction of MI355X_MICROARCH.md (FETCH_SIZE counts 64 B per 128-B request for 16 B/lane reads); "
            "traffic = fetch_x2 + write.")
    kern = {}
    for k in sorted(set(f) | set(w)):
        fr, wr = f.get(k, (0, 0.0))[1], w.get(k, (0, 0.0))[1]
        kern[k] = {"fetch_raw": fr, "fetch_x2": 2 * fr, "write": wr, "traffic": 2 * fr + wr,
                   "traffic_per_view": (2 * fr + wr) / views, "dispatches": f.get(k, w.get(k))[0]}
    json.dump({"source": src, "views_per_launch": views, "build_id": build_id,
               "workload": workload or {"P": 100000, "W": 1920, "H": 1080, "sh_degree": 0}, "note": note, "kernels": kern},
              open(out_prefix + ".json", "w"), indent=1)
    with open(out_prefix + ".md", "w") as md:
        md.write(f"# HBM traffic per kernel (rocprofv3 PMC), build {build_id}\n\n{src}\n\n{note}\n\n")
        md.write("| kernel | FETCH_SIZE raw MB | x2 MB | WRITE_SIZE MB | traffic MB / launch | MB / view |\n|---|---|---|---|---|---|\n")
        for k, v in kern.items():
            md.write(f"| {k} | {v['fetch_raw']/1e6:.1f} | {v['fetch_x2']/1e6:.1f} | {v['write']/1e6:.1f} | "
                     f"{v['traffic']/1e6:.1f} | {v['traffic_per_view']/1e6:.2f} |\n")


if __name__ == "__main__":
    # argv: fetch.db write.db out_prefix [views] [build_id] [P W H sh_degree]
    wl = None
    if len(sys.argv) > 9:
        wl = dict(zip(("P", "W", "H", "sh_degree"), map(int, sys.argv[6:10])))
    main(sys.argv[1], sys.argv[2], sys.argv[3], int(sys.argv[4]) if len(sys.argv) > 4 else 32,
         sys.argv[5] if len(sys.argv) > 5 else "", wl)
