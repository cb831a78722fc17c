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


def lds_table(path, out, header):
    """LDS pipe counters per kernel (profile_all.sh pass `lds`): instructions, busy / stall / bank-conflict cycles."""
    cur = sqlite3.connect(path).cursor()
    rows = {}
    for name, counter, n, total in cur.execute(
            "select kernel_name, counter_name, count(*), sum(value) from counters_collection group by kernel_name, counter_name"):
        m = re.match(r"(ggs_k_\w+)", name)
        if m:
            rows.setdefault(m.group(1), {})[counter] = (n, total)
    with open(out, "w") as f:
        f.write(header)
        f.write("| kernel | waves / launch | LDS instr / wave | ACTIVE_INST_LDS % of wave-cycles | WAIT_INST_LDS % | LDS_IDX_ACTIVE / LDS instr | "
                "BANK_CONFLICT / LDS_IDX_ACTIVE % | LDS_IDX_ACTIVE / BUSY_CYCLES % |\n|---|---|---|---|---|---|---|---|\n")
        for k, c in sorted(rows.items()):
            g = lambda n: c.get(n, (1, 0.0))[1]
            launches = c["SQ_WAVES"][0]
            waves = max(g("SQ_WAVES"), 1.0)
            wc = max(g("SQ_WAVE_CYCLES"), 1.0)
            li = max(g("SQ_INSTS_LDS"), 1.0)
            idx = max(g("SQ_LDS_IDX_ACTIVE"), 1.0)
            f.write(f"| {k} | {waves / launches:.0f} | {li / waves:.1f} | {100 * g('SQ_ACTIVE_INST_LDS') / wc:.1f} | "
                    f"{100 * g('SQ_WAIT_INST_LDS') / wc:.1f} | {idx / li:.2f} | {100 * g('SQ_LDS_BANK_CONFLICT') / idx:.1f} | "
                    f"{100 * idx / max(g('SQ_BUSY_CYCLES'), 1.0):.1f} |\n")


def generic_table(path, out, header):
    """Any --pmc pass: every collected counter per launch of each kernel of this library, beside the kernel's duration."""
    cur = sqlite3.connect(path).cursor()
    rows, names = {}, []
    for name, counter, n, total in cur.execute(
            "select kernel_name, counter_name, count(*), sum(value) from counters_collection group by kernel_name, counter_name"):
        m = re.match(r"(ggs_k_\w+|\(anonymous namespace\)::k_\w+)", name)
        if m:
            rows.setdefault(m.group(1), {})[counter] = (n, total)
            if counter not in names:
                names.append(counter)
    dur = {}
    for name, avg in cur.execute("select name, avg(end-start) from kernels group by name"):
        m = re.match(r"(ggs_k_\w+|\(anonymous namespace\)::k_\w+)", name)
        if m:
            dur[m.group(1)] = avg / 1e3
    names.sort()
    with open(out, "w") as f:
        f.write(header)
        f.write("| kernel | launches | " + " | ".join(n + " / launch" for n in names) + " | kernel us |\n|---|---|" + "---|" * (len(names) + 1) + "\n")
        for k, c in sorted(rows.items()):
            n = max(v[0] for v in c.values())
            f.write(f"| {k} | {n} | " + " | ".join(f"{c.get(x, (1, 0.0))[1] / n:.4g}" for x in names) + f" | {dur.get(k, 0):.1f} |\n")


def valu_table(path, prefix, bid, bargs, workload, derived_path=None):
    """VALUBusy = 100 sum(SQ_ACTIVE_INST_VALU) / CU_NUM / max(GRBM_GUI_ACTIVE) and VALUUtilization = 100 sum(SQ_THREAD_CYCLES_VALU)
    / (sum(SQ_ACTIVE_INST_VALU) 64) -- rocprofiler-sdk's own gfx950 formulas (counter_defs.yaml), evaluated per kernel."""
    cur = sqlite3.connect(path).cursor()
    rows = {}
    for name, counter, n, total, mx in cur.execute(
            "select kernel_name, counter_name, count(*), sum(value), max(value) from counters_collection group by kernel_name, counter_name"):
        m = re.match(r"(ggs_k_\w+)", name)
        if m:
            rows.setdefault(m.group(1), {})[counter] = (n, total, mx)
    dur = {}
    for name, avg in cur.execute("select name, avg(end-start) from kernels group by name"):
        m = re.match(r"(ggs_k_\w+)", name)
        if m:
            dur[m.group(1)] = avg / 1e3
    CU, XCD = 256, 8                      # GRBM_GUI_ACTIVE arrives summed over the 8 XCDs; the formula wants the max = sum / 8
    derived = {}
    if derived_path:                      # rocprofv3 evaluating its own derived metrics (per dispatch; averaged here)
        for name, counter, avg in sqlite3.connect(derived_path).cursor().execute(
                "select kernel_name, counter_name, avg(value) from counters_collection group by kernel_name, counter_name"):
            m = re.match(r"(ggs_k_\w+)", name)
            if m:
                derived.setdefault(m.group(1), {})[counter] = avg
    out = {"build_id": bid, "workload": workload, "bench_args": bargs, "views_per_launch": int(os.environ.get("PMC_VIEWS", "32")), "kernels": {}}
    with open(prefix + "_valu.md", "w") as f:
        f.write(f"# VALU occupancy and lane activity (rocprofv3 --pmc, own pass), build {bid}\n\n`python bench.py --cpu-views 0 --loop-views 0 "
                f"--extra-configs 0 {bargs} --steps 1 --warmup 0 --views {out['views_per_launch']}`.  VALUBusy and VALUUtilization are "
                "rocprofiler-sdk's gfx950 formulas; GRBM_GUI_ACTIVE is taken per XCD (the counter arrives summed over the 8) and summed over the launches of a kernel.\n\n| kernel | launches | VALU insts / launch | GRBM_GUI_ACTIVE cycles / launch | VALUBusy % | active lanes per VALU "
                "instruction (VALUUtilization %) | SQ_BUSY_CYCLES / launch | rocprofv3 VALUBusy | rocprofv3 VALUUtilization |\n|---|---|---|---|---|---|---|---|---|\n")
        for k, c in sorted(rows.items()):
            g = lambda n: c.get(n, (1, 0.0, 0.0))[1]
            n = c["SQ_INSTS_VALU"][0]
            gui = max(g("GRBM_GUI_ACTIVE") / XCD, 1.0)
            act = max(g("SQ_ACTIVE_INST_VALU"), 1.0)
            busy = 100.0 * act / CU / gui
            util = 100.0 * g("SQ_THREAD_CYCLES_VALU") / (act * 64.0)
            out["kernels"][k] = {"valu_busy_pct": round(busy, 2), "lane_activity_pct": round(util, 2), "valu_insts_per_launch": g("SQ_INSTS_VALU") / n,
                                 "gui_active_cycles_per_launch": gui / n,
                                 # raw inputs of bench.py's counter-derived figures (round 6): quad-cycles with a VALU instruction in flight,
                                 # summed over the waves; SIMD-busy quad-cycles; the kernel's duration in THIS (profiled) pass
                                 "active_inst_valu_per_launch": g("SQ_ACTIVE_INST_VALU") / n, "sq_busy_cycles_per_launch": g("SQ_BUSY_CYCLES") / n,
                                 "kernel_us": dur.get(k)}
            dv = derived.get(k, {})
            if dv:
                out["kernels"][k].update(rocprof_valu_busy_pct=dv.get("VALUBusy"), rocprof_lane_activity_pct=dv.get("VALUUtilization"))
            f.write(f"| {k} | {n} | {g('SQ_INSTS_VALU') / n:.3e} | {gui / n:.3e} | {busy:.1f} | {util:.1f} | {g('SQ_BUSY_CYCLES') / n:.3e} | "
                    f"{dv.get('VALUBusy', float('nan')):.1f} | {dv.get('VALUUtilization', float('nan')):.1f} |\n")
    json.dump(out, open(prefix + "_valu.json", "w"), indent=1)


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
    l = db(tag, "lds")
    if l:
        lds_table(l, prefix + "_lds_counters.md",
                  f"# LDS counters (rocprofv3 --pmc, one pass, kernel-trace only), build {bid}\n\n`python bench.py --cpu-views 0 "
                  f"--loop-views 0 --extra-configs 0 {bargs} --steps 1 --warmup 0 --views 32`; SQ cycle counters are in quad-cycles, "
                  f"summed over the SQs they are collected on.\n\n")
    u = db(tag, "valu")
    if u:
        valu_table(u, prefix, bid, bargs, workload, db(tag, "valud"))
    for kind, what in (("atomic", "L2 atomic requests"), ("vmem", "vector-memory instructions and wave waits")):
        a = db(tag, kind)
        if a:
            generic_table(a, prefix + f"_{kind}_counters.md",
                          f"# {what} (rocprofv3 --pmc, one pass, kernel-trace only), build {bid}\n\n`python bench.py --cpu-views 0 "
                          f"--loop-views 0 --extra-configs 0 {bargs} --pipeline 0 --steps 1 --warmup 0 --views {os.environ.get('PMC_VIEWS', '32')}`; "
                          f"SQ cycle counters are in quad-cycles.\n\n")
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
