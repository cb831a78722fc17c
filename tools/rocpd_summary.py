#!/usr/bin/env python
"""Summarise a rocprofv3 rocpd SQLite result (this image's rocprofv3 writes .db, not CSV) into the
per-kernel --stats table: calls, total / average / min / max duration, share of GPU time.
Usage: python tools/rocpd_summary.py gpurun_out/prof/x_results.db > profiles/rNN_kernel_stats.md
       python tools/rocpd_summary.py x_results.db --cycles 64 --anchor k_adam_multi
           only the last 64 periods of a periodic workload (an iteration replayed as a graph): the window from the end of the
           65th-last dispatch of the anchor kernel to the end of its last one, so set-up work is left out"""
import sqlite3
import sys


def main(path, top=30, cycles=0, anchor=None):
    cur = sqlite3.connect(path).cursor()
    where = ""
    if cycles:
        ends = [r[0] for r in cur.execute("select end from kernels where name like ? order by end", (f"%{anchor}%",))]
        if len(ends) <= cycles:
            raise SystemExit(f"only {len(ends)} dispatches of {anchor}")
        t0, t1 = ends[-cycles - 1], ends[-1]
        where = f"where end > {t0} and end <= {t1} "
    rows = list(cur.execute(
        "select name, count(*), sum(end-start), avg(end-start), min(end-start), max(end-start), "
        "max(vgpr_count), max(sgpr_count), max(lds_size), max(scratch_size), max(grid_x), max(grid_y), max(workgroup_x) "
        f"from kernels {where}group by name order by 3 desc"))
    tot = sum(r[2] for r in rows) or 1
    if cycles:
        print(f"Last {cycles} periods (anchor `{anchor}`): {(t1 - t0) / cycles / 1e3:.1f} us per period on the GPU timeline, "
              f"{tot / cycles / 1e3:.1f} us of it inside kernels, {sum(r[1] for r in rows) / cycles:.1f} dispatches per period\n")
    print("| kernel | calls | total ms | avg us | min us | max us | % | vgpr | sgpr | lds B | scratch B | grid(x,y) | wg |")
    print("|---|---|---|---|---|---|---|---|---|---|---|---|---|")
    for r in rows[:top]:
        print(f"| `{r[0][:80]}` | {r[1]} | {r[2]/1e6:.3f} | {r[3]/1e3:.1f} | {r[4]/1e3:.1f} | {r[5]/1e3:.1f} | "
              f"{100*r[2]/tot:.1f} | {r[6]} | {r[7]} | {r[8]} | {r[9]} | {r[10]},{r[11]} | {r[12]} |")


if __name__ == "__main__":
    a = sys.argv[1:]
    kw = {}
    if "--cycles" in a:
        kw["cycles"] = int(a[a.index("--cycles") + 1])
        kw["anchor"] = a[a.index("--anchor") + 1]
    main(a[0], **kw)
