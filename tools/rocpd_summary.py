#!/usr/bin/env python
"""Summarise a rocprofv3 rocpd SQLite result (this image's rocprofv3 writes .db, not CSV) into the
per-kernel --stats table: calls, total / average / min / max duration, share of GPU time.
Usage: python tools/rocpd_summary.py gpurun_out/prof/x_results.db > profiles/rNN_kernel_stats.md"""
import sqlite3
import sys


def main(path, top=30):
    cur = sqlite3.connect(path).cursor()
    rows = list(cur.execute(
        "select name, count(*), sum(end-start), avg(end-start), min(end-start), max(end-start), "
        "max(vgpr_count), max(sgpr_count), max(lds_size), max(scratch_size), max(grid_x), max(grid_y), max(workgroup_x) "
        "from kernels group by name order by 3 desc"))
    tot = sum(r[2] for r in rows) or 1
    print("| kernel | calls | total ms | avg us | min us | max us | % | vgpr | sgpr | lds B | scratch B | grid(x,y) | wg |")
    print("|---|---|---|---|---|---|---|---|---|---|---|---|---|")
    for r in rows[:top]:
        print(f"| `{r[0][:80]}` | {r[1]} | {r[2]/1e6:.3f} | {r[3]/1e3:.1f} | {r[4]/1e3:.1f} | {r[5]/1e3:.1f} | "
              f"{100*r[2]/tot:.1f} | {r[6]} | {r[7]} | {r[8]} | {r[9]} | {r[10]},{r[11]} | {r[12]} |")


if __name__ == "__main__":
    main(sys.argv[1])
