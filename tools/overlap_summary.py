#!/usr/bin/env python
"""How much of a rocprofv3 kernel trace (rocpd .db) ran CONCURRENTLY: busy time of the GPU timeline (union of all kernel
intervals), sum of the kernel durations, and for every pair of kernel names the time both had a dispatch in flight.
Usage: python tools/overlap_summary.py x_results.db [--last-ms 50]   (only the last N ms of the trace: the timed region)"""
import sqlite3
import sys


def main(path, last_ms=0.0, top=8):
    cur = sqlite3.connect(path).cursor()
    rows = list(cur.execute("select name, start, end from kernels order by start"))
    if not rows:
        raise SystemExit("no kernels in the trace")
    t_end = max(r[2] for r in rows)
    if last_ms:
        rows = [r for r in rows if r[1] >= t_end - last_ms * 1e6]
    short = lambda n: n.split("(")[0].replace("ggs_k_", "")[:28]
    # sweep line over start / end events
    ev = []
    for i, (n, s, e) in enumerate(rows):
        ev.append((s, 1, i))
        ev.append((e, 0, i))
    ev.sort()
    active, last_t, busy, conc = set(), ev[0][0], 0, 0
    pair = {}
    for t, kind, i in ev:
        dt = t - last_t
        if active and dt > 0:
            busy += dt
            if len(active) > 1:
                conc += dt
                names = sorted({short(rows[j][0]) for j in active})
                for a in range(len(names)):
                    for b in range(a, len(names)):
                        if a != b or sum(1 for j in active if short(rows[j][0]) == names[a]) > 1:
                            pair[(names[a], names[b])] = pair.get((names[a], names[b]), 0) + dt
        last_t = t
        (active.add if kind else active.discard)(i)
    tot = sum(e - s for _, s, e in rows)
    span = rows and (max(r[2] for r in rows) - min(r[1] for r in rows))
    print(f"{len(rows)} dispatches, timeline span {span / 1e6:.3f} ms, GPU busy (union) {busy / 1e6:.3f} ms, "
          f"sum of kernel durations {tot / 1e6:.3f} ms, >= 2 kernels in flight {conc / 1e6:.3f} ms ({100.0 * conc / max(busy, 1):.1f} % of busy)\n")
    print("| kernels in flight together | ms | % of busy |\n|---|---|---|")
    for (a, b), v in sorted(pair.items(), key=lambda kv: -kv[1])[:top]:
        print(f"| `{a}` + `{b}` | {v / 1e6:.3f} | {100.0 * v / max(busy, 1):.1f} |")
    per = {}
    for n, s, e in rows:
        d = per.setdefault(short(n), [0, 0])
        d[0] += 1
        d[1] += e - s
    print("\n| kernel | calls | total ms | avg us |\n|---|---|---|---|")
    for n, (c, d) in sorted(per.items(), key=lambda kv: -kv[1][1])[:12]:
        print(f"| `{n}` | {c} | {d / 1e6:.3f} | {d / c / 1e3:.1f} |")


if __name__ == "__main__":
    a = sys.argv[1:]
    kw = {}
    if "--last-ms" in a:
        kw["last_ms"] = float(a[a.index("--last-ms") + 1])
    main(a[0], **kw)
