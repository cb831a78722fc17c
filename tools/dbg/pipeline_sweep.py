#!/usr/bin/env python
"""Round 5: serial vs pipelined (two-stream) step of bench.py over launch-set sizes, graph-replayed and eager, plus the 20-view
rank step of an 8-GPU run on one GPU.  Prints one line per configuration: views/s, ms per step.  Usage: pipeline_sweep.py [reps]"""
import json
import re
import subprocess
import sys

BASE = [sys.executable, "bench.py", "--cpu-views", "0", "--loop-views", "0", "--extra-configs", "0", "--warmup", "5", "--timing-only"]
CONFIGS = [
    ("160v chunk80 serial  m2d0", ["--steps", "60", "--chunk", "80", "--means2d", "0", "--pipeline", "0"]),
    ("160v chunk80 serial  m2d1", ["--steps", "60", "--chunk", "80", "--pipeline", "0"]),
    ("160v chunk80 piped       ", ["--steps", "60", "--chunk", "80", "--pipeline", "1"]),
    ("160v chunk40 serial      ", ["--steps", "60", "--chunk", "40", "--pipeline", "0"]),
    ("160v chunk40 piped       ", ["--steps", "60", "--chunk", "40", "--pipeline", "1"]),
    ("160v chunk40 staged      ", ["--steps", "60", "--chunk", "40", "--pipeline", "2"]),
    ("160v chunk20 staged      ", ["--steps", "60", "--chunk", "20", "--pipeline", "2"]),
    ("160v chunk32 piped       ", ["--steps", "60", "--chunk", "32", "--pipeline", "1"]),
    ("160v chunk20 serial      ", ["--steps", "60", "--chunk", "20", "--pipeline", "0"]),
    ("160v chunk20 piped       ", ["--steps", "60", "--chunk", "20", "--pipeline", "1"]),
    ("160v chunk16 piped       ", ["--steps", "60", "--chunk", "16", "--pipeline", "1"]),
    ("160v chunk10 piped       ", ["--steps", "60", "--chunk", "10", "--pipeline", "1"]),
    ("160v chunk40 piped eager ", ["--steps", "60", "--chunk", "40", "--pipeline", "1", "--no-graph"]),
    ("160v chunk40 serial eager", ["--steps", "60", "--chunk", "40", "--no-graph", "--pipeline", "0"]),
    ("rank: 20v chunk20 serial ", ["--steps", "300", "--views", "20", "--chunk", "20", "--pipeline", "0"]),
    ("rank: 20v chunk10 serial ", ["--steps", "300", "--views", "20", "--chunk", "10", "--pipeline", "0"]),
    ("rank: 20v chunk10 piped  ", ["--steps", "300", "--views", "20", "--chunk", "10", "--pipeline", "1"]),
    ("rank: 20v chunk5  piped  ", ["--steps", "300", "--views", "20", "--chunk", "5", "--pipeline", "1"]),
    ("rank: 20v overlap 2      ", ["--steps", "300", "--views", "20", "--chunk", "10", "--overlap", "2", "--pipeline", "0"]),
]
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 2
sel = sys.argv[2] if len(sys.argv) > 2 else ""
for rep in range(reps):
    for name, extra in CONFIGS:
        if sel and not re.search(sel, name):
            continue
        p = subprocess.run(BASE + extra, capture_output=True, text=True)
        line = [l for l in p.stdout.splitlines() if l.startswith("{")]
        if not line:
            print(f"{name} rep {rep}: FAILED rc={p.returncode} {p.stderr[-400:]}", flush=True)
            continue
        d = json.loads(line[-1])
        print(f"{name} rep {rep}: {d['value']:9.1f} views/s  {d['ms_per_step']:7.3f} ms/step", flush=True)
