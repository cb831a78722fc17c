#!/usr/bin/env python
"""Brute-force 3-NN (ggs_dist2_3nn) at 100k points, GPU-side time with HIP events, first call and steady state separately:
VERDICT r4 flagged 13.4 ms in r04_next_rows.md against 4.17 ms in r03 with an unchanged kernel."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "gaussian-garments_amd"))
import torch  # noqa: E402
from ggsplat import synthetic as S  # noqa: E402
from simple_knn._C import distCUDA2  # noqa: E402

v, f = S.skirt_mesh()
pts = v[f].mean(1).cuda()
ev = [torch.cuda.Event(enable_timing=True) for _ in range(42)]
torch.cuda.synchronize()
for i in range(41):
    ev[i].record()
    distCUDA2(pts, brute_force=True)
ev[41].record()
torch.cuda.synchronize()
ms = [ev[i].elapsed_time(ev[i + 1]) for i in range(41)]
print(f"brute-force 3-NN, {pts.shape[0]} points: call 1 {ms[0]:.3f} ms, calls 2-5 {sum(ms[1:5]) / 4:.3f} ms, "
      f"calls 6-41 mean {sum(ms[5:]) / 36:.3f} / min {min(ms[5:]):.3f} / max {max(ms[5:]):.3f} ms")
