#!/usr/bin/env python3
"""Concurrency in a multi-stream rocprofv3 kernel trace: fraction of the busy span with
0/1/2/3+ kernels running, with >=1 K4 kernel running, and per-queue activity.
usage: trace_overlap.py <dir-or-kernel_trace.csv>"""
import csv
import glob
import os
import sys
from collections import defaultdict

p = sys.argv[1]
if os.path.isdir(p):
    p = sorted(glob.glob(os.path.join(p, "**", "*_kernel_trace.csv"), recursive=True))[0]
rows = [r for r in csv.DictReader(open(p)) if "uwspr::" in r["Kernel_Name"]]
# pipeline part only: up to the first sweep kernel (k_grid_hyps / k_prep_hyps)
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
cut = next((i for i, r in enumerate(rows) if "k_grid_hyps" in r["Kernel_Name"] or "k_prep_hyps" in r["Kernel_Name"]), len(rows))
rows = rows[:cut]
# skip warmup: keep the middle 60 %
n = len(rows)
rows = rows[n // 5: n - n // 5]
ev = []
for r in rows:
    k4 = "k4_" in r["Kernel_Name"]
    ev.append((int(r["Start_Timestamp"]), 1, k4))
    ev.append((int(r["End_Timestamp"]), -1, k4))
ev.sort()
t_prev = ev[0][0]
run = k4run = 0
hist = defaultdict(int)
k4busy = 0
for t, d, k4 in ev:
    dt = t - t_prev
    hist[min(run, 4)] += dt
    if k4run > 0:
        k4busy += dt
    run += d
    if k4:
        k4run += d
    t_prev = t
span = ev[-1][0] - ev[0][0]
print("span %.2f ms, %d kernels" % (span / 1e6, len(rows)))
for k in sorted(hist):
    print("  %s kernels running: %5.1f %%" % (("%d" % k) if k < 4 else ">=4", 100.0 * hist[k] / span))
print("  >=1 K4 kernel running: %5.1f %%" % (100.0 * k4busy / span))
q = defaultdict(int)
for r in rows:
    q[r.get("Queue_Id", "?")] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
print("  per-queue busy %:", {k: round(100.0 * v / span, 1) for k, v in q.items()})
