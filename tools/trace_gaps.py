#!/usr/bin/env python3
"""Timeline of one bench step from a rocprofv3 kernel trace: kernel durations and
the idle gaps between consecutive kernels.  usage: trace_gaps.py <kernel_trace.csv> [step_index]"""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# a step starts at each k1_spectrogram
starts = [i for i, r in enumerate(rows) if "k1_spectrogram" in r["Kernel_Name"]]
which = int(sys.argv[2]) if len(sys.argv) > 2 else len(starts) // 2
a, b = starts[which], starts[which + 1]
prev_end = None
tot_k = tot_gap = 0
for r in rows[a:b]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    gap = (s - prev_end) / 1e3 if prev_end else 0.0
    name = r["Kernel_Name"].split("(")[0].replace("void ", "")[:50]
    print("%8.2f us gap  %8.2f us  %s" % (gap, (e - s) / 1e3, name))
    tot_k += e - s
    tot_gap += max(0, s - prev_end) if prev_end else 0
    prev_end = e
print("kernels %.1f us, gaps %.1f us, step span %.1f us" % (tot_k / 1e3, tot_gap / 1e3,
      (int(rows[b]["Start_Timestamp"]) - int(rows[a]["Start_Timestamp"])) / 1e3))
