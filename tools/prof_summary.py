#!/usr/bin/env python3
"""Summarise rocprofv3 CSV output (kernel trace and/or PMC counter collection)
per kernel: calls, average duration, and average counter value per launch.
usage: prof_summary.py <dir> [name-filter]"""
import csv
import glob
import os
import sys
from collections import defaultdict


def short(name):
    n = name.split("(")[0]
    return n.replace("void ", "")[:60]


def main():
    d = sys.argv[1]
    filt = sys.argv[2] if len(sys.argv) > 2 else "uwspr"
    for f in sorted(glob.glob(os.path.join(d, "**", "*_kernel_trace.csv"), recursive=True)):
        agg = defaultdict(lambda: [0, 0.0, defaultdict(int)])
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"]
            if filt not in k:
                continue
            a = agg[short(k)]
            a[0] += 1
            a[1] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
            a[2][r.get("Grid_Size", "?")] += 1
        print("# kernel trace:", os.path.relpath(f, d))
        print("%-44s %6s %12s %12s" % ("kernel", "calls", "avg_us", "total_ms"))
        for k, (n, t, g) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
            print("%-44s %6d %12.2f %12.3f" % (k, n, t / n / 1e3, t / 1e6))
    for f in sorted(glob.glob(os.path.join(d, "**", "*_counter_collection.csv"), recursive=True)):
        agg = defaultdict(lambda: defaultdict(lambda: [0, 0.0, 0.0]))
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"]
            if filt not in k:
                continue
            key = (short(k), r["Grid_Size"])
            a = agg[key][r["Counter_Name"]]
            a[0] += 1
            a[1] += float(r["Counter_Value"])
            a[2] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
        print("# counters:", os.path.relpath(f, d))
        print("%-40s %10s %-12s %6s %16s %10s" % ("kernel", "grid", "counter", "calls", "avg_value", "avg_us"))
        for (k, g), cs in sorted(agg.items()):
            for c, (n, v, t) in cs.items():
                print("%-40s %10s %-12s %6d %16.3f %10.2f" % (k, g, c, n, v / n, t / n / 1e3))


if __name__ == "__main__":
    main()
