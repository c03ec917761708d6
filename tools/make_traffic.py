#!/usr/bin/env python3
"""profiles/k4_traffic.json from the FETCH_SIZE / WRITE_SIZE passes written by
tools/run_profiles.sh: fabric bytes of the K4 launches of one pipeline step.
usage: make_traffic.py gpurun_out/prof_<tag> [summary-name-for-the-source-field]"""
import csv
import glob
import json
import os
import sys
from collections import defaultdict

PIPELINE_GRIDS = {"165888", "497664", "331776", "829440", "196608"}   # the 6 K4 launches of a 256-frame step


def per_step(d, counter):
    tot = defaultdict(lambda: [0, 0.0])
    for f in glob.glob(os.path.join(d, "**", "*_counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"]
            if "uwspr::k4_" not in k or r["Counter_Name"] != counter or r["Grid_Size"] not in PIPELINE_GRIDS:
                continue
            a = tot[(k.split("(")[0], r["Grid_Size"])]
            a[0] += 1
            a[1] += float(r["Counter_Value"])
    # launches per step: S1 and S4 share one (kernel, grid) key; S0 does too when the ring form is off
    calls = min(n for n, _ in tot.values())
    return sum(v for _, v in tot.values()) / calls, {"%s grid %s" % k: v[1] / v[0] for k, v in tot.items()}


def main():
    d = sys.argv[1]
    src = sys.argv[2] if len(sys.argv) > 2 else "profiles/r01_final_rocprof_summary.txt"
    fetch_kb, fdetail = per_step(os.path.join(d, "pmc_fetch"), "FETCH_SIZE")
    write_kb, _ = per_step(os.path.join(d, "pmc_write"), "WRITE_SIZE")
    fetch = fetch_kb * 1024 * 2
    write = write_kb * 1024
    out = {
        "kernel_family": "k4 (S0 k4_group<5>, S1/S4 k4_fstage<5>, S2 k4_tonecorr<1>, S3 k4_ring<5,16>, S5 k4_ring<6,8>): the 6 launches of one bench step",
        "bytes_per_launch": (fetch + write) / 6.0,
        "fetch_bytes_per_step_corrected": fetch,
        "write_bytes_per_step": write,
        "fetch_kb_per_launch_uncorrected": fdetail,
        "correction": "FETCH_SIZE x2 (calibrated on a 2 GiB streaming read with 8-byte loads: tools/fetch_calib.hip), WRITE_SIZE x1, KB->B x1024",
        "note": "the 92 MB batch is resident in the 256 MiB Infinity Cache; FETCH_SIZE counts fabric requests incl. Infinity-Cache hits, so this is an upper bound on HBM traffic",
        "source": "%s (rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE passes of bench.py --steps 3 --warmup 1 --no-cpu)" % src,
    }
    json.dump(out, open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "profiles", "k4_traffic.json"), "w"), indent=1)
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
