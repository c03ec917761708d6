#!/usr/bin/env python3
"""profiles/k4_traffic.json from the PMC passes written by tools/run_profiles.sh: per schedule
form, the fabric bytes (FETCH_SIZE x2 + WRITE_SIZE, see `correction`) of the tone-correlation
launches of one bench step and their VALU issue utilisation.
usage: make_traffic.py gpurun_out/prof_<tag> <summary-file-for-the-source-field>"""
import csv
import glob
import json
import os
import sys
from collections import defaultdict

FAMILY = {"fused": ("uwspr::k6_sched",), "staged": ("uwspr::k4_",)}
STEPS = 4          # --steps 3 --warmup 1 pipeline steps per pass, + trial/aux calls are excluded by grid size


def rows(d):
    for f in glob.glob(os.path.join(d, "**", "*_counter_collection.csv"), recursive=True):
        yield from csv.DictReader(open(f))


def per_launch(d, family, counters):
    """{(kernel, grid): {counter: (launches, sum)}} for the family's launches"""
    acc = defaultdict(lambda: defaultdict(lambda: [0, 0.0]))
    for r in rows(d):
        k = r["Kernel_Name"]
        if not any(fm in k for fm in family) or r["Counter_Name"] not in counters:
            continue
        a = acc[(k.split("(")[0].replace("void ", ""), r["Grid_Size"])][r["Counter_Name"]]
        a[0] += 1
        a[1] += float(r["Counter_Value"])
    return acc


def main():
    d, src = sys.argv[1], (sys.argv[2] if len(sys.argv) > 2 else "profiles/r02_final_rocprof_summary.txt")
    out = {"correction": "FETCH_SIZE x2 (calibrated on a 2 GiB streaming read with 8-byte loads: tools/fetch_calib.hip), "
                         "WRITE_SIZE x1, KB->B x1024; fabric requests incl. Infinity-Cache hits: an upper bound on HBM traffic",
           "workload": "bench.py --no-cpu --no-sweep --sched <form> --streams 1: 256 frames per step, rotating over 5 distinct batches"}
    for form, fam in FAMILY.items():
        fe = per_launch(os.path.join(d, "pmc_fetch_" + form), fam, ("FETCH_SIZE",))
        wr = per_launch(os.path.join(d, "pmc_write_" + form), fam, ("WRITE_SIZE",))
        sq = per_launch(os.path.join(d, "pmc_sq_" + form), fam, ("SQ_INSTS_VALU", "GRBM_GUI_ACTIVE"))
        # one bench step launches every (kernel, grid) of the family once, except S1/S4 (one key, twice)
        def step_sum(acc, c):
            calls = min(v[c][0] for v in acc.values()) if acc else 1
            return sum(v[c][1] for v in acc.values()) / max(calls, 1)
        fetch = step_sum(fe, "FETCH_SIZE") * 1024 * 2
        write = step_sum(wr, "WRITE_SIZE") * 1024
        insts = step_sum(sq, "SQ_INSTS_VALU")
        cycles = step_sum(sq, "GRBM_GUI_ACTIVE") / 8.0          # summed over the 8 XCDs
        out[form] = {
            "kernels": sorted("%s grid %s" % k for k in fe),
            "bytes_per_step": fetch + write, "fetch_bytes_per_step_corrected": fetch, "write_bytes_per_step": write,
            "valu_wave_instructions_per_step": insts, "busy_cycles_per_step": cycles,
            # a SIMD issues at most one wave64 VALU instruction every 2 cycles: 1024 SIMDs x cycles / 2 slots
            "valu_issue_utilisation": insts / (cycles * 1024 / 2.0) if cycles else None,
            "source": "%s (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE / SQ_INSTS_VALU GRBM_GUI_ACTIVE passes)" % src,
        }
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, root)
    import gr_uwspr_amd as G
    # the digest of the sources the counters were TAKEN on (tools/run_profiles.sh writes it on the GPU box); a profile
    # directory of an earlier round has none: then the tree's own, as before.  bench.py refuses the figures for any other build
    dpath = os.path.join(sys.argv[1], "library_sources_sha256.txt")
    here = G.native.source_digest()
    taken = open(dpath).read().strip() if os.path.exists(dpath) else here
    if taken != here:
        print("WARNING: the counters were taken on sources %s..., this tree is %s...: bench.py will report them stale"
              % (taken[:12], here[:12]), file=sys.stderr)
    out["library_sources_sha256"] = taken
    json.dump(out, open(os.path.join(root, "profiles", "k4_traffic.json"), "w"), indent=1)
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
