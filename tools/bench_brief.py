#!/usr/bin/env python3
"""One-screen summary of a bench.py JSON line (stdin or a file)."""
import json
import sys
txt = open(sys.argv[1]).read() if len(sys.argv) > 1 else sys.stdin.read()
d = json.loads(txt.strip().splitlines()[-1])
print("value %.0f frames/s  ms/step %.4f  chosen %s" % (d["value"], d["ms_per_step"], d.get("chosen")))
print("repeats", [round(x) for x in d["repeats"]["frames_per_s"]])
print("single stream", d.get("single_stream"))
print("trial ms/step", d["config"].get("trial_ms_per_step"))
r = d["roofline"]
print("roofline frac %.4f  kernel_ms_per_step %.4f  achieved %.2f" % (r["frac"], r["kernel_ms_per_step"], r["achieved"]))
for k in ("end_to_end_decoded", "fast_search"):
    if k in d:
        v = d[k]
        if not isinstance(v, dict):
            continue
        print(k, {q: v[q] for q in v if q in ("frames_per_s", "min", "max", "value")})
sc = d.get("parity_spot_check")
if sc:
    print("parity_spot_check: %d frames, equal %s, form %s, lanes %s %s" % (sc["frames"], sc["equal"], sc["form"], sc["lanes"], sc["mismatches"] or ""))
print("traffic", r.get("traffic"), "stale:", r.get("traffic_stale"))
for k in ("configs4_n1", "configs4"):
    if isinstance(d.get(k), dict) and "rows" in d[k]:
        print(k, [(x["snr_db"], x["decoded"], x["frames"], round(x["gpu_frames_per_s"])) for x in d[k]["rows"]])
if isinstance(d.get("cpu_baseline"), dict):
    print("cpu_baseline %.0f frames/s on %d threads (%s)" % (d["cpu_baseline"]["value"], d["cpu_baseline"]["cores"], d["cpu_baseline"].get("cpu_model")))
