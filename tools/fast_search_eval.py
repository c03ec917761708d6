#!/usr/bin/env python3
"""Option fast_search = 1 against the exact path (the north-star's FMA / shuffle-tree form of
the search stages S0..S4, S5 -- the soft symbols -- stays exact).  Over N frames at mixed SNR, every
refined candidate: how often the search lands elsewhere (shift1 / f1 / drift1 differ), how far sync1 moves,
whether the soft symbols still agree where the search agrees, and what changes in the decode set.

    python tools/fast_search_eval.py [N]        (GPU box; default 10500 frames = the round-1 soak size)
"""
import json, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import gr_uwspr_amd as G

N = int(sys.argv[1]) if len(sys.argv) > 1 else 10500
PER, CH = 2, 300
tot = dict(frames=0, cands=0, worth=0, same_search=0, shift_diff=0, f1_diff=0, drift_diff=0, worth_diff=0,
           sym_diff_given_same_search=0, dec_exact=0, dec_fast=0, dec_only_exact=0, dec_only_fast=0, msg_diff=0)
max_rel = 0.0
rels = []
worst = None
t_exact = t_fast = 0.0
cx = G.Context(options={"sched": 0})
cf = G.Context(options={"fast_search": 1})
snrs = (-18.0, -22.0, -25.0, -27.0, -29.0, -31.0, None)
done = 0
k = 0
while done < N:
    n = min(CH, N - done)
    snr = snrs[k % len(snrs)]
    if snr is None:
        fr = (0.5 * np.random.default_rng(k).standard_normal((n, 45000, 2))).astype(np.float32)
    else:
        fr = G.synth.make_frames(n, seed=0xFA57 + 1000 * k, snr_db=snr, maxdrift=0.0)
    t0 = time.perf_counter(); ce, oe = cx.pipeline_batch(fr, max_per_frame=PER); t_exact += time.perf_counter() - t0
    t0 = time.perf_counter(); cq, of = cf.pipeline_batch(fr, max_per_frame=PER); t_fast += time.perf_counter() - t0
    me, _, oke = G.decode_batch(oe.reshape(-1))
    mf, _, okf = G.decode_batch(of.reshape(-1))
    for b in range(n):
        assert ce[b].tobytes() == cq[b].tobytes()          # the coarse search is untouched
        for j in range(min(PER, len(ce[b]))):
            i = b * PER + j
            a, f = oe[b, j], of[b, j]
            tot["cands"] += 1
            tot["worth"] += int(a["worth_a_try"])
            sd = int(a["shift1"]) != int(f["shift1"]); fd = float(a["f1"]) != float(f["f1"]); dd = float(a["drift1"]) != float(f["drift1"])
            wd = int(a["worth_a_try"]) != int(f["worth_a_try"])
            tot["shift_diff"] += sd; tot["f1_diff"] += fd; tot["drift_diff"] += dd; tot["worth_diff"] += wd
            if a["sync1"] != 0:
                rel = abs(float(f["sync1"]) - float(a["sync1"])) / abs(float(a["sync1"]))
                rels.append(rel)
                if rel > max_rel:
                    max_rel, worst = rel, {"sync1_exact": float(a["sync1"]), "sync1_fast": float(f["sync1"]), "snr_db": snr,
                                           "worth_a_try": int(a["worth_a_try"])}
            if not (sd or fd or dd or wd):
                tot["same_search"] += 1
                tot["sym_diff_given_same_search"] += int(a["symbols"].tobytes() != f["symbols"].tobytes())
            tot["dec_exact"] += int(oke[i]); tot["dec_fast"] += int(okf[i])
            tot["dec_only_exact"] += int(oke[i] and not okf[i]); tot["dec_only_fast"] += int(okf[i] and not oke[i])
            tot["msg_diff"] += int(oke[i] and okf[i] and me[i].tobytes() != mf[i].tobytes())
    done += n
    tot["frames"] = done
    k += 1
    if k % 5 == 0:
        print("...", done, "frames", flush=True)
cx.close(); cf.close()
tot["sync1_max_rel_err"] = max_rel
rels = np.array(rels)
tot["sync1_rel_err"] = {"median": float(np.median(rels)), "p99": float(np.quantile(rels, 0.99)), "p999": float(np.quantile(rels, 0.999)),
                        "above_1e-5": int((rels > 1e-5).sum()), "of": int(rels.size), "worst": worst}
tot["host_call_seconds_exact_fast"] = [t_exact, t_fast]
print(json.dumps(tot, indent=1))
os.makedirs("gpurun_out", exist_ok=True)
json.dump(tot, open("gpurun_out/fast_search_eval.json", "w"), indent=1)
