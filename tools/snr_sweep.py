#!/usr/bin/env python3
"""BASELINE configs[4]: the reference's example recording (examples/150613_1920.wav,
committed as tests/golden/150613_1920_int16.npz) + AWGN over an SNR sweep, decoded end
to end: K0 front-end -> FDR -> S0..S5 schedule (GPU) -> deinterleave + Fano + unpack
(host), beside the CPU path (oracle kernels + the same host tail) on the same frames.

usage: snr_sweep.py [seeds_per_snr] [--json out.json]      (bench.py calls run() for its configs4_n1 key)
Prints one row per SNR: decode rate of `VE3EMB FN42 33`, GPU and CPU decode sets equal?,
GPU and CPU seconds.  SNR is referred to 2500 Hz like WSPR reports; the recording's own
SNR is estimated from its 375 S/s spectrum and noise is added up to the target.
"""
import json
import os
import sys
import time
from concurrent.futures import ThreadPoolExecutor

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import gr_uwspr_amd as G  # noqa: E402
import oracle_py as O  # noqa: E402

WANT = "VE3EMB FN42 33"
PER = 4            # candidates per frame taken through the schedule
SNRS = (-20.0, -22.0, -24.0, -26.0, -28.0, -30.0)


def estimate_native(frame, f1):
    """(signal power, noise density per Hz) of the 375 S/s complex frame around f1."""
    z = frame[:, 0].astype(np.float64) + 1j * frame[:, 1].astype(np.float64)
    n = z.size
    Z = np.fft.fft(z) / n
    f = np.fft.fftfreq(n, 1 / 375.0)
    p = np.abs(Z) ** 2                       # mean-square contribution per bin
    binw = 375.0 / n
    sig = np.abs(f - f1) <= 4.0
    ref = (np.abs(f - f1) > 8.0) & (np.abs(f) < 50.0)
    n0 = np.median(p[ref]) / np.log(2.0) / binw   # exponential bins: mean = median / ln 2
    ps = p[sig].sum() - n0 * binw * sig.sum()
    return ps, n0


def decode_texts(rec):
    dec = G.decode_candidate(rec)
    return None if dec is None else G.unpack_message(dec[0])[1]


def cpu_decode(fdr, frame):
    texts = []
    for c in fdr.transform(frame)[:PER]:
        d = O.demod_candidate(c, 1500, frame)
        rec = np.zeros(1, G.native.DEMOD_DTYPE)[0]
        for k in ("f1", "drift1", "sync1", "shift1", "worth_a_try", "jig_sync", "jig_rms", "jig_shift", "symbols"):
            rec[k] = d[k]
        t = decode_texts(rec)
        if t is not None:
            texts.append(t)
    return texts


def run(seeds=8, cpu_seeds=None, quiet=False, rank=0, world=1):
    """The sweep; the CPU leg (and the GPU == CPU comparison) covers the first cpu_seeds frames of every SNR.
    rank / world (bench.py --gpus N): the `seeds` noisy copies of every SNR are sharded round-robin like the frames of the
    search path (copy s on rank s mod world, its noise a function of s: the union over the ranks is the one-rank sweep);
    this rank's rows count ITS copies, the caller adds them up; host threads = this rank's share of the host
    (uwspr_host_set_ranks)."""
    mine = list(range(rank, seeds, world))
    cpu_seeds = len(mine) if cpu_seeds is None else min(cpu_seeds, len(mine))
    say = (lambda *a: None) if quiet else print
    x = np.load(os.path.join(ROOT, "tests", "golden", "150613_1920_int16.npz"))["x"].astype(np.float32) / 32768.0
    ctx = G.Context()
    lazy = G.Context()           # the reference's own early exit: try 0 only, the rest on demand
    lazy.set_tries(1)
    O.lib(); O.pr3()
    nw = max(1, min(16, G.host_threads()))       # affinity, cgroup quota and the ranks sharing the host applied
    fdrs = [O.FDR() for _ in range(nw)]
    # the recording's own SNR is a property of the recording: estimated once through the WIDE compact front-end (the
    # flowgraph's chain passes 1500 +- 10 Hz only: no noise-reference band is left beside the signal)
    wide = G.Context(options={"frontend": G.FRONTEND_COMPACT})
    clean = wide.frontend(x[None])
    cands, out = wide.pipeline_batch(clean, max_per_frame=PER)
    wide.close()
    f1 = None
    for j in range(min(PER, len(cands[0]))):
        if decode_texts(out[0, j]) == WANT:
            f1 = float(out[0, j]["f1"])
            break
    assert f1 is not None, "the clean recording must decode"
    ps, n0 = estimate_native(clean[0], f1)
    native = 10 * np.log10(ps / (n0 * 2500.0))
    say("recording: %s at %+.2f Hz, native SNR %.1f dB in 2500 Hz" % (WANT, f1, native))
    rows = []
    for snr in SNRS:
        n0_target = ps / (2500.0 * 10 ** (snr / 10.0))
        sigma = np.sqrt(max(n0_target - n0, 0.0) * 12000.0)     # real AWGN at 12 kS/s: density sigma^2/12000 per Hz
        if not mine:                                   # more ranks than copies: this rank has nothing at this SNR
            rows.append({"snr_db": snr, "frames": 0, "decoded": 0, "other_decodes": 0, "gpu_equals_cpu": True, "gpu_s": 0.0,
                         "gpu_lazy_s": 0.0, "lazy_records_resumed": 0, "records": 0, "cpu_s": 0.0, "cpu_frames": 0,
                         "cpu_threads": nw, "gpu_frames_per_s": 0.0, "gpu_lazy_frames_per_s": 0.0, "cpu_frames_per_s": 0.0})
            continue
        audio = np.empty((len(mine), x.size), np.float32)
        for k, s in enumerate(mine):
            rng = np.random.Generator(np.random.Philox(int(1000 * -snr) + s))
            audio[k] = x + sigma * rng.standard_normal(x.size).astype(np.float32)
        seeds_l = len(mine)
        t0 = time.time()
        frames = ctx.frontend(audio)
        cands, out = ctx.pipeline_batch(frames, max_per_frame=PER)
        msg, _, okv = G.decode_batch(out, nthreads=0)      # [seeds*PER] records, host threads
        gpu_texts = []
        for b in range(seeds_l):
            gpu_texts.append([G.unpack_message(msg[b * PER + j])[1] for j in range(min(PER, len(cands[b])))
                              if okv[b * PER + j]])
        tg = time.time() - t0
        # lazy flow: uwspr_set_tries(1) -> Fano on try 0 -> uwspr_demod_resume for what did not decode
        t0 = time.time()
        frames_l = lazy.frontend(audio)
        cands_l, out_l = lazy.pipeline_batch(frames_l, max_per_frame=PER)
        msg_l, _, ok_l = G.decode_batch(out_l, nthreads=0)
        need = ((out_l["worth_a_try"] != 0) & ~ok_l.reshape(out_l.shape)).astype(np.uint8)
        resumed = int(need.sum())
        if resumed:
            out_r = lazy.demod_resume(frames_l, need, None, max_per_frame=PER)
            idx = np.flatnonzero(need.reshape(-1))
            msg_r, _, ok_r = G.decode_batch(out_r.reshape(-1)[idx], nthreads=0)
            msg_l[idx] = msg_r
            ok_l[idx] = ok_r
        lazy_texts = []
        for b in range(seeds_l):
            lazy_texts.append([G.unpack_message(msg_l[b * PER + j])[1] for j in range(min(PER, len(cands_l[b])))
                               if ok_l[b * PER + j]])
        tl = time.time() - t0
        t0 = time.time()
        cpu_texts = []
        if cpu_seeds:
            with ThreadPoolExecutor(nw) as ex:
                cpu_texts = list(ex.map(lambda a: cpu_decode(fdrs[a % nw], frames[a]), range(cpu_seeds)))
        tc = max(time.time() - t0, 1e-9)
        ok = sum(WANT in t for t in gpu_texts)
        false_dec = sum(len([u for u in t if u != WANT]) for t in gpu_texts)
        same = gpu_texts[:cpu_seeds] == cpu_texts and lazy_texts[:cpu_seeds] == cpu_texts and gpu_texts == lazy_texts
        rows.append({"snr_db": snr, "frames": seeds_l, "decoded": ok, "other_decodes": false_dec,
                     "gpu_equals_cpu": same, "gpu_s": tg, "gpu_lazy_s": tl, "lazy_records_resumed": resumed,
                     "records": int((out_l["worth_a_try"] != 0).sum()), "cpu_s": tc, "cpu_frames": cpu_seeds,
                     "cpu_threads": nw, "gpu_frames_per_s": seeds_l / tg, "gpu_lazy_frames_per_s": seeds_l / tl,
                     "cpu_frames_per_s": cpu_seeds / tc})
        say("SNR %5.1f dB: %2d/%d decoded, %d other decodes, GPU==lazy==CPU %s, GPU %.1f ms eager / %.1f ms lazy "
              "(%d of %d records resumed) (host audio in, front-end + search + Fano), CPU %.1f ms on %d threads "
              "(search + Fano, no front-end)"
              % (snr, ok, seeds_l, false_dec, same, 1e3 * tg, 1e3 * tl, resumed, rows[-1]["records"], 1e3 * tc, nw))
    ctx.close()
    lazy.close()
    return {"recording": "examples/150613_1920.wav", "native_snr_db": float(native), "front_end": "K0, option frontend = 0 (the flowgraph's GNU Radio chain)",
            "candidates_per_frame": PER, "host_threads": G.host_threads(), "rank": rank, "world": world, "rows": rows}


def main():
    seeds = int(sys.argv[1]) if len(sys.argv) > 1 and not sys.argv[1].startswith("-") else 8
    jpath = sys.argv[sys.argv.index("--json") + 1] if "--json" in sys.argv else None
    res = run(seeds)
    if jpath:
        json.dump(res, open(jpath, "w"), indent=1)
    print("mismatches:", sum(not r["gpu_equals_cpu"] for r in res["rows"]))
    return 0 if all(r["gpu_equals_cpu"] for r in res["rows"]) else 1


if __name__ == "__main__":
    sys.exit(main())
