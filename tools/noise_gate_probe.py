#!/usr/bin/env python3
"""How often does a frame WITHOUT a decodable transmission reach the Fano decoder, and what does that
cost the host?  (sync_and_demodulate_impl.cc:443 worth_a_try = sync1 > 0.10; cc:470-480 gates
sync > 0.12 and rms > 40.6 per jiggered try; a try that does not decode runs to the 10000-cycles-per-bit
time-out.)  Noise-only frames and frames holding a transmission that starts outside the search range."""
import sys, os, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import gr_uwspr_amd as G

def stats(name, fr):
    c = G.Context()
    cands, out = c.pipeline_batch(fr, max_per_frame=1)
    c.close()
    r = out[:, 0]
    worth = r["worth_a_try"] > 0
    gate = (r["jig_sync"] > 0.12) & (r["jig_rms"] > 52.0 * 50 / 64) & worth[:, None]
    t0 = time.perf_counter()
    msgs, idt, ok = G.decode_batch(r)
    dt = time.perf_counter() - t0
    print("%-28s frames %4d  worth %5.1f %%  tries through the gate per frame %5.2f  decoded %5.1f %%  host %.1f ms/frame (%d threads)"
          % (name, len(fr), 100 * worth.mean(), gate.sum() / len(fr), 100 * ok.mean(), 1e3 * dt / len(fr), G.host_threads()))

n = 128
rng = np.random.default_rng(1)
stats("noise only", (G.synth.sigma_for_snr(-20.0) * rng.standard_normal((n, 45000, 2))).astype(np.float32))
sig = G.synth.make_frames(n, seed=5, snr_db=None)[:, 375:375 + 162 * 256]
for start in (375, 6000, 20000, -20000):
    fr = (G.synth.sigma_for_snr(-20.0) * rng.standard_normal((n, 45000, 2))).astype(np.float32)
    if start >= 0:
        m = min(sig.shape[1], 45000 - start)
        fr[:, start:start + m] += sig[:, :m]
    else:
        fr[:, :sig.shape[1] + start] += sig[:, -start:]
    stats("transmission at %d" % start, fr)
