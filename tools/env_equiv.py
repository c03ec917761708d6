"""Byte-equality of the staged schedule under options (GPU box):
    python3 tools/env_equiv.py stage_kernels=0 [reuse=0 ...]
runs uwspr_pipeline_batch + uwspr_demod_batch on seeded frames with the default options and with the given
ones (fresh contexts, staged form) and compares every output byte."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import gr_uwspr_amd as G  # noqa: E402


def run(opts, frames, maxdrift):
    o = {"sched": 0}
    o.update(opts)
    c = G.Context(maxdrift=maxdrift, options=o)
    try:
        cands, out = c.pipeline_batch(frames, max_per_frame=3)
        out2 = c.demod_batch(frames, cands, max_per_frame=3)
        return b"".join(x.tobytes() for x in cands) + out.tobytes() + out2.tobytes()
    finally:
        c.close()


if __name__ == "__main__":
    env = {k: int(v) for k, v in (a.split("=", 1) for a in sys.argv[1:])}
    bad = 0
    for seed, snr, md in ((11, -20.0, 0), (12, -27.0, 0), (13, -24.0, 4)):
        fr = G.synth.make_frames(96, seed=seed, snr_db=snr, maxdrift=float(md))
        a = run({}, fr, md)
        b = run(env, fr, md)
        same = a == b
        bad += 0 if same else 1
        print("seed %d snr %.0f maxdrift %d: %s (%d bytes)" % (seed, snr, md, "identical" if same else "DIFFERENT", len(a)))
    sys.exit(1 if bad else 0)
