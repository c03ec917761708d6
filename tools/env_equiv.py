"""Byte-equality of the staged schedule under environment switches (GPU box):
    python3 tools/env_equiv.py UWSPR_K4_FSPLIT=1 [MORE=1 ...]
runs uwspr_pipeline_batch + uwspr_demod_batch on seeded frames with the default switches and with the given
ones (fresh contexts, staged form) and compares every output byte."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import gr_uwspr_amd as G  # noqa: E402


def run(env, frames, maxdrift):
    keep = {k: os.environ.get(k) for k in env}
    os.environ.update(env)
    os.environ.setdefault("UWSPR_OPTIONS", "sched=0")
    c = G.Context(maxdrift=maxdrift)
    try:
        cands, out = c.pipeline_batch(frames, max_per_frame=3)
        out2 = c.demod_batch(frames, cands, max_per_frame=3)
        return b"".join(x.tobytes() for x in cands) + out.tobytes() + out2.tobytes()
    finally:
        c.close()
        for k, v in keep.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v


if __name__ == "__main__":
    env = dict(a.split("=", 1) for a in sys.argv[1:])
    bad = 0
    for seed, snr, md in ((11, -20.0, 0), (12, -27.0, 0), (13, -24.0, 4)):
        fr = G.synth.make_frames(96, seed=seed, snr_db=snr, maxdrift=float(md))
        a = run({}, fr, md)
        b = run(env, fr, md)
        same = a == b
        bad += 0 if same else 1
        print("seed %d snr %.0f maxdrift %d: %s (%d bytes)" % (seed, snr, md, "identical" if same else "DIFFERENT", len(a)))
    sys.exit(1 if bad else 0)
