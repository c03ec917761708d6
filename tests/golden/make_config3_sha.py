#!/usr/bin/env python3
"""Full-size BASELINE configs[2] digest: 1024 seeded frames x 200 (freq, lag, drift)
hypotheses, sync + 162 soft symbols each, computed with the CPU oracle
(oracle/uwspr_oracle.c, already pinned bit-exact on the committed subset).
Writes tests/golden/config3_digest.json: sha256 of the 204800x162 symbol bytes and of
the 204800 binary32 metrics (hypothesis order of gr_uwspr_amd.sweep.sweep_grid), plus
every 97th metric for a tolerance check.  ~1 minute on 8 threads."""
import hashlib
import json
import os
import sys
from concurrent.futures import ThreadPoolExecutor

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import oracle_py as O  # noqa: E402
import gr_uwspr_amd as G  # noqa: E402

B, SEED, SNR = 1024, 0xC3C3, -20.0


def centres(meta):
    c = np.zeros(len(meta), O.CAND_DTYPE)
    c["freq"] = np.array([np.float32(m["f_off"]) for m in meta], np.float32)
    c["shift"] = 368
    return c


def main():
    O.build(ref=False)
    lin = np.zeros(1, O.CAND_DTYPE)[0]

    def one_frame(b):
        fr, meta = G.synth.make_frames(1, seed=SEED, snr_db=SNR, first=b, return_meta=True)
        hy = G.sweep.sweep_grid(centres(meta))
        sy = np.zeros(200, np.float32)
        sm = np.zeros((200, 162), np.uint8)
        for q, h in enumerate(hy):
            s, _, _, y = O.sync_and_demodulate(lin, 1500, fr[0], float(h["f0"]), 0, 0, 0.0,
                                               int(h["lag"]), 0, 0, 1, float(h["drift"]), 50, 2)
            sy[q] = s
            sm[q] = y
        return sy, sm

    with ThreadPoolExecutor(8) as ex:
        res = list(ex.map(one_frame, range(B)))
    sync = np.concatenate([r[0] for r in res])
    sym = np.concatenate([r[1] for r in res])
    out = {"frames": B, "seed": SEED, "snr_db": SNR, "hypotheses": int(sync.size),
           "symbols_sha256": hashlib.sha256(sym.tobytes()).hexdigest(),
           "sync_sha256": hashlib.sha256(sync.tobytes()).hexdigest(),
           "sync_every_97th": [float(x) for x in sync[::97]]}
    json.dump(out, open(os.path.join(HERE, "config3_digest.json"), "w"))
    print(out["symbols_sha256"], out["sync_sha256"], sync.size)


if __name__ == "__main__":
    main()
