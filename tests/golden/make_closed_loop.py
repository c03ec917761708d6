#!/usr/bin/env python3
"""Makes tests/golden/closed_loop_int16.npz: the INPUT of the reference's closed-loop demo
(README.md:61, examples/WaveFilePlusNoiseDecode.grc) as data.

The flowgraph adds two repeating wav sources at 12 kS/s,
    blocks_wavfile_source_0  examples/test_1500_Hz.wav   x 0.1   (blocks_multiply_const_vxx_1, grc:802)
    blocks_wavfile_source_1  examples/whales_12000sps.wav x 1    (blocks_multiply_const_vxx_0, grc:751;
                             nchan = 1: channel 0 of the two-channel file)
and feeds the sum to the receiver chain.  Committed here: the int16 samples of the two files (mono,
channel 0 of the second) so that a test can form the same sum; the known answer is the message the
sender was generated from, `VE3EMB FN25 30` (README.md:37).

Run in the build container (needs /root/reference):  python tests/golden/make_closed_loop.py
"""
import os

import numpy as np
import scipy.io.wavfile as wav

REF = "/root/reference/examples"
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "closed_loop_int16.npz")

r1, tx = wav.read(os.path.join(REF, "test_1500_Hz.wav"))
r2, wh = wav.read(os.path.join(REF, "whales_12000sps.wav"))
assert r1 == r2 == 12000 and tx.dtype == np.int16 and wh.dtype == np.int16 and wh.ndim == 2
np.savez_compressed(OUT, tx=tx, whales=np.ascontiguousarray(wh[:, 0]), rate=np.int32(12000),
                    tx_gain=np.float32(0.1), whales_gain=np.float32(1.0))
print(OUT, os.path.getsize(OUT), "bytes; tx", tx.shape, "whales", wh[:, 0].shape)
