"""FFTW3f is third-party and unpinned, so the spectrogram is "parity unpinned" at bit level.  This test
turns that into a number: the FDR restatement (oracle) run on the spectrogram of a SECOND, independently ordered
binary32 FFT (scipy's pocketfft in complex64 against the oracle's radix-2 DIT) over 2 400 seeded frames from -18 dB to
-31 dB, noise only, and a wider band with drifting signals (profiles/r04_fft_gap.json, tools/fft_gap.py):

  invariant (0 differences in 2 099 candidates): the number of candidates, their order, freq, m_type, shift, the
      straight-line parameters (V1, V2, p1, p2) and the linear drift -- everything sync_and_demodulate reads from a
      candidate (sync_and_demodulate_impl.cc:404-407, 389), hence every soft-symbol byte and every decode;
  within tolerance: `sync` (<= 1e-5 relative: measured 5.2e-7), `snr` (<= 1e-4 relative: measured 2.0e-5; it is
      10 log10 of a value close to its floor).
A flip is possible in principle (a near-tie in the running-best rule, FDR_impl.cc:360, 392); none occurs here."""
import os
from concurrent.futures import ProcessPoolExecutor

import numpy as np

import fft_gap_common as F


def test_candidates_survive_a_second_fft_on_the_reference_fixture(oracle):
    iq = oracle.read_c2(os.path.join(os.path.dirname(__file__), "golden", "VE3EMB.c2"))
    f = oracle.FDR()
    rng = np.random.default_rng(12)
    for sigma in (0.0, 2.0, 6.0):                       # the clean frame and two noisy copies
        x = (iq + sigma * rng.standard_normal(iq.shape)).astype(np.float32)
        r = F.compare_iq(f, x)
        assert r["cands"] >= 1 and r["same"] == r["cands"], (sigma, r)
        assert r["sync_rel"] <= 1e-5 and r["snr_rel"] <= 1e-4, (sigma, r)
    # the two transforms do differ: this is not the same FFT twice
    ps_a, ps_b = f.spectrogram(iq), F.spectrogram_pocketfft(f, iq)
    assert ps_a.tobytes() != ps_b.tobytes()
    assert np.max(np.abs(ps_a - ps_b)) <= 1e-5 * np.max(ps_a)


def test_candidates_survive_a_second_fft_on_2400_seeded_frames():
    jobs = F.workload(300)
    with ProcessPoolExecutor(max_workers=min(8, os.cpu_count() or 1)) as ex:
        tot = F.merge(list(ex.map(F.compare_frame, jobs, chunksize=8)))
    assert tot["frames"] == 2400 and tot["cands"] >= 2000, tot
    # invariant: count, order, freq, type, shift, model parameters, drift
    assert tot["npk_diff"] == 0 and tot["order_or_freq_diff"] == 0 and tot["int_field_diff"] == 0 and tot["drift_diff"] == 0, tot
    assert tot["same"] == tot["cands"], tot
    # floats: within BASELINE's tolerance (sync), snr a decade wider (see the module text)
    assert tot["sync_rel"] <= 1e-5 and tot["snr_rel"] <= 1e-4, tot
