"""The FDR restatement run with a SECOND, independently ordered binary32 FFT.

The reference takes its spectrogram from FFTW3f (lib/FDR_impl.cc:123-132, 244), third-party and unpinned: `ps` can
match any other implementation only to binary32 FFT tolerance.  The oracle's own FFT is a radix-2
decimation-in-time transform; here the same windowed rows go through scipy's pocketfft in complex64 (mixed radix,
another operation order) and everything downstream of `ps` -- FDR_impl.cc:257-409 -- runs unchanged through the
oracle.  Comparing the two candidate lists says which outputs survive a different FFT."""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
for p in (os.path.join(HERE, ".."), os.path.join(HERE, "..", "oracle")):
    if p not in sys.path:
        sys.path.insert(0, p)

INT_FIELDS = ("m_type", "shift", "p1", "p2")


def spectrogram_pocketfft(f, iq):
    """FDR_impl.cc:222-254 with scipy.fft (complex64) in place of fftwf_execute."""
    import scipy.fft
    fl, n, size = f.f.fl, f.f.n, f.f.size
    x = np.ascontiguousarray(iq, np.float32).reshape(fl, 2)
    w = f.window()
    idx = (np.arange(n)[:, None] * (size // 4) + np.arange(size)[None, :])
    rows = (x[idx, 0].astype(np.float64) * w[None, :].astype(np.float64)).astype(np.float32) \
        + 1j * (x[idx, 1].astype(np.float64) * w[None, :].astype(np.float64)).astype(np.float32)
    F = scipy.fft.fft(rows.astype(np.complex64), axis=1)
    assert F.dtype == np.complex64
    re, im = F.real.astype(np.float32), F.imag.astype(np.float32)
    ps = re * re + im * im                                   # cc:250 (binary32)
    return np.ascontiguousarray(np.fft.fftshift(ps, axes=1))  # cc:246-253: k = j + 256 mod 512


def fdr_from_ps(f, ps):
    psavg, smraw, smspec, noise = f.stats(ps)
    cands = f.peaks(smspec)
    out = []
    for c in cands:
        r, _ = f.search(ps, c)
        out.append(r)
    return out


def compare_frame(args):
    """-> dict of counts for one frame (worker of the process pool)"""
    import oracle_py as O
    kind, seed, snr, kw = args
    import gr_uwspr_amd as G
    if kind == "noise":
        iq = (0.5 * np.random.default_rng(seed).standard_normal((45000, 2))).astype(np.float32)
    else:
        iq = G.synth.make_frames(1, seed=seed, snr_db=snr, maxdrift=float(kw.get("maxdrift", 0)))[0]
    f = O.FDR(**kw)
    return compare_iq(f, iq)


def compare_iq(f, iq):
    a = fdr_from_ps(f, f.spectrogram(iq))
    b = fdr_from_ps(f, spectrogram_pocketfft(f, iq))
    r = dict(frames=1, cands=len(a), npk_diff=int(len(a) != len(b)), order_or_freq_diff=0, int_field_diff=0,
             drift_diff=0, same=0, sync_rel=0.0, snr_rel=0.0, ps_rel=0.0)
    if len(a) != len(b):
        return r
    for x, y in zip(a, b):
        if np.float32(x["freq"]).tobytes() != np.float32(y["freq"]).tobytes():
            r["order_or_freq_diff"] += 1
            continue
        ints = all(int(x[k]) == int(y[k]) for k in ("m_type", "shift"))
        if ints and int(x["m_type"]) == 1:
            ints = all(x[k] == y[k] for k in ("V1", "V2", "p1", "p2"))
        drift = int(x["m_type"]) == 0 and ints and x.tobytes()[24:28] != y.tobytes()[24:28]
        r["int_field_diff"] += int(not ints)
        r["drift_diff"] += int(drift)
        if ints and not drift:
            r["same"] += 1
            if float(x["sync"]) != 0:
                r["sync_rel"] = max(r["sync_rel"], abs(float(y["sync"]) - float(x["sync"])) / abs(float(x["sync"])))
            if float(x["snr"]) != 0:
                r["snr_rel"] = max(r["snr_rel"], abs(float(y["snr"]) - float(x["snr"])) / abs(float(x["snr"])))
    return r


def merge(rs):
    tot = {}
    for r in rs:
        for k, v in r.items():
            if k.endswith("_rel"):
                tot[k] = max(tot.get(k, 0.0), v)
            else:
                tot[k] = tot.get(k, 0) + v
    return tot


def workload(n_per_level=300):
    """2 100 frames at the flowgraph defaults (-18 .. -31 dB and noise only) + 300 at hbw 40 / maxdrift 2"""
    jobs = []
    for k, snr in enumerate((-18.0, -22.0, -25.0, -27.0, -29.0, -31.0)):
        jobs += [("sig", 0xF0F0 + 1000 * k + i, snr, {}) for i in range(n_per_level)]
    jobs += [("noise", 5000 + i, None, {}) for i in range(n_per_level)]
    jobs += [("sig", 0xABC0 + i, -24.0, {"halfbandwidth": 40, "maxdrift": 2}) for i in range(n_per_level)]
    return jobs
