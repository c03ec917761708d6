"""FDR_impl.cc:303 `10*log10(smspec[j])` -- g++ resolves it to log10f; the HIP kernel (k2_spectrum.hip) takes log10 in
binary64 and rounds once.  libm's log10f is not correctly rounded, so the two differ in the last bit for a few per cent
of all binary32 arguments (counted below: whatever this image's libm does), inside the 1e-5 the `snr` field is checked
to.  What could matter is the ORDER of the candidates: the reference sorts them by that value with a strict `<`
(cc:307-318), and a last-bit difference can make or break a tie.  This test measures both: the share of arguments on
which the routes differ, and -- over seeded frames with every local maximum of the smoothed spectrum kept -- whether the
order of the peaks ever depends on the route."""
import ctypes as C
import struct

import numpy as np

from oracle import oracle_py as O
import gr_uwspr_amd as G


def _bits(x):
    return struct.unpack("<I", struct.pack("<f", x))[0]


def test_log10_routes_differ_in_the_last_bit_only():
    L = O.lib()
    first = C.c_uint32(0)
    lo, hi = _bits(2.0 ** -6), _bits(2.0 ** 14)            # smspec after normalisation: 0.1 * min_snr .. a few thousand
    n = (hi - lo + 6) // 7
    bad = L.orc_log10_gap(lo, hi, 7, C.byref(first))
    print("log10f against (float)log10((double)x): %d of %d arguments differ (%.2f %%)" % (bad, n, 100.0 * bad / n))
    assert 0 <= bad < n // 4                               # (a last-bit matter, not a different function)


def _orders(f, smspec):
    """candidate frequencies in the oracle's order (log10f) and in the kernel's (binary64 log10, rounded once)"""
    ca = f.peaks(smspec)                                     # cc:293-319: log10f, bubble sort with strict `<`
    j = np.array([k for k in range(1, len(smspec) - 1) if smspec[k] > smspec[k - 1] and smspec[k] > smspec[k + 1]],
                 dtype=np.int64)[: f.maxfreqs]
    snr_b = np.float32(10) * np.log10(smspec[j].astype(np.float64)).astype(np.float32)
    order_b = sorted(range(len(j)), key=lambda i: -float(snr_b[i]))    # stable = bubble sort with strict `<`
    fb = ((j[order_b] - f.f.hpbm).astype(np.float32) * np.float32(f.f.df)).astype(np.float32)
    assert len(ca) == len(j)
    return ca["freq"].astype(np.float32), fb, snr_b


def test_peak_order_does_not_depend_on_the_log10_route():
    """Real frames hold one or two peaks above min_snr, so the spectra are made here: every second bin a local maximum
    with a value drawn log-uniformly from the range the normalised spectrum takes (min_snr .. 4000)."""
    f = O.FDR(halfbandwidth=80, maxfreqs=200)
    rng = np.random.default_rng(303)
    n = f.f.finpb
    peaks = pairs = ties = changed = 0
    for trial in range(400):
        sm = np.full(n, np.float32(0.1 * 0.19952623), np.float32)
        k = np.arange(1, n - 1, 2)
        sm[k] = np.exp(rng.uniform(np.log(0.2), np.log(4000.0), len(k))).astype(np.float32)
        fa, fb, snr_b = _orders(f, sm)
        peaks += len(fa)
        pairs += len(fa) * (len(fa) - 1) // 2
        ties += len(snr_b) - len(set(snr_b.tolist()))
        changed += int(fa.tobytes() != fb.tobytes())
    print("%d spectra, %d peaks, %d ordered pairs, %d tied values, %d spectra whose order depends on the route"
          % (400, peaks, pairs, ties, changed))
    assert peaks > 20000
    assert changed == 0


def test_peak_order_of_neighbouring_values():
    """The adversarial case: peaks whose values are NEIGHBOURS in binary32 (1..3 ulp apart), where a log10f that is not
    monotonic -- or rounds two neighbours apart that the binary64 route rounds together -- would order them differently.
    Reported, and bounded: this is the one place where the candidate ORDER depends on the C library."""
    f = O.FDR(halfbandwidth=80, maxfreqs=200)
    rng = np.random.default_rng(404)
    n = f.f.finpb
    spectra = changed = 0
    for trial in range(300):
        sm = np.full(n, np.float32(0.1 * 0.19952623), np.float32)
        k = np.arange(1, n - 1, 2)
        base = np.exp(rng.uniform(np.log(0.2), np.log(4000.0), (len(k) + 3) // 4)).astype(np.float32)
        v = np.repeat(base.view(np.uint32), 4)[: len(k)] + rng.integers(0, 4, len(k)).astype(np.uint32)   # groups of 4 neighbours
        sm[k] = rng.permutation(v.view(np.float32))
        fa, fb, _ = _orders(f, sm)
        spectra += 1
        changed += int(fa.tobytes() != fb.tobytes())
    print("neighbouring values: %d of %d spectra ordered differently by the two routes" % (changed, spectra))
    # a property of this libm's log10f, not of the kernel: recorded in DESIGN section 4; nothing to assert beyond sanity
    assert spectra == 300
