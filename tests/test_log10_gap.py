"""FDR_impl.cc:303 `10*log10(smspec[j])` -- g++ resolves it to log10f, and libm's log10f is not correctly rounded: its
last bit is the host libm's.  Rounds 1-4 took log10 in binary64 on the device and compared `snr` at 1e-5 (8 % of the
arguments differ in the last bit, and the reference sorts its candidates by that value).  Round 5: the kernel restates
glibc 2.35's log10f -- the libm of this image, where the oracle and oracle/_ref run -- operation for operation
(k2_spectrum.hip: log10f_glibc235; the same restatement in C: oracle/uwspr_oracle.c: orc_log10f_glibc235), and

  * here, on the CPU: the restatement equals the host's log10f for EVERY positive binary32 argument, zero, infinity and
    the subnormals included (2^31 - 2^23 + 1 patterns), in both forms libm can select on x86-64 (multiply-adds of logf
    fused or not -- they never differ after the final rounding to binary32);
  * on the GPU: the device function equals the host's log10f on every argument the normalised spectrum can hold
    (2^-6 .. 2^14: 168 M patterns), on a stride through the whole positive range, and on the special values;
  * test_gpu_parity.py's cand_equal compares the `snr` bytes of every candidate.

Round 6: the PINNED PLATFORM is glibc 2.35 / x86-64, and the oracle's FDR calls the restatement, not the host's log10f
(uwspr_oracle.c: orc_fdr_peaks).  A host whose libm has another log10f (glibc >= 2.40's is correctly rounded) changes
neither the oracle nor the product; on such a host the first test -- the statement "this host's log10f is glibc 2.35's"
-- is reported as an expected failure carrying the C library's version: the kernel's `snr` would differ from a
reference built against THAT libm in the last bit on a few per cent of the values, inside BASELINE's 1e-5."""
import os
import ctypes as C
import struct
from concurrent.futures import ThreadPoolExecutor

import numpy as np
import pytest

from oracle import oracle_py as O


def _bits(x):
    return struct.unpack("<I", struct.pack("<f", x))[0]


def test_restated_log10f_is_this_libms_for_every_positive_binary32():
    L = O.lib()
    lo, hi = 0, 0x7F800001                       # +0, every subnormal and normal value, +inf
    nt = 8
    edges = [lo + (hi - lo) * k // (4 * nt) for k in range(4 * nt + 1)]

    def walk(args):
        a, b, fma = args
        first = C.c_uint32(0)
        bad = L.orc_log10f_walk(a, b, 1, fma, C.byref(first))
        return bad, first.value
    jobs = [(edges[k], edges[k + 1], fma) for fma in (0, 1) for k in range(4 * nt)]
    with ThreadPoolExecutor(nt) as ex:            # (ctypes releases the GIL: 2 x 2.1 G evaluations on the host's cores)
        res = list(ex.map(walk, jobs))
    bad = sum(r[0] for r in res)
    if bad:
        try:
            libc = os.confstr("CS_GNU_LIBC_VERSION")
        except (ValueError, OSError):
            libc = "unknown C library"
        pytest.xfail("this host's log10f (%s) is not glibc 2.35's on %d of 2 x %d binary32 arguments (first: %s): the oracle "
                     "and the kernel stay on the pinned platform's log10f (glibc 2.35 / x86-64)"
                     % (libc, bad, hi - lo, [(hex(r[1]), j[2]) for j, r in zip(jobs, res) if r[0]][:2]))
    # negative arguments and NaN: NaN out (payloads not compared)
    for x in (-1.0, -0.0, float("nan"), -float("inf")):
        want = np.float32(np.log10(np.float32(x))) if x != -0.0 else np.float32(-np.inf)
        got = np.float32(L.orc_log10f_glibc235(x, 0))
        assert (np.isnan(want) and np.isnan(got)) or want == got, x


@pytest.mark.gpu
def test_device_snr_is_the_pinned_log10f_to_the_bit(G):
    """The device's 10 * log10f against cc:303 as the oracle has it (orc_snr_db -> the glibc 2.35 restatement; on the pinned
    platform that IS the host's log10f: the test above)."""
    import torch
    L = O.lib()
    c = G.Context()
    fp = C.POINTER(C.c_float)
    total = bad = 0
    try:
        def check(bits):
            nonlocal total, bad
            x = bits.view(torch.float32)
            got = c.debug_snr_db(x).cpu().numpy()
            xh = x.cpu().numpy()
            want = np.empty_like(xh)
            L.orc_snr_db(xh.ctypes.data_as(fp), want.ctypes.data_as(fp), xh.size)
            nan = np.isnan(want)
            assert (np.isnan(got) == nan).all()
            d = (got.view(np.uint32) != want.view(np.uint32)) & ~nan
            total += xh.size
            bad += int(d.sum())
            assert not d.any(), (hex(int(xh.view(np.uint32)[np.argmax(d)])), got[np.argmax(d)], want[np.argmax(d)])
        lo, hi = _bits(2.0 ** -6), _bits(2.0 ** 14)          # what the normalised spectrum can hold, every pattern
        step = 1 << 24
        for a in range(lo, hi, step):
            check(torch.arange(a, min(a + step, hi), dtype=torch.int32, device="cuda"))
        # the whole positive range on a stride, the subnormals' ends and the special values
        check(torch.arange(0, 0x7F800001, 251, dtype=torch.int64, device="cuda").to(torch.int32))
        check(torch.tensor([0, 1, 2, 0x007FFFFF, 0x00800000, 0x3F7FFFFF, 0x3F800000, 0x3F800001, 0x7F7FFFFF, 0x7F800000,
                            0x7FC00000, -0x80000000, -0x40800000], dtype=torch.int32, device="cuda"))
    finally:
        c.close()
    print("device 10*log10f against the oracle's (glibc 2.35 restated): %d arguments, %d differ" % (total, bad))
