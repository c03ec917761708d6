"""ctypes doorway to the CPU oracle -- TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import
this module.  The product (gr-uwspr_amd/, include/) never does.

liboracle.so        = oracle/uwspr_oracle.c (our restatement)
_ref/libuwspr_ref.so = the real reference's slm.cc / Fano.cc / helpers.cc built
                       where they lie (oracle/Makefile `ref`), reached through
                       oracle/ref_shim.cc.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
NSYM, NSLM, NK0, NIFR, NJIG = 162, 125, 26, 5, 17


class ModeNonlinear(C.Structure):
    _fields_ = [("V1", C.c_double), ("V2", C.c_double), ("p1", C.c_int32), ("p2", C.c_int32)]


class _U(C.Union):
    _fields_ = [("lin_drift", C.c_float), ("nl", ModeNonlinear)]


class Candidate(C.Structure):
    """lib/candidate_t.h:27-50 (48 bytes)."""
    _anonymous_ = ("u",)
    _fields_ = [("freq", C.c_float), ("snr", C.c_float), ("drift", C.c_float),
                ("sync", C.c_float), ("shift", C.c_int32), ("m_type", C.c_int32),
                ("u", _U)]


assert C.sizeof(Candidate) == 48

CAND_DTYPE = np.dtype([("freq", "<f4"), ("snr", "<f4"), ("drift", "<f4"), ("sync", "<f4"),
                       ("shift", "<i4"), ("m_type", "<i4"), ("V1", "<f8"), ("V2", "<f8"),
                       ("p1", "<i4"), ("p2", "<i4")])
assert CAND_DTYPE.itemsize == 48


class Fdr(C.Structure):
    _fields_ = [("fs", C.c_int), ("fl", C.c_int), ("spb", C.c_int), ("maxdrift", C.c_int),
                ("maxfreqs", C.c_int), ("halfbandwidth", C.c_int), ("cf", C.c_int),
                ("threshold", C.c_float), ("size", C.c_int), ("m", C.c_int), ("hpbm", C.c_int),
                ("n", C.c_int), ("finpb", C.c_int), ("noiseidx", C.c_int), ("df", C.c_float),
                ("min_snr", C.c_float), ("w", C.POINTER(C.c_float)), ("tw", C.POINTER(C.c_float))]


class DemodOut(C.Structure):
    _fields_ = [("f1", C.c_float), ("drift1", C.c_float), ("sync1", C.c_float),
                ("shift1", C.c_int32), ("worth_a_try", C.c_int32),
                ("jig_sync", C.c_float * NJIG), ("jig_rms", C.c_float * NJIG),
                ("jig_shift", C.c_int32 * NJIG), ("symbols", (C.c_ubyte * NSYM) * NJIG)]


def build(ref=True):
    """Compile liboracle.so (and _ref/ when /root/reference exists)."""
    targets = ["all"] + (["ref"] if ref else [])
    subprocess.run(["make", "-s", "-C", _HERE] + targets, check=True)


def _load(path):
    if not os.path.exists(path):
        build()
    return C.CDLL(path)


_lib = None
_ref = None


def lib():
    global _lib
    if _lib is None:
        L = _load(os.path.join(_HERE, "liboracle.so"))
        fp = C.POINTER(C.c_float)
        L.orc_fdr_init.argtypes = [C.POINTER(Fdr)] + [C.c_int] * 8
        L.orc_fdr_init.restype = C.c_int
        L.orc_fdr_free.argtypes = [C.POINTER(Fdr)]
        L.orc_fdr_cell_hyps.argtypes = [C.POINTER(Fdr)]
        L.orc_fdr_spectrogram.argtypes = [C.POINTER(Fdr), fp, fp]
        L.orc_fdr_stats.argtypes = [C.POINTER(Fdr), fp, fp, fp, fp, fp]
        L.orc_fdr_peaks.argtypes = [C.POINTER(Fdr), fp, C.c_void_p]
        L.orc_fdr_peaks.restype = C.c_int
        L.orc_fdr_search.argtypes = [C.POINTER(Fdr), fp, C.c_void_p, fp]
        L.orc_fdr_transform.argtypes = [C.POINTER(Fdr), fp, C.c_void_p]
        L.orc_fdr_transform.restype = C.c_int
        L.orc_log10f_walk.argtypes = [C.c_uint32, C.c_uint32, C.c_uint32, C.c_int, C.POINTER(C.c_uint32)]
        L.orc_log10f_walk.restype = C.c_long
        L.orc_log10f_glibc235.argtypes = [C.c_float, C.c_int]
        L.orc_log10f_glibc235.restype = C.c_float
        L.orc_snr_db.argtypes = [fp, fp, C.c_long]
        L.orc_snr_db.restype = None
        L.orc_slm_frequency_drift.argtypes = [C.c_double, C.c_double, C.c_int, C.c_int,
                                              C.c_float, C.c_float]
        L.orc_slm_frequency_drift.restype = C.c_float
        L.orc_slm_generate.argtypes = [C.c_int, C.POINTER(C.c_double), C.POINTER(C.c_double),
                                       C.POINTER(C.c_int), C.POINTER(C.c_int)]
        L.orc_slm_generate.restype = C.c_int
        L.orc_sync_and_demodulate.argtypes = [
            C.c_void_p, C.c_int, fp, fp, C.c_long, C.POINTER(C.c_ubyte), fp, C.c_int, C.c_int,
            C.c_float, C.POINTER(C.c_int), C.c_int, C.c_int, C.c_int, fp, C.c_int, fp, C.c_int]
        L.orc_deinterleave.argtypes = [C.POINTER(C.c_ubyte)]
        L.orc_demod_candidate.argtypes = [C.c_void_p, C.c_int, fp, fp, C.c_long,
                                          C.POINTER(DemodOut)]
        L.orc_symbols_rms.argtypes = [C.POINTER(C.c_ubyte)]
        L.orc_symbols_rms.restype = C.c_float
        L.orc_pr3.restype = C.POINTER(C.c_ubyte)
        _lib = L
    return _lib


def ref():
    """The real reference objects (slm / Fano / helpers); None if not built."""
    global _ref
    if _ref is None:
        path = os.path.join(_HERE, "_ref", "libuwspr_ref.so")
        if not os.path.exists(path):
            if os.path.isdir("/root/reference/lib"):
                build()
            else:
                return None
        R = C.CDLL(path)
        R.ref_slm_frequency_drift.argtypes = [C.c_double, C.c_double, C.c_int, C.c_int,
                                              C.c_float, C.c_float]
        R.ref_slm_frequency_drift.restype = C.c_float
        R.ref_slm_generate_all.argtypes = [C.POINTER(C.c_double), C.POINTER(C.c_double),
                                           C.POINTER(C.c_int), C.POINTER(C.c_int), C.c_int]
        R.ref_slm_generate_all.restype = C.c_int
        R.ref_fano_mettab.argtypes = [C.POINTER(C.c_int)]
        R.ref_fano_encode.argtypes = [C.POINTER(C.c_ubyte), C.POINTER(C.c_ubyte), C.c_uint]
        R.ref_fano_decode.argtypes = [C.POINTER(C.c_uint), C.POINTER(C.c_uint),
                                      C.POINTER(C.c_uint), C.POINTER(C.c_ubyte),
                                      C.POINTER(C.c_ubyte), C.c_uint, C.c_int, C.c_uint]
        R.ref_fano_decode.restype = C.c_int
        R.ref_unpk.argtypes = [C.POINTER(C.c_byte), C.c_char_p]
        R.ref_unpk.restype = C.c_int
        _ref = R
    return _ref


def _fp(a):
    return a.ctypes.data_as(C.POINTER(C.c_float))


def pr3():
    return np.ctypeslib.as_array(lib().orc_pr3(), shape=(NSYM,)).copy()


class FDR:
    """Oracle twin of gr::uwspr::FDR (include/uwspr/FDR.h:49-50)."""

    def __init__(self, fs=375, fl=45000, spb=256, maxdrift=0, maxfreqs=200,
                 halfbandwidth=10, cf=1500, threshold=10):
        self.f = Fdr()
        rc = lib().orc_fdr_init(C.byref(self.f), fs, fl, spb, maxdrift, maxfreqs,
                                halfbandwidth, cf, threshold)
        if rc != 0:
            raise ValueError("orc_fdr_init failed: %d" % rc)
        self.maxfreqs = maxfreqs

    def __del__(self):
        try:
            lib().orc_fdr_free(C.byref(self.f))
        except Exception:
            pass

    @property
    def cell_hyps(self):
        return lib().orc_fdr_cell_hyps(C.byref(self.f))

    def window(self):
        return np.ctypeslib.as_array(self.f.w, shape=(self.f.size,)).copy()

    def spectrogram(self, iq):
        iq = np.ascontiguousarray(iq, dtype=np.float32).reshape(-1)
        assert iq.size == 2 * self.f.fl
        ps = np.empty((self.f.n, self.f.size), dtype=np.float32)
        lib().orc_fdr_spectrogram(C.byref(self.f), _fp(iq), _fp(ps))
        return ps

    def stats(self, ps):
        ps = np.ascontiguousarray(ps, dtype=np.float32)
        psavg = np.empty(self.f.size, np.float32)
        smraw = np.empty(self.f.finpb, np.float32)
        smspec = np.empty(self.f.finpb, np.float32)
        noise = C.c_float()
        lib().orc_fdr_stats(C.byref(self.f), _fp(ps), _fp(psavg), _fp(smraw), _fp(smspec),
                            C.byref(noise))
        return psavg, smraw, smspec, noise.value

    def peaks(self, smspec):
        smspec = np.ascontiguousarray(smspec, dtype=np.float32)
        cands = np.zeros(self.maxfreqs, CAND_DTYPE)
        npk = lib().orc_fdr_peaks(C.byref(self.f), _fp(smspec), cands.ctypes.data)
        return cands[:npk].copy()

    def search(self, ps, cand, want_grid=False):
        """cand: 1-element CAND_DTYPE array (freq, snr set). Returns (cand, grid)."""
        ps = np.ascontiguousarray(ps, dtype=np.float32)
        c = np.array(cand, dtype=CAND_DTYPE).reshape(1).copy()
        grid = None
        gp = None
        if want_grid:
            grid = np.empty((NIFR, NK0, self.cell_hyps), np.float32)
            gp = _fp(grid)
        lib().orc_fdr_search(C.byref(self.f), _fp(ps), c.ctypes.data, gp)
        return c[0], grid

    def transform(self, iq):
        iq = np.ascontiguousarray(iq, dtype=np.float32).reshape(-1)
        assert iq.size == 2 * self.f.fl
        cands = np.zeros(self.maxfreqs, CAND_DTYPE)
        npk = lib().orc_fdr_transform(C.byref(self.f), _fp(iq), cands.ctypes.data)
        return cands[:npk].copy()


def slm_frequency_drift(V1, V2, p1, p2, cf, t):
    return lib().orc_slm_frequency_drift(V1, V2, p1, p2, cf, t)


def slm_instances():
    out = []
    for i in range(NSLM + 1):
        V1, V2, p1, p2 = C.c_double(), C.c_double(), C.c_int(), C.c_int()
        if not lib().orc_slm_generate(i, C.byref(V1), C.byref(V2), C.byref(p1), C.byref(p2)):
            break
        out.append((V1.value, V2.value, p1.value, p2.value))
    return out


# sync_and_demodulate_impl.cc:92: `npoints = 45000`, the sample bound every call of the fine search is
# given (cc:413-465) WHATEVER the frame length fl is: with fl > 45000 the samples from 45000 on are
# ignored by the fine search (the spectrogram, FDR_impl.cc:118, still uses them).  With fl < 45000 the
# reference would index past its fl-sized arrays (cc:340: `float idat[fl]`): defined here as np = fl.
NPOINTS = 45000


def npoints(nsamples):
    return min(NPOINTS, int(nsamples))


def sync_and_demodulate(cand, cf, iq, f1, ifmin, ifmax, fstep, shift1, lagmin, lagmax, lagstep,
                        drift1, symfac, mode, np_points=None):
    """Argument-for-argument twin of sync_and_demodulate_impl.cc:126 (np_points = the `np` argument;
    default: what demodulate() passes, cc:92).  Returns (sync, shift1, f1, symbols[162])."""
    iq = np.asarray(iq, dtype=np.float32).reshape(-1, 2)
    idat = np.ascontiguousarray(iq[:, 0])
    qdat = np.ascontiguousarray(iq[:, 1])
    c = np.array(cand, dtype=CAND_DTYPE).reshape(1).copy()
    symbols = np.zeros(NSYM, np.uint8)
    f1c, sh, dr, sy = C.c_float(f1), C.c_int(shift1), C.c_float(drift1), C.c_float(0)
    lib().orc_sync_and_demodulate(c.ctypes.data, cf, _fp(idat), _fp(qdat),
                                  np_points or npoints(idat.size),
                                  symbols.ctypes.data_as(C.POINTER(C.c_ubyte)), C.byref(f1c),
                                  ifmin, ifmax, fstep, C.byref(sh), lagmin, lagmax, lagstep,
                                  C.byref(dr), symfac, C.byref(sy), mode)
    return sy.value, sh.value, f1c.value, symbols


def demod_candidate(cand, cf, iq, np_points=None):
    """The per-candidate body of demodulate(), cc:403-482, with the reference's npoints (cc:92)."""
    iq = np.asarray(iq, dtype=np.float32).reshape(-1, 2)
    idat = np.ascontiguousarray(iq[:, 0])
    qdat = np.ascontiguousarray(iq[:, 1])
    c = np.array(cand, dtype=CAND_DTYPE).reshape(1).copy()
    out = DemodOut()
    lib().orc_demod_candidate(c.ctypes.data, cf, _fp(idat), _fp(qdat), np_points or npoints(idat.size), C.byref(out))
    return {
        "f1": out.f1, "drift1": out.drift1, "sync1": out.sync1, "shift1": out.shift1,
        "worth_a_try": out.worth_a_try,
        "jig_sync": np.array(out.jig_sync, np.float32),
        "jig_rms": np.array(out.jig_rms, np.float32),
        "jig_shift": np.array(out.jig_shift, np.int32),
        "symbols": np.frombuffer(bytes(out.symbols), np.uint8).reshape(NJIG, NSYM).copy(),
    }


def cand_diff(a, e):
    """Fields of a GPU candidate `a` that differ from the oracle's `e` (FDR.transform): what the PDU carries
    (FDR_impl.cc:414-455), binary32 fields by their bytes.  -> list of field names."""
    bad = [k for k in ("m_type", "shift") if int(a[k]) != int(e[k])]
    bad += [k for k in ("freq", "sync", "snr") if np.float32(a[k]).tobytes() != np.float32(e[k]).tobytes()]
    if int(e["m_type"]) == 1:
        bad += [k for k in ("V1", "V2", "p1", "p2") if a[k] != e[k]]
    elif a.tobytes()[24:28] != e.tobytes()[24:28]:      # m_linear.drift
        bad.append("drift")
    return bad


def record_diff(o, d):
    """Fields of a GPU record `o` (a `uwspr_demod_out` element) that differ from the oracle's `d` = demod_candidate(...):
    integers by value, binary32 fields by their BYTES, the soft-symbol vectors of all 17 tries byte for byte (cc:457-482).
    The one comparison the tests, tools/soak_parity.py and bench.py's `parity_spot_check` share.  -> list of field names."""
    bad = []
    if int(o["worth_a_try"]) != int(d["worth_a_try"]):
        bad.append("worth_a_try")
    if int(o["shift1"]) != int(d["shift1"]):
        bad.append("shift1")
    for k in ("f1", "drift1", "sync1"):
        if np.float32(o[k]).tobytes() != np.float32(d[k]).tobytes():
            bad.append(k)
    if d["worth_a_try"]:
        if not (np.asarray(o["symbols"]) == d["symbols"]).all():
            bad.append("symbols")
        if not (np.asarray(o["jig_shift"]) == d["jig_shift"]).all():
            bad.append("jig_shift")
        for k in ("jig_sync", "jig_rms"):
            if np.asarray(o[k], np.float32).tobytes() != d[k].tobytes():
                bad.append(k)
    return bad


def deinterleave(symbols):
    s = np.array(symbols, dtype=np.uint8).copy()
    lib().orc_deinterleave(s.ctypes.data_as(C.POINTER(C.c_ubyte)))
    return s


def symbols_rms(symbols):
    s = np.ascontiguousarray(symbols, dtype=np.uint8)
    return lib().orc_symbols_rms(s.ctypes.data_as(C.POINTER(C.c_ubyte)))


# ---- real-reference helpers (oracle/_ref) ---------------------------------

def ref_fano_decode(symbols, nbits=81, delta=60, maxcycles=10000):
    """lib/Fano.cc:110 on deinterleaved symbols. Returns (not_decoded, data[11], metric, cycles)."""
    R = ref()
    s = np.array(symbols, dtype=np.uint8).copy()
    data = np.zeros(11, np.uint8)
    metric, cycles, maxnp = C.c_uint(), C.c_uint(), C.c_uint()
    rc = R.ref_fano_decode(C.byref(metric), C.byref(cycles), C.byref(maxnp),
                           data.ctypes.data_as(C.POINTER(C.c_ubyte)),
                           s.ctypes.data_as(C.POINTER(C.c_ubyte)), nbits, delta, maxcycles)
    return rc, data, metric.value, cycles.value


def ref_fano_encode(data_bytes):
    R = ref()
    d = np.array(data_bytes, dtype=np.uint8).copy()
    out = np.zeros(d.size * 16, np.uint8)
    R.ref_fano_encode(out.ctypes.data_as(C.POINTER(C.c_ubyte)),
                      d.ctypes.data_as(C.POINTER(C.c_ubyte)), d.size)
    return out


def ref_mettab():
    out = np.zeros((2, 256), np.int32)
    ref().ref_fano_mettab(out.ctypes.data_as(C.POINTER(C.c_int)))
    return out


def ref_unpk(message7):
    m = np.array(message7, dtype=np.int8).copy()
    buf = C.create_string_buffer(32)
    ref().ref_unpk(m.ctypes.data_as(C.POINTER(C.c_byte)), buf)
    return buf.value.decode("ascii", "replace")


def read_c2(path):
    """.c2 layout per c2file_source_impl.cc:80-96: 14-byte name, int32 type,
    float64 freq, 2*45000 float32 interleaved; Q is negated on load."""
    raw = open(path, "rb").read()
    buf = np.frombuffer(raw, dtype="<f4", count=90000, offset=14 + 4 + 8).reshape(45000, 2).copy()
    buf[:, 1] = -buf[:, 1]
    return buf
