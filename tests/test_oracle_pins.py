"""Pins the CPU oracle (oracle/uwspr_oracle.c) before it is trusted as a checker.

  * SLM: bit-exact against the REAL reference (oracle/_ref built from
    lib/slm.cc, or the committed table generated from it) and lib/slm_qa.cc.
  * FDR + sync_and_demodulate on examples/VE3EMB.c2: the known answers the
    survey recorded from the real reference (tests/golden/ve3emb_known.json):
    candidate fields, sync to 9 digits, the decoded blob and message.
  * FFT: against a float64 DFT (FFTW3f is third-party and unpinned).
  * committed oracle vectors are reproduced bit for bit (determinism).
"""
import json
import os

import numpy as np
import pytest

from conftest import GOLDEN


@pytest.fixture(scope="module")
def known():
    return json.load(open(os.path.join(GOLDEN, "ve3emb_known.json")))


def test_pr3_matches_generator_table(oracle, G):
    assert (oracle.pr3() == G.synth.PR3).all()
    assert int(oracle.pr3().sum()) == 63 and oracle.pr3().size == 162


def test_slm_against_real_reference_table(oracle):
    """lib/slm.cc:36-73 -- every (instance, cf, t) value bit-exact."""
    inst = np.load(os.path.join(GOLDEN, "slm_instances_ref.npy"))
    tab = np.load(os.path.join(GOLDEN, "slm_table_ref.npy"))
    mine = oracle.slm_instances()
    assert len(mine) == 125 == len(inst)
    for i, (V1, V2, p1, p2) in enumerate(mine):
        assert (V1, V2, p1, p2) == tuple(inst[i])          # generator order, slm.cc:76-116
        for ci, cf in enumerate((1500.0, 3000.0)):
            for t in range(120):
                a = np.float32(oracle.slm_frequency_drift(V1, V2, p1, p2, cf, float(t)))
                assert a.tobytes() == tab[i, ci, t].tobytes(), (i, cf, t)


def test_slm_against_live_reference(oracle):
    R = oracle.ref()
    if R is None:
        pytest.skip("oracle/_ref not built (no /root/reference here)")
    rng = np.random.default_rng(3)
    for _ in range(2000):
        V1, V2 = rng.uniform(-3, 3, 2)
        p1, p2 = int(rng.integers(-50, 50)), int(rng.integers(0, 900))
        t = float(rng.integers(0, 120))
        a = np.float32(oracle.slm_frequency_drift(V1, V2, p1, p2, 1500.0, t))
        b = np.float32(R.ref_slm_frequency_drift(V1, V2, p1, p2, 1500.0, t))
        assert a.tobytes() == b.tobytes()


def test_slm_qa_known_answer(oracle, known):
    """lib/slm_qa.cc:31-56: V=(1,-2), p=(0,50), t=0..119, printed with %g-style 6 digits."""
    txt = open(os.path.join(GOLDEN, "slm_qa.txt")).read().split("\n")[0].split()
    vals = [oracle.slm_frequency_drift(1.0, -2.0, 0, 50, 1500.0, float(t)) for t in range(120)]
    assert len(txt) == 120
    for t, (s, v) in enumerate(zip(txt, vals)):
        assert "%g" % v == s, (t, s, v)
    assert vals[known["slm_qa_zero_at_t"]] == 0.0
    assert abs(vals[-1] - known["slm_qa_tail"]) < 1e-5


def test_fdr_constants(oracle, known):
    f = oracle.FDR()
    d = known["derived"]
    assert (f.f.size, f.f.m, f.f.n, f.f.hpbm, f.f.finpb, f.f.noiseidx) == \
        (d["size"], d["m"], d["n"], d["hpbm"], d["finpb"], d["noiseidx"])
    assert f.f.df == d["df"] and abs(f.f.min_snr - d["min_snr"]) < 1e-8
    w = f.window()
    assert w[0] == 0.0 and abs(w[255] - np.sin(np.pi * 255 / 511)) < 1e-7  # divisor 511, cc:104


def test_fdr_rejects_what_the_reference_cannot_run(oracle):
    with pytest.raises(ValueError):
        oracle.FDR(halfbandwidth=188)        # > fs/2: reference exit(-1)s, FDR_impl.cc:85-90
    with pytest.raises(ValueError):
        oracle.FDR(halfbandwidth=187)        # GRC default: reference reads out of bounds


def test_ve3emb_fdr_known_answer(oracle, ve3emb, known):
    nz = np.nonzero(ve3emb[:, 0])[0]
    assert [int(nz[0]), int(nz[-1])] == known["nonzero_sample_span"]
    c = oracle.FDR().transform(ve3emb)
    assert len(c) == known["npk"]
    c = c[0]
    assert int(c["m_type"]) == known["m_type"] and int(c["shift"]) == known["shift"]
    assert float(c["freq"]) == known["freq"]
    assert (float(c["V1"]), float(c["V2"]), int(c["p1"]), int(c["p2"])) == \
        (known["V1"], known["V2"], known["p1"], known["p2"])
    assert "%.9f" % c["sync"] == "%.9f" % known["sync"]
    assert min(abs(float(c["snr"]) - s) for s in known["snr_any_of"]) < 5e-6  # FFT-dependent
    k2 = known["threshold_1e6"]
    c2 = oracle.FDR(threshold=1000000).transform(ve3emb)[0]
    assert int(c2["m_type"]) == k2["m_type"] and float(c2["freq"]) == k2["freq"]
    assert int(c2["shift"]) == k2["shift"] and "%.9f" % c2["sync"] == "%.9f" % k2["sync"]
    lin_drift = np.frombuffer(c2.tobytes()[24:28], np.float32)[0]
    assert lin_drift == k2["drift"]


def test_ve3emb_decodes_through_real_reference_fano(oracle, ve3emb, known):
    """oracle FDR -> oracle schedule -> soft symbols -> REAL lib/Fano.cc -> REAL unpk_."""
    if oracle.ref() is None:
        pytest.skip("oracle/_ref not built")
    c = oracle.FDR().transform(ve3emb)[0]
    d = oracle.demod_candidate(c, 1500, ve3emb)
    assert d["worth_a_try"] == 1
    blob = None
    for idt in range(17):
        if d["jig_sync"][idt] > 0.12 and d["jig_rms"][idt] > 40.625:
            rc, data, _, _ = oracle.ref_fano_decode(oracle.deinterleave(d["symbols"][idt]))
            if rc == 0:
                blob = data[:7]
                break
    assert blob is not None and bytes(blob).hex() == known["blob_hex"]
    msg = [int(x) - 256 if x > 127 else int(x) for x in blob]
    assert oracle.ref_unpk(msg) == known["message"]


def test_nonlinear_t0_equivalence(oracle, ve3emb):
    """with t defined as 0, nonlinear V=(-1,-1),p=(0,650) is the
    linear hypothesis shifted by +1.0 Hz, bit-identical sync and symbols."""
    nl = np.zeros(1, oracle.CAND_DTYPE)[0]
    nl["m_type"] = 1; nl["V1"] = -1.0; nl["V2"] = -1.0; nl["p2"] = 650
    lin = np.zeros(1, oracle.CAND_DTYPE)[0]
    s1, _, _, y1 = oracle.sync_and_demodulate(nl, 1500, ve3emb, -1.0, 0, 0, 0.0, 368, 0, 0, 1, 0.0, 50, 2)
    s2, _, _, y2 = oracle.sync_and_demodulate(lin, 1500, ve3emb, 0.0, 0, 0, 0.0, 368, 0, 0, 1, 0.0, 50, 2)
    assert np.float32(s1).tobytes() == np.float32(s2).tobytes() and (y1 == y2).all()


def test_fft_against_float64_dft(oracle, ve3emb):
    """FFTW3f is unpinned third-party arithmetic: any DFT must agree to fp32 rounding."""
    rng = np.random.default_rng(5)
    iq = (ve3emb + 0.5 * rng.standard_normal(ve3emb.shape)).astype(np.float32)
    f = oracle.FDR()
    ps = f.spectrogram(iq)
    w = f.window().astype(np.float64)
    x = iq[:, 0].astype(np.float64) + 1j * iq[:, 1].astype(np.float64)
    for i in (0, 1, 173, 347):
        seg = (x[128 * i:128 * i + 512].real.astype(np.float32) * w.astype(np.float32)).astype(np.float64) + \
            1j * (x[128 * i:128 * i + 512].imag.astype(np.float32) * w.astype(np.float32)).astype(np.float64)
        ref = np.abs(np.fft.fftshift(np.fft.fft(seg))) ** 2
        err = np.abs(ps[i] - ref) / (ref.max())
        assert err.max() < 1e-5


def test_committed_oracle_vectors_reproduce(oracle, G):
    v = np.load(os.path.join(GOLDEN, "oracle_vectors.npz"))
    frames = G.synth.make_frames(4, seed=0xC0FFEE, snr_db=-20.0)
    f = oracle.FDR()
    for b in range(4):
        c = f.transform(frames[b])
        assert len(c) == v["npk"][b]
        assert c.tobytes() == v["cands"][b, :len(c)].astype(oracle.CAND_DTYPE).tobytes()
    d = oracle.demod_candidate(v["cands"][1, 0], 1500, frames[1])
    assert (d["symbols"] == v["demod_symbols"][1]).all()
    assert d["jig_sync"].tobytes() == v["demod_jig_sync"][1].tobytes()
