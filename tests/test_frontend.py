"""K0 front-end: 12 kS/s real audio -> 375 S/s complex frames.
In the reference this stage is GNU Radio's own filter blocks (third-party, taps
version-dependent: parity unpinned), so the kernel is checked against a float64
restatement of ITS OWN published formula, and end to end on the reference's example
recording examples/150613_1920.wav (committed as tests/golden/150613_1920_int16.npz),
whose known decode is `VE3EMB FN42 33`."""
import os

import numpy as np
import pytest

from conftest import GOLDEN


def _ref_frontend(x, g, nout=45000):
    import scipy.signal as ss
    z = ss.fftconvolve(x.astype(np.float64), g.astype(np.complex128))
    D = (len(g) - 1) // 2
    idx = D + 32 * np.arange(nout)
    y = np.zeros(nout, np.complex128)
    ok = idx < len(z)
    y[ok] = z[idx[ok]]
    return y


def test_taps_are_a_unit_gain_lowpass_times_the_mixer(G):
    g = G.frontend_taps()
    assert len(g) == 1025
    h = np.abs(g)
    assert np.allclose(h, h[::-1], atol=1e-9)                 # linear phase
    D = 512
    k = np.arange(1025)
    mix = np.exp(-1j * np.pi * (D - k) / 4)
    hr = (g / mix).real
    assert abs(hr.sum() - 1.0) < 1e-5                         # unit DC gain
    assert np.abs((g / mix).imag).max() < 1e-6
    H = np.abs(np.fft.rfft(hr, 1 << 16))
    f = np.fft.rfftfreq(1 << 16, 1 / 12000.0)
    assert H[f < 60].min() > 0.98 and H[f > 187.5].max() < 2e-3   # pass band / alias band


@pytest.mark.gpu
def test_kernel_matches_float64_formula(G):
    rng = np.random.default_rng(0)
    x = rng.standard_normal((2, 45000 * 32)).astype(np.float32)
    x[1, 700000:] = 0.0
    ctx = G.Context()
    try:
        y = ctx.frontend(x)
        ys = ctx.frontend(x[:, :500000])          # short record: zero beyond the end
    finally:
        ctx.close()
    g = G.frontend_taps()
    for b in range(2):
        ref = _ref_frontend(x[b], g)
        got = y[b, :, 0] + 1j * y[b, :, 1]
        assert np.abs(got - ref).max() <= 1e-5 * np.abs(ref).max()
    ref = _ref_frontend(x[0, :500000], g)
    got = ys[0, :, 0] + 1j * ys[0, :, 1]
    assert np.abs(got - ref).max() <= 1e-5 * np.abs(ref).max()
    assert not ys[0, 16000:].any()


@pytest.mark.gpu
def test_reference_recording_decodes_end_to_end(G):
    """BASELINE configs[4] plumbing: wav -> K0 -> FDR -> schedule -> Fano -> unpack."""
    rec = np.load(os.path.join(GOLDEN, "150613_1920_int16.npz"))
    x = rec["x"].astype(np.float32) / 32768.0
    rng = np.random.default_rng(1)
    ctx = G.Context()
    try:
        frames = ctx.frontend(np.stack([x, x + 0.5 * rng.standard_normal(x.size).astype(np.float32)]))
        cands, out = ctx.pipeline_batch(frames, max_per_frame=2)
    finally:
        ctx.close()
    for b in range(2):
        texts = []
        for j in range(min(2, len(cands[b]))):
            dec = G.decode_candidate(out[b, j])
            if dec is not None:
                texts.append(G.unpack_message(dec[0])[1])
        assert "VE3EMB FN42 33" in texts, (b, texts)


@pytest.mark.gpu
def test_synthetic_audio_round_trip(G):
    audio, meta = G.synth.make_audio(3, snr_db=-18.0)
    ctx = G.Context()
    try:
        frames = ctx.frontend(audio)
        cands, out = ctx.pipeline_batch(frames, max_per_frame=1)
    finally:
        ctx.close()
    for b in range(3):
        dec = G.decode_candidate(out[b, 0])
        assert dec is not None
        assert (np.unpackbits(dec[0].view(np.uint8))[:50] == meta[b]["bits"]).all()
        assert abs(float(out[b, 0]["f1"]) - meta[b]["f_off"]) < 0.2
