"""Seeded synthetic WSPR frames (SURVEY.md section 8(d) generator).

Frame b: 50 random message bits -> K=32 r=1/2 convolutional code (polynomials
0xf2d05351 / 0xe4613c47, lib/Fano.cc:54-100) -> bit-reversal interleave (the
inverse of sync_and_demodulate_impl.cc:265-282) -> channel symbol
s_i = pr3[i] + 2*data_i -> continuous-phase 4-FSK, tone (s_i-1.5)*375/256 Hz,
256 samples/symbol at 375 S/s, unit amplitude, starting at sample 375 (like
examples/VE3EMB.c2) -> + carrier offset, + optional linear drift, + complex
AWGN.  Output is the PDU payload: interleaved (I,Q) float32, [B, 45000, 2].

This is test/bench input plumbing, not part of the measured path.
"""
import numpy as np

FS = 375.0
FL = 45000
NSYM = 162
SPB = 256
START = 375
POLY1, POLY2 = 0xF2D05351, 0xE4613C47

_PR3_WORDS = (0x07A47103, 0x58B340A4, 0x56349558, 0xE2CDC904, 0x63580CA0, 0x00000000)
PR3 = np.array([(_PR3_WORDS[k >> 5] >> (k & 31)) & 1 for k in range(NSYM)], np.uint8)


def _parity32(v):
    v = v ^ (v >> 16)
    v = v ^ (v >> 8)
    v = v ^ (v >> 4)
    v = v ^ (v >> 2)
    v = v ^ (v >> 1)
    return (v & 1).astype(np.uint8)


def interleave_order():
    """order[p] = j: encoded symbol p is transmitted at position j."""
    order = []
    i = 0
    while len(order) < NSYM:
        j = int("{:08b}".format(i)[::-1], 2)
        if j < NSYM:
            order.append(j)
        i += 1
    return np.array(order)


def encode_messages(bits50):
    """bits50: [B,50] of 0/1 -> channel symbols [B,162] in 0..3."""
    bits50 = np.asarray(bits50, dtype=np.uint64)
    B = bits50.shape[0]
    bits = np.zeros((B, 81), np.uint64)
    bits[:, :50] = bits50
    state = np.zeros(B, np.uint64)
    enc = np.zeros((B, NSYM), np.uint8)
    for k in range(81):
        state = ((state << np.uint64(1)) | bits[:, k]) & np.uint64(0xFFFFFFFF)
        enc[:, 2 * k] = _parity32(state & np.uint64(POLY1))
        enc[:, 2 * k + 1] = _parity32(state & np.uint64(POLY2))
    tx = np.zeros_like(enc)
    tx[:, interleave_order()] = enc
    return (PR3[None, :] + 2 * tx).astype(np.uint8)


def sigma_for_snr(snr_db_2500):
    """AWGN sigma per component for a unit-amplitude tone:
    SNR_2500 = 10log10(1/(2 sigma^2)) - 8.24 dB (375 Hz -> 2500 Hz bandwidth)."""
    return float(np.sqrt(0.5 * 10.0 ** (-(snr_db_2500 + 8.24) / 10.0)))


def make_frames(B, seed=0xC0FFEE, snr_db=-20.0, halfbandwidth=10, maxdrift=0.0, first=0,
                return_meta=False):
    """numpy generator; frame index b uses seed + first + b (reproducible per frame)."""
    frames = np.zeros((B, FL, 2), np.float32)
    meta = []
    sigma = sigma_for_snr(snr_db) if snr_db is not None else 0.0
    t = np.arange(NSYM * SPB, dtype=np.float64) / FS
    tc = t - t[-1] / 2.0
    for b in range(B):
        rng = np.random.Generator(np.random.Philox(seed + first + b))
        bits = rng.integers(0, 2, size=(1, 50))
        sym = encode_messages(bits)[0].astype(np.float64)
        f_off = rng.uniform(-(halfbandwidth - 4), (halfbandwidth - 4))
        drift = rng.uniform(-maxdrift, maxdrift) if maxdrift > 0 else 0.0
        ftone = np.repeat((sym - 1.5) * FS / SPB, SPB) + f_off + 0.5 * drift * tc / (t[-1] / 2.0)
        phase = 2.0 * np.pi * np.cumsum(ftone) / FS
        sig = np.zeros((FL, 2), np.float64)
        sig[START:START + NSYM * SPB, 0] = np.cos(phase)
        sig[START:START + NSYM * SPB, 1] = np.sin(phase)
        if sigma > 0:
            sig += sigma * rng.standard_normal((FL, 2))
        frames[b] = sig.astype(np.float32)
        meta.append({"bits": bits[0].astype(np.uint8), "f_off": f_off, "drift": drift})
    return (frames, meta) if return_meta else frames


def make_frames_torch(B, device, seed=0xC0FFEE, snr_db=-20.0, halfbandwidth=10, maxdrift=0.0,
                      chunk=1024):
    """Same signal model synthesised in HBM with torch (bench input only; the
    random streams differ from make_frames)."""
    import torch
    rng = np.random.Generator(np.random.Philox(seed))
    bits = rng.integers(0, 2, size=(B, 50))
    sym = encode_messages(bits)
    f_off = rng.uniform(-(halfbandwidth - 4), (halfbandwidth - 4), size=B)
    drift = rng.uniform(-maxdrift, maxdrift, size=B) if maxdrift > 0 else np.zeros(B)
    sigma = sigma_for_snr(snr_db) if snr_db is not None else 0.0
    out = torch.zeros((B, FL, 2), dtype=torch.float32, device=device)
    g = torch.Generator(device=device)
    g.manual_seed(int(seed) & 0x7FFFFFFF)
    nt = NSYM * SPB
    tc = (torch.arange(nt, dtype=torch.float64, device=device) / FS)
    half = float(nt - 1) / FS / 2.0
    tcn = (tc - half) / half
    for s in range(0, B, chunk):
        e = min(B, s + chunk)
        tone = torch.from_numpy((sym[s:e].astype(np.float64) - 1.5) * FS / SPB).to(device)
        f = tone.repeat_interleave(SPB, dim=1)
        f = f + torch.from_numpy(f_off[s:e]).to(device)[:, None]
        f = f + 0.5 * torch.from_numpy(drift[s:e]).to(device)[:, None] * tcn[None, :]
        phase = 2.0 * np.pi * torch.cumsum(f, dim=1) / FS
        out[s:e, START:START + nt, 0] = torch.cos(phase).float()
        out[s:e, START:START + nt, 1] = torch.sin(phase).float()
        if sigma > 0:
            out[s:e] += sigma * torch.randn((e - s, FL, 2), generator=g, device=device,
                                            dtype=torch.float32)
    return out


def make_audio(B, seed=0xA0D10, snr_db=-15.0, f_off=None, rate=12000, carrier=1500.0):
    """Real 12 kS/s audio records (B x 1.44 M samples) carrying one WSPR transmission
    each at `carrier` + offset Hz: input for the K0 front-end.
    Symbol length 8192 samples (256 at 375 S/s), start at sample 375*32."""
    nin = FL * 32
    out = np.zeros((B, nin), np.float32)
    meta = []
    spb = SPB * 32
    for b in range(B):
        rng = np.random.Generator(np.random.Philox(seed + b))
        bits = rng.integers(0, 2, size=(1, 50))
        sym = encode_messages(bits)[0].astype(np.float64)
        fo = rng.uniform(-4, 4) if f_off is None else f_off
        ftone = np.repeat((sym - 1.5) * FS / SPB, spb) + fo + carrier
        phase = 2.0 * np.pi * np.cumsum(ftone) / rate
        sig = np.zeros(nin)
        s0 = START * 32
        sig[s0:s0 + NSYM * spb] = np.cos(phase)
        # complex baseband amplitude after the mixer is 1/2: sigma for SNR in 2500 Hz
        if snr_db is not None:
            sigma = 0.5 * np.sqrt(10.0 ** (-(snr_db) / 10.0) * (rate / 2.0) / 2500.0) * np.sqrt(2.0)
            sig += sigma * rng.standard_normal(nin)
        out[b] = sig.astype(np.float32)
        meta.append({"bits": bits[0].astype(np.uint8), "f_off": fo})
    return out, meta
