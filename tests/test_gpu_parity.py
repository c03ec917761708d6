"""GPU parity: the HIP path (through the C ABI) against the CPU oracle on the same
seeded inputs, and against the committed golden vectors.  Integer / byte /
index fields bit-exact; float metrics within 1e-5 relative (north_star), and in
practice bit-exact except where noted (log10, binary64 sin/cos libraries).
"""
import json
import os

import numpy as np
import pytest

from conftest import GOLDEN

pytestmark = pytest.mark.gpu
RTOL = 1e-5   # BASELINE.json north_star tolerance for float metrics


@pytest.fixture(scope="module")
def ctx(G):
    c = G.Context()
    yield c
    c.close()


@pytest.fixture(scope="module")
def frames(G):
    return G.synth.make_frames(4, seed=0xC0FFEE, snr_db=-20.0)


@pytest.fixture(scope="module")
def vec():
    return np.load(os.path.join(GOLDEN, "oracle_vectors.npz"))


def cand_equal(a, b, lin_strict=True):
    """Compare the fields the PDU carries (FDR_impl.cc:414-455)."""
    assert int(a["m_type"]) == int(b["m_type"])
    assert int(a["shift"]) == int(b["shift"])
    assert np.float32(a["freq"]).tobytes() == np.float32(b["freq"]).tobytes()
    assert abs(float(a["sync"]) - float(b["sync"])) <= RTOL * abs(float(b["sync"]))
    # cc:303 to the bit: the kernel restates this libm's log10f (tests/test_log10_gap.py)
    assert np.float32(a["snr"]).tobytes() == np.float32(b["snr"]).tobytes()
    if int(a["m_type"]) == 0:
        da = np.frombuffer(a.tobytes()[24:28], np.float32)[0]
        db = np.frombuffer(b.tobytes()[24:28], np.float32)[0]
        assert da == db
    else:
        for k in ("V1", "V2", "p1", "p2"):
            assert a[k] == b[k]


def test_info_matches_reference_constants(ctx):
    i = ctx.info
    assert (i.size, i.m, i.hpbm, i.n, i.finpb, i.noiseidx) == (512, 256, 14, 348, 28, 8)
    assert i.df == 0.732421875 and i.cell_hyps == 126
    assert i.device_name.decode().startswith("gfx950")


def test_spectrogram_and_stats_bit_exact(ctx, oracle, frames):
    """K1 + K2 vs the oracle: ps tile, psavg, smspec, noise (rows a3, a4)."""
    ctx.fdr_batch(frames)
    ps, psavg, smraw, smspec, noise = ctx.fdr_spectrum(frames.shape[0])
    f = oracle.FDR()
    lo, w = ctx.info.band_lo, ctx.info.band_w
    for b in range(frames.shape[0]):
        ops = f.spectrogram(frames[b])
        assert ps[b].tobytes() == ops[:, lo:lo + w].tobytes()
        opsavg, osmraw, osmspec, onoise = f.stats(ops)
        assert psavg[b].tobytes() == opsavg[lo:lo + w].tobytes()
        assert smraw[b].tobytes() == osmraw.tobytes()
        assert smspec[b].tobytes() == osmspec.tobytes()
        assert np.float32(noise[b]).tobytes() == np.float32(onoise).tobytes()


@pytest.mark.parametrize("rows", ["29", "22", "5"])
def test_spectrogram_rows_per_wavefront_and_wide_band(G, oracle, frames, monkeypatch, rows):
    """K1 walks 29 consecutive rows per wavefront for large batches and 6 for small ones (the
    batches of these tests): the large-batch walk, an uneven split (22: last group 18 rows) and a
    short one give the same bytes as the oracle; so does the full (unpruned) pass C that a band
    wider than +-64 columns selects (halfbandwidth = 60)."""
    monkeypatch.setenv("UWSPR_OPTIONS", "k1_rows=" + rows)   # (shapes the context when it is created)
    for kw in ({}, {"halfbandwidth": 60}):
        c = G.Context(**kw)
        try:
            c.fdr_batch(frames)
            ps = c.fdr_spectrum(frames.shape[0])[0]
            lo, w = c.info.band_lo, c.info.band_w
        finally:
            c.close()
        f = oracle.FDR(**kw)
        assert (lo >= 192 and lo + w <= 320) == (not kw)      # the default band takes the pruned pass
        for b in range(frames.shape[0]):
            assert ps[b].tobytes() == f.spectrogram(frames[b])[:, lo:lo + w].tobytes(), (rows, kw, b)


def test_fdr_candidates_match_oracle(ctx, oracle, frames):
    """rows a5-a11: candidate lists, order, selection."""
    got = ctx.fdr_batch(frames)
    f = oracle.FDR()
    for b in range(frames.shape[0]):
        exp = f.transform(frames[b])
        assert len(got[b]) == len(exp) > 0
        for a, e in zip(got[b], exp):
            cand_equal(a, e)


def test_fdr_candidates_match_committed_vectors(ctx, frames, vec):
    got = ctx.fdr_batch(frames)
    for b in range(frames.shape[0]):
        n = int(vec["npk"][b])
        assert len(got[b]) == n
        for a, e in zip(got[b], vec["cands"][b, :n]):
            cand_equal(a, e)


def test_fdr_syncgrid_bit_exact(ctx, oracle, frames, vec):
    """rows a6-a10: every one of the 16380 hypothesis metrics of candidate 0."""
    ctx.keep_syncgrid(1)
    try:
        ctx.fdr_batch(frames)
        grid = ctx.fdr_syncgrid(frames.shape[0])
    finally:
        ctx.keep_syncgrid(0)
    assert grid.shape[2:] == (5, 26, 126)
    for b in range(frames.shape[0]):
        assert grid[b, 0].tobytes() == vec["grid0"][b].tobytes()


def test_ve3emb_known_answers_on_gpu(ctx, G, ve3emb):
    known = json.load(open(os.path.join(GOLDEN, "ve3emb_known.json")))
    cands, out = ctx.pipeline_batch(ve3emb[None], max_per_frame=1)
    c = cands[0][0]
    assert len(cands[0]) == known["npk"] and int(c["m_type"]) == known["m_type"]
    assert float(c["freq"]) == known["freq"] and int(c["shift"]) == known["shift"]
    assert (float(c["V1"]), float(c["V2"]), int(c["p1"]), int(c["p2"])) == (-1.0, -1.0, 0, 650)
    assert "%.9f" % c["sync"] == "%.9f" % known["sync"]
    dec = G.decode_candidate(out[0, 0])
    assert dec is not None
    msg, idt = dec
    assert bytes(msg.view(np.uint8)).hex() == known["blob_hex"]
    assert G.unpack_message(msg) == (0, known["message"])


def test_fdr_linear_drift_grid_and_threshold(G, oracle, frames):
    """maxdrift > 0 (linear sweep, row a7) and a threshold that disables SLM wins."""
    for kw in ({"maxdrift": 2}, {"threshold": 1000000}, {"halfbandwidth": 40, "maxdrift": 1}):
        c = G.Context(**kw)
        try:
            got = c.fdr_batch(frames[:2])
        finally:
            c.close()
        f = oracle.FDR(**kw)
        for b in range(2):
            exp = f.transform(frames[b])
            assert len(got[b]) == len(exp)
            for a, e in zip(got[b], exp):
                cand_equal(a, e)


def test_sync_sweep_config3_grid(ctx, G, oracle, frames, vec):
    """rows a13/a14: 2 frames x 200 (freq, lag, drift) hypotheses, soft symbols
    byte-exact, sync within tolerance (committed vectors + live oracle spot check)."""
    hy = vec["sweep_hyps"]
    sync, sym = ctx.sync_sweep(frames, hy, soft=True)
    assert (sym == vec["sweep_symbols"]).all()
    np.testing.assert_allclose(sync, vec["sweep_sync"], rtol=RTOL, atol=0)
    assert sync.tobytes() == vec["sweep_sync"].tobytes()  # in practice bit-exact
    lin = np.zeros(1, oracle.CAND_DTYPE)[0]
    for q in (0, 57, 399):
        h = hy[q]
        s, _, _, y = oracle.sync_and_demodulate(lin, 1500, frames[h["frame"]], float(h["f0"]), 0, 0,
                                                0.0, int(h["lag"]), 0, 0, 1, float(h["drift"]), 50, 2)
        assert (y == sym[q]).all() and np.float32(s).tobytes() == sync[q].tobytes()


def test_sync_grid_equals_flat_sweep(ctx, G, oracle, frames, vec):
    """uwspr_sync_grid (shared windows + shared phasors) is bit-identical to the
    flat uwspr_sync_sweep on the expanded hypothesis list: config-3 grid shape,
    a nonlinear centre, an edge-of-frame centre, odd sizes and a > 8 lag list."""
    N = G.native
    cents = np.zeros(4, N.CAND_DTYPE)
    cents[:] = vec["cands"][:, 0]
    cents[2]["m_type"] = 1; cents[2]["V1"] = -1.0; cents[2]["V2"] = 2.0; cents[2]["p1"] = 0; cents[2]["p2"] = 450
    cents[3]["shift"] = 40          # lags reach before sample 0 -> cc:205 skipping
    cases = [
        (np.arange(-2, 3) * np.float32(0.25), np.array([-1, -0.5, 0, 0.5, 1], np.float32),
         np.array([-96, -64, -32, 0, 32, 64, 96, 128], np.int32)),                   # configs[2]
        (np.array([0.0, 0.05, -0.05], np.float32), np.array([0.0], np.float32),
         np.array([0, -8, 8, -16, 16, -24, 24, -32, 32, -40, 40], np.int32)),        # 11 lags: two blocks
        (np.array([0.1], np.float32), np.array([0.0, 0.5], np.float32), np.array([5], np.int32)),
        (np.arange(7, dtype=np.float32) * np.float32(0.125), np.array([0.25], np.float32),
         np.array([3600, 3400, 100], np.int32)),                                     # runs past the frame end
    ]
    for df, dd, dl in cases:
        sync, sym = ctx.sync_grid(frames, cents, df, dd, dl, soft=True)
        hy = np.zeros(sync.size, N.HYP_DTYPE)
        q = 0
        for b in range(4):
            for i in range(df.size):
                for j in range(dd.size):
                    for k in range(dl.size):
                        h = hy[q]
                        h["frame"] = b; h["m_type"] = cents[b]["m_type"]
                        h["f0"] = np.float32(cents[b]["freq"]) + np.float32(df[i])
                        lin = np.frombuffer(cents[b].tobytes()[24:28], np.float32)[0]
                        h["drift"] = np.float32(lin) + np.float32(dd[j]) if cents[b]["m_type"] == 0 else 0.0
                        h["lag"] = int(cents[b]["shift"]) + int(dl[k])
                        for key in ("V1", "V2", "p1", "p2"):
                            h[key] = cents[b][key] if cents[b]["m_type"] == 1 else 0
                        q += 1
        fsync, fsym = ctx.sync_sweep(frames, hy, soft=True)
        assert sync.reshape(-1).tobytes() == fsync.tobytes()
        assert (sym.reshape(-1, 162) == fsym).all()
    # and against the committed oracle vectors for the configs[2] grid of frames 0,1
    df, dd, dl = cases[0]
    sync, sym = ctx.sync_grid(frames[:2], vec["cands"][:2, 0], df, dd, dl, soft=True)
    order = vec["sweep_hyps"]   # generated f-outer, lag, drift-inner: reorder to [f][drift][lag]
    idx = np.arange(400).reshape(2, 5, 8, 5).transpose(0, 1, 3, 2).reshape(-1)
    assert (sym.reshape(-1, 162) == vec["sweep_symbols"][idx]).all()
    assert sync.reshape(-1).tobytes() == vec["sweep_sync"][idx].tobytes()


def test_sync_sweep_edges(ctx, G, oracle, frames):
    """lags that run off either end of the frame (n<=0 and n>=np are skipped,
    cc:205), nonlinear hypotheses, skipped (frame<0) entries, odd batch sizes."""
    N = G.native
    hy = np.zeros(9, N.HYP_DTYPE)
    hy["frame"] = [0, 0, 1, 1, -1, 2, 3, 3, 3]
    hy["lag"] = [-300, -1, 0, 3600, 100, 3528, 1, 368, 5000]
    hy["f0"] = [0.5, -0.25, 0.0, 1.0, 0.0, -1.5, 0.1, 0.0, 2.0]
    hy["drift"] = [0.0, 0.5, -1.0, 0.0, 0.0, 2.0, 0.0, 0.0, -0.5]
    hy["m_type"][6] = 1; hy["V1"][6] = -2.0; hy["V2"][6] = 1.0; hy["p2"][6] = 250
    hy["m_type"][7] = 1; hy["V1"][7] = 1.0; hy["V2"][7] = 2.0; hy["p2"][7] = 50
    sync, sym = ctx.sync_sweep(frames, hy, soft=True)
    for q, h in enumerate(hy):
        if h["frame"] < 0:
            assert sync[q] == np.float32(-1e30) and not sym[q].any()
            continue
        cand = np.zeros(1, oracle.CAND_DTYPE)[0]
        cand["m_type"] = h["m_type"]; cand["V1"] = h["V1"]; cand["V2"] = h["V2"]
        cand["p1"] = h["p1"]; cand["p2"] = h["p2"]
        s, _, _, y = oracle.sync_and_demodulate(cand, 1500, frames[h["frame"]], float(h["f0"]), 0, 0,
                                                0.0, int(h["lag"]), 0, 0, 1, float(h["drift"]), 50, 2)
        assert (y == sym[q]).all(), q
        assert np.float32(s).tobytes() == sync[q].tobytes(), q


def test_sync_and_demodulate_calls_read_like_the_reference(ctx, G, oracle, frames, vec):
    """Argument-for-argument form of sync_and_demodulate() (cc:126-131), modes 0/1/2."""
    N = G.native
    cands = vec["cands"]
    calls = np.zeros(6, N.CALL_DTYPE)
    specs = [  # frame, f1, ifmin, ifmax, fstep, shift1, lagmin, lagmax, lagstep, drift1, mode
        (0, float(cands[0, 0]["freq"]), 0, 0, 0.0, 256, 128, 384, 64, 0.0, 0),
        (0, float(cands[0, 0]["freq"]), -2, 2, 0.25, 320, 0, 0, 64, 0.0, 1),
        (1, 0.1, -2, 2, 0.05, 368, 0, 0, 16, 0.5, 1),
        (2, float(cands[2, 0]["freq"]), 0, 0, 0.0, 300, 268, 332, 16, -0.5, 0),
        (3, -0.3, 0, 0, 0.0, 360, 0, 0, 16, 0.0, 2),
        (3, 0.0, -1, 3, 0.1, 368, 0, 0, 8, 1.0, 1),
    ]
    for q, s in enumerate(specs):
        c = calls[q]
        c["frame"], c["f1"], c["ifmin"], c["ifmax"], c["fstep"], c["shift1"] = s[:6]
        c["lagmin"], c["lagmax"], c["lagstep"], c["drift1"], c["mode"] = s[6:]
        c["symfac"] = 50
    res = ctx.sync_and_demodulate(frames, calls)
    lin = np.zeros(1, oracle.CAND_DTYPE)[0]
    for q, s in enumerate(specs):
        sy, sh, f1, y = oracle.sync_and_demodulate(lin, 1500, frames[s[0]], s[1], s[2], s[3], s[4],
                                                   s[5], s[6], s[7], s[8], s[9], 50, s[10])
        assert np.float32(sy).tobytes() == res[q]["sync"].tobytes()
        if s[10] <= 1:
            assert sh == res[q]["shift1"] and np.float32(f1).tobytes() == res[q]["f1"].tobytes()
        else:
            assert (y == res[q]["symbols"]).all()


def test_schedule_matches_oracle(ctx, oracle, frames, vec):
    """row a15: the S0..S5 refinement schedule, every candidate of every frame."""
    cands = [vec["cands"][b, :int(vec["npk"][b])] for b in range(4)]
    per = max(len(c) for c in cands)
    out = ctx.demod_batch(frames, cands, max_per_frame=per)
    for b in range(4):
        for j in range(len(cands[b])):
            d = oracle.demod_candidate(cands[b][j], 1500, frames[b])
            o = out[b, j]
            assert int(o["worth_a_try"]) == d["worth_a_try"] and int(o["shift1"]) == d["shift1"]
            for k in ("f1", "drift1", "sync1"):
                assert np.float32(o[k]).tobytes() == np.float32(d[k]).tobytes(), (b, j, k)
            if d["worth_a_try"]:
                assert (o["symbols"] == d["symbols"]).all()
                assert (o["jig_shift"] == d["jig_shift"]).all()
                assert o["jig_sync"].tobytes() == d["jig_sync"].tobytes()
                assert o["jig_rms"].tobytes() == d["jig_rms"].tobytes()
        for j in range(len(cands[b]), per):
            assert int(out[b, j]["worth_a_try"]) == 0 and not out[b, j]["symbols"].any()
    assert (out[:, 0]["symbols"] == vec["demod_symbols"]).all()


def test_schedule_forms_are_identical(G, frames, vec):
    """The refinement schedule exists in three independent forms -- the fused kernel (k6_sched), the staged launches
    with the packed / ring / pair kernels (k4_lag0, k4_fpack, k4_ring, k4_dpair) and the staged launches with the flat
    kernel (k4_tonecorr) for every stage -- each with and without the phasor tables and the stage-winner reuse:
    byte-identical records.  (A fourth, the rows form of round 4, lost under three streams and left the library in
    round 5: profiles/HISTORY.md.)"""
    cands = [vec["cands"][b, :int(vec["npk"][b])] for b in range(4)]
    per = max(len(c) for c in cands)
    outs = []
    for opts in ({"sched": 1},
                 {"sched": 0, "stage_kernels": 0, "reuse": 0, "phasor_tables": 0},
                 {"sched": 0, "stage_kernels": 0},
                 {"sched": 0, "stage_kernels": 1},
                 {"sched": 0, "stage_kernels": 1, "reuse": 0},
                 {"sched": 0, "stage_kernels": 1, "phasor_tables": 0},
                 {"sched": 0, "stage_kernels": 1, "k4_forms": 0},      # S5 through the LDS-ring kernel (round 4's form)
                 {"sched": 0, "stage_kernels": 1, "k4_forms": 0, "reuse": 0},
                 {"sched": 0, "stage_kernels": 1, "k4_forms": 3},      # S0 double-buffered, one barrier per chunk (round 6)
                 {"sched": 0, "stage_kernels": 1, "k4_forms": 5},      # S0: + its lags on two wavefronts per tone
                 {"sched": 1, "reuse": 0}):
        c = G.Context(options=opts)
        try:
            assert all(c.get_option(k) == v for k, v in opts.items())
            outs.append(c.demod_batch(frames, cands, max_per_frame=per))
        finally:
            c.close()
    for o in outs[1:]:
        assert outs[0].tobytes() == o.tobytes()
    assert (outs[0][:, 0]["symbols"] == vec["demod_symbols"]).all()


def test_options_are_checked(G, ctx):
    with pytest.raises(G.UwsprError):
        ctx.set_option("no_such_option", 1)
    with pytest.raises(G.UwsprError):
        ctx.set_option("k3_tile", 1)          # shapes the context at creation: UWSPR_OPTIONS only
    assert ctx.get_option("sched") in (0, 1)


def test_schedule_forms_on_random_candidates(G, oracle):
    """Hand-made candidates that the packed / table kernels must not trip over: frames with 0..5 candidates (dead
    slots between live ones), shifts from before the frame start to the last one whose windows fit, drifting linear
    models (no phasor table: the recurrence kernels take them) next to drift-free and straight-line ones in the same
    workgroups.  Fused kernel, staged form and staged form with the flat kernel only: identical bytes;
    a sample of the records against the oracle."""
    rng = np.random.default_rng(2024)
    frames = np.concatenate([G.synth.make_frames(4, seed=606, snr_db=-17.0),
                             (0.5 * rng.standard_normal((2, 45000, 2))).astype(np.float32)])
    per = 5
    cands = []
    for b in range(6):
        n = [5, 0, 3, 1, 0, 4][b]
        c = np.zeros(n, oracle.CAND_DTYPE)
        for j in range(n):
            c[j]["freq"] = np.float32(rng.uniform(-8, 8))
            c[j]["shift"] = int(rng.choice([-50, 0, 200, 375, 1500, 3300, 3600]))
            c[j]["sync"] = 0.3
            if rng.random() < 0.5:
                c[j]["m_type"] = 1
                c[j]["V1"] = float(rng.integers(-2, 3)); c[j]["V2"] = float(rng.integers(-2, 3))
                c[j]["p1"] = 0; c[j]["p2"] = int(rng.choice([50, 250, 450, 650, 850]))
            else:
                c[j]["m_type"] = 0
                drift = np.float32(rng.choice([0.0, 0.0, 0.5, -1.5]))
                c[j] = np.frombuffer(c[j].tobytes()[:24] + drift.tobytes() + c[j].tobytes()[28:], oracle.CAND_DTYPE)[0]
        cands.append(c)
    cands[0][0]["freq"] = frames.dtype.type(0.0)       # one candidate on the generated signal's grid
    outs = {}
    for name, opts in (("fused", {"sched": 1}),
                       ("staged", {"sched": 0}),
                       ("staged-lds-ring", {"sched": 0, "k4_forms": 0}),
                       ("staged-no-tables", {"sched": 0, "phasor_tables": 0}),     # every wavefront on per-lane recurrences
                       ("staged-no-tables-no-reuse", {"sched": 0, "phasor_tables": 0, "reuse": 0}),
                       ("staged-flat", {"sched": 0, "stage_kernels": 0, "phasor_tables": 0})):
        c = G.Context(options=opts)
        try:
            outs[name] = c.demod_batch(frames, cands, max_per_frame=per)
        finally:
            c.close()
    for name in ("staged", "staged-lds-ring", "staged-no-tables", "staged-no-tables-no-reuse", "staged-flat"):
        assert outs[name].tobytes() == outs["fused"].tobytes(), name
    for b, j in ((0, 0), (0, 3), (2, 1), (3, 0), (5, 2)):
        d = oracle.demod_candidate(cands[b][j], 1500, frames[b])
        o = outs["staged"][b, j]
        assert int(o["shift1"]) == d["shift1"] and int(o["worth_a_try"]) == d["worth_a_try"], (b, j)
        for k in ("f1", "drift1", "sync1"):
            assert np.float32(o[k]).tobytes() == np.float32(d[k]).tobytes(), (b, j, k)
        if d["worth_a_try"]:
            assert (o["symbols"] == d["symbols"]).all(), (b, j)


@pytest.mark.parametrize("fused", ["1", "0"])
def test_whole_pipeline_on_fresh_and_growing_contexts(G, frames, monkeypatch, fused):
    """uwspr_pipeline_batch (candidates straight from the context's own FDR: the stage-0 lag-group launch is then
    left out when maxdrift = 0) in both schedule forms: the FIRST call of a fresh context, a larger batch after a
    small one (every per-slot scratch buffer grows between the calls) and the small one again give the bytes of the
    reference configuration (fused form, one context per call), host and device frames, eager and lazy + resume."""
    import torch
    big = np.concatenate([frames, frames[::-1], frames, frames[1:3]])     # 14 frames
    ref = {}
    for name, fr in (("small", frames[:3]), ("big", big)):
        c = G.Context(options={"sched": 1})
        try:
            ref[name] = c.pipeline_batch(fr, max_per_frame=2)
        finally:
            c.close()
    c = G.Context(options={"sched": int(fused)})
    try:
        for name, fr in (("small", frames[:3]), ("big", big), ("small", frames[:3])):
            cands, out = c.pipeline_batch(fr, max_per_frame=2)
            assert out.tobytes() == ref[name][1].tobytes(), (fused, name)
            for a, b in zip(cands, ref[name][0]):
                assert a.tobytes() == b.tobytes()
        dev = torch.from_numpy(big).cuda()
        cands, out = c.pipeline_batch(dev, max_per_frame=2)
        assert out.tobytes() == ref["big"][1].tobytes()
        c.set_tries(1)
        c.pipeline_batch(big, max_per_frame=2)
        res = c.demod_resume(big, np.ones((14, 2), np.uint8), None, max_per_frame=2)
        assert res.tobytes() == ref["big"][1].tobytes()
    finally:
        c.close()


def test_pipeline_device_pointers_equal_host_pointers(ctx, G, frames):
    """The same batch through UWSPR_DEVICE pointers (HBM-resident, what bench.py times)."""
    import torch
    dev = torch.from_numpy(frames).cuda()
    torch.cuda.synchronize()
    c1, o1 = ctx.pipeline_batch(frames, max_per_frame=2)
    c2, o2 = ctx.pipeline_batch(dev, max_per_frame=2)
    for a, b in zip(c1, c2):
        assert a.tobytes() == b.tobytes()
    assert o1.tobytes() == o2.tobytes()


def test_many_candidates_per_frame(G, oracle):
    """Frames carrying three transmissions at different offsets in a 40 Hz half band:
    several candidates per frame go through the compacted K3 work list and the
    schedule with max_per_frame > 1; everything against the oracle."""
    rng = np.random.default_rng(5)
    parts = [G.synth.make_frames(3, seed=900 + 10 * k, snr_db=None, halfbandwidth=40) for k in range(3)]
    frames = parts[0] + 0.8 * parts[1] + 0.7 * parts[2]
    frames = (frames + 1.5 * rng.standard_normal(frames.shape)).astype(np.float32)
    kw = {"halfbandwidth": 40, "maxdrift": 1}
    c = G.Context(**kw)
    try:
        cands, out = c.pipeline_batch(frames, max_per_frame=6)
    finally:
        c.close()
    f = oracle.FDR(**kw)
    ndec = 0
    for b in range(3):
        exp = f.transform(frames[b])
        assert len(exp) == len(cands[b]) >= 2
        for a, e in zip(cands[b], exp):
            cand_equal(a, e)
        for j in range(min(6, len(exp))):
            d = oracle.demod_candidate(exp[j], 1500, frames[b])
            o = out[b, j]
            assert int(o["worth_a_try"]) == d["worth_a_try"] and int(o["shift1"]) == d["shift1"]
            assert np.float32(o["f1"]).tobytes() == np.float32(d["f1"]).tobytes()
            if d["worth_a_try"]:
                assert (o["symbols"] == d["symbols"]).all()
                ndec += G.decode_candidate(o) is not None
    assert ndec == 8   # 2 + 3 + 3 transmissions found and decodable (oracle + real Fano say the same)


def test_gather_slabs_kernel_matches_reference_packing(ctx, G, frames):
    """uwspr_pack_slabs (what bench.py gathers across ranks) == dist.pack_slabs."""
    import torch
    from gr_uwspr_amd import dist as D
    N = G.native
    dev = torch.from_numpy(frames).cuda()
    B = frames.shape[0]
    cands_t = torch.empty(B * ctx.maxfreqs * 48, dtype=torch.uint8, device="cuda")
    npk_t = torch.empty(B, dtype=torch.int32, device="cuda")
    out_t = torch.empty(B * N.DEMOD_DTYPE.itemsize, dtype=torch.uint8, device="cuda")
    slab_t = torch.empty((B, D.SLAB_BYTES), dtype=torch.uint8, device="cuda")
    torch.cuda.synchronize()
    ctx.pipeline_batch_into(dev, cands_t, npk_t, out_t, max_per_frame=1)
    ctx.pack_slabs_into(B, D.SLAB_K, slab_t)
    ctx.synchronize()
    ref = D.pack_slabs(cands_t, npk_t, out_t, ctx.maxfreqs, 1, N.DEMOD_DTYPE.itemsize)
    got = slab_t.cpu().numpy()
    refn = ref.cpu().numpy()
    npk = npk_t.cpu().numpy()
    for b in range(B):
        k = min(int(npk[b]), D.SLAB_K)
        assert (got[b, :16 + 48 * k] == refn[b, :16 + 48 * k]).all()
        assert not got[b, 16 + 48 * k:16 + 48 * D.SLAB_K].any()      # zero-filled past npk
        assert (got[b, 16 + 48 * D.SLAB_K:] == refn[b, 16 + 48 * D.SLAB_K:]).all()
    host = np.zeros((B, D.SLAB_BYTES), np.uint8)
    ctx.pack_slabs_into(B, D.SLAB_K, host)
    assert (host == got).all()
    # uwspr_pipeline_slabs: the same bytes from the pipeline call itself, both schedule forms, eager and lazy
    for fused in ("1", "0"):
        c2 = G.Context(options={"sched": int(fused)})
        try:
            for tries in (17, 1):
                c2.set_tries(tries)
                slab2 = torch.zeros((B, D.SLAB_BYTES), dtype=torch.uint8, device="cuda")
                c2.pipeline_slabs(D.SLAB_K, slab2)
                c2.pipeline_batch_into(dev, cands_t, npk_t, out_t, max_per_frame=1)
                c2.synchronize()
                assert (slab2.cpu().numpy() == got).all(), (fused, tries)
                slab3 = torch.zeros((B, D.SLAB_BYTES), dtype=torch.uint8, device="cuda")   # one shot: not written again
                c2.pipeline_batch_into(dev, cands_t, npk_t, out_t, max_per_frame=1)
                c2.synchronize()
                assert not slab3.cpu().numpy().any()
        finally:
            c2.close()


def test_full_size_properties_256_frames(ctx, G):
    """BASELINE configs[1] size: 256 frames.  Size-independent properties:
    (1) batch result == per-frame result (frames are independent),
    (2) every -20 dB frame decodes back to the bits it was generated from."""
    frames, meta = G.synth.make_frames(256, seed=77, snr_db=-20.0, return_meta=True)
    cands, out = ctx.pipeline_batch(frames, max_per_frame=1)
    idx = [0, 31, 128, 255]
    for b in idx:
        c1, o1 = ctx.pipeline_batch(frames[b:b + 1], max_per_frame=1)
        assert c1[0].tobytes() == cands[b].tobytes() and o1[0].tobytes() == out[b].tobytes()
    ok = 0
    for b in range(256):
        dec = G.decode_candidate(out[b, 0])
        if dec is not None:
            bits = np.unpackbits(dec[0].view(np.uint8))[:50]
            ok += int((bits == meta[b]["bits"]).all())
    assert ok >= 250, ok


def test_context_rejects_bad_parameters(G):
    N = G.native
    with pytest.raises(N.UwsprError) as e:
        G.Context(halfbandwidth=188)
    assert e.value.status == -1          # reference exit(-1)s, FDR_impl.cc:85-90
    with pytest.raises(N.UwsprError) as e:
        G.Context(halfbandwidth=187)     # GRC XML default: the reference reads out of bounds
    assert e.value.status == -2


def test_argument_errors_are_status_codes(ctx, G, frames):
    """Bad calls come back as UWSPR_ERR_ARG with a message; nothing aborts the process."""
    import ctypes as C
    N = G.native
    L = N.lib()
    fr = np.ascontiguousarray(frames[:1])
    sync = np.zeros(4, np.float32)
    hy = np.zeros(1, N.HYP_DTYPE)
    hy["frame"] = 3          # only one frame in the batch
    rc = L.uwspr_sync_sweep(ctx.h, C.c_void_p(fr.ctypes.data), 1, C.c_void_p(hy.ctypes.data), 1, N.HOST,
                            C.c_void_p(sync.ctypes.data), None)
    assert rc == -6 and b"frame" in L.uwspr_last_error(ctx.h)
    assert L.uwspr_sync_sweep(ctx.h, None, 1, C.c_void_p(hy.ctypes.data), 1, N.HOST,
                              C.c_void_p(sync.ctypes.data), None) == -6
    assert L.uwspr_fdr_read_syncgrid(ctx.h, 1, C.c_void_p(sync.ctypes.data)) == -6   # grid not kept
    calls = np.zeros(1, N.CALL_DTYPE)
    calls["mode"] = 7
    res = np.zeros(1, N.RESULT_DTYPE)
    assert L.uwspr_sync_and_demodulate_batch(ctx.h, C.c_void_p(fr.ctypes.data), 1, N.HOST,
                                             C.c_void_p(calls.ctypes.data), 1,
                                             C.c_void_p(res.ctypes.data)) == -6
    calls["mode"] = 1; calls["symfac"] = 64; calls["lagstep"] = 1
    assert L.uwspr_sync_and_demodulate_batch(ctx.h, C.c_void_p(fr.ctypes.data), 1, N.HOST,
                                             C.c_void_p(calls.ctypes.data), 1,
                                             C.c_void_p(res.ctypes.data)) == -3
    # the context is still usable afterwards
    assert len(ctx.fdr_batch(fr)[0]) >= 1


def test_nonlinear_candidate_through_the_call_form(ctx, G, oracle, frames):
    """sync_and_demodulate() with a nonlinear candidate (t = 0 rule) in all three modes."""
    N = G.native
    calls = np.zeros(3, N.CALL_DTYPE)
    cand = np.zeros(1, oracle.CAND_DTYPE)[0]
    cand["m_type"] = 1; cand["V1"] = 2.0; cand["V2"] = -1.0; cand["p1"] = 0; cand["p2"] = 250
    specs = [(1, 0.4, 0, 0, 0.0, 300, 236, 364, 32, 0.0, 0), (1, 0.4, -2, 2, 0.25, 300, 0, 0, 64, 0.0, 1),
             (1, 0.4, 0, 0, 0.0, 364, 0, 0, 16, 0.0, 2)]
    for q, s in enumerate(specs):
        c = calls[q]
        c["frame"], c["f1"], c["ifmin"], c["ifmax"], c["fstep"], c["shift1"] = s[:6]
        c["lagmin"], c["lagmax"], c["lagstep"], c["drift1"], c["mode"] = s[6:]
        c["symfac"] = 50
        for k in ("m_type", "V1", "V2", "p1", "p2"):
            c["candidate"][k] = cand[k]
    res = ctx.sync_and_demodulate(frames, calls)
    for q, s in enumerate(specs):
        sy, sh, f1, y = oracle.sync_and_demodulate(cand, 1500, frames[s[0]], s[1], s[2], s[3], s[4],
                                                   s[5], s[6], s[7], s[8], s[9], 50, s[10])
        assert np.float32(sy).tobytes() == res[q]["sync"].tobytes()
        if s[10] <= 1:
            assert sh == res[q]["shift1"] and np.float32(f1).tobytes() == res[q]["f1"].tobytes()
        else:
            assert (y == res[q]["symbols"]).all()


def test_schedule_at_the_frame_edges(ctx, G, oracle):
    """Transmissions that start at sample ~15 or run past the end of the frame: the
    schedule's lags reach n <= 0 and n >= np (cc:205 skips them), which takes the
    bounds-checked loaders of k4_group / k4_ring / k4_tonecorr instead of their
    interior fast paths.  Hand-made candidates put the shifts right at the edges."""
    base = G.synth.make_frames(2, seed=4711, snr_db=-12.0)
    early = np.roll(base, -360, axis=1); early[:, -360:] = 0     # signal starts at sample 15
    late = np.roll(base, 3400, axis=1); late[:, :3400] = 0       # last symbols fall off the end
    frames = np.concatenate([early, late])
    fdr = oracle.FDR()
    cands = []
    for b in range(4):
        c = fdr.transform(frames[b])[:2].copy()
        assert len(c) >= 1
        extra = c[:1].copy()
        extra["shift"] = 0 if b < 2 else 3775 + 128              # S0 lags -128..128 / beyond the end
        drifting = c[:1].copy()                                   # a linear candidate WITH drift: per-symbol
        drifting["m_type"] = 0                                    # frequencies (k4_fstage's general path)
        raw = bytearray(drifting.tobytes()); raw[24:28] = np.float32(1.5).tobytes()
        drifting = np.frombuffer(bytes(raw), c.dtype).copy()
        cands.append(np.concatenate([c, extra, drifting]))
    per = max(len(c) for c in cands)
    out = ctx.demod_batch(frames, cands, max_per_frame=per)
    worth = 0
    for b in range(4):
        for j in range(len(cands[b])):
            d = oracle.demod_candidate(cands[b][j], 1500, frames[b])
            o = out[b, j]
            assert int(o["worth_a_try"]) == d["worth_a_try"] and int(o["shift1"]) == d["shift1"], (b, j)
            for k in ("f1", "drift1", "sync1"):
                assert np.float32(o[k]).tobytes() == np.float32(d[k]).tobytes(), (b, j, k)
            if d["worth_a_try"]:
                worth += 1
                assert (o["symbols"] == d["symbols"]).all()
                assert (o["jig_shift"] == d["jig_shift"]).all()
                assert o["jig_sync"].tobytes() == d["jig_sync"].tobytes()
    assert worth >= 4
    assert min(int(out[b, j]["shift1"]) for b in range(2) for j in range(len(cands[b]))) < 64


def test_schedule_on_a_silent_frame(ctx, G, oracle):
    """All-zero samples: every metric is 0/0 = NaN, no hypothesis ever beats the -1e30
    default (cc:227-231), the reference's 0 / 0.0 defaults land in the state and the
    candidate is not worth a try.  Also the path on which stage-winner reuse must stay off."""
    frames = np.zeros((2, 45000, 2), np.float32)
    frames[1] = G.synth.make_frames(1, seed=5, snr_db=-15.0)[0]      # a normal frame beside it
    fdr = oracle.FDR()
    real = fdr.transform(frames[1])[:1]
    ghost = real.copy()                                              # same parameters on the silent frame
    nl = real.copy()
    nl["m_type"] = 1; nl["V1"] = -1.0; nl["V2"] = 2.0; nl["p1"] = 0; nl["p2"] = 450
    cands = [np.concatenate([ghost, nl]), np.concatenate([real, nl])]
    out = ctx.demod_batch(frames, cands, max_per_frame=2)
    for b in range(2):
        for j in range(2):
            d = oracle.demod_candidate(cands[b][j], 1500, frames[b])
            o = out[b, j]
            assert int(o["worth_a_try"]) == d["worth_a_try"], (b, j)
            assert int(o["shift1"]) == d["shift1"], (b, j)
            for k in ("f1", "drift1", "sync1"):
                assert np.float32(o[k]).tobytes() == np.float32(d[k]).tobytes(), (b, j, k)
            if d["worth_a_try"]:
                assert (o["symbols"] == d["symbols"]).all()
    assert int(out[0, 0]["worth_a_try"]) == 0 and int(out[1, 0]["worth_a_try"]) == 1


def test_other_carrier_and_frame_parameters(G, oracle):
    """Parameters away from the flowgraph defaults: cf = 2000 and 3000 (the SLM Doppler reach,
    hence the coarse tile width, scales with cf), a wide band with few candidates kept
    (maxfreqs = 3), cf = 500, and carriers whose tile no longer fits LDS as float4 per centre:
    cf = 6000 (plain sqrt rows in LDS, four gathers per symbol) and cf = 24000 (reach +-64 bins:
    the rows live in an HBM scratch).  Candidates and the whole schedule against the oracle.
    A geometry the build does not implement is a status code, not a crash."""
    with pytest.raises(G.UwsprError) as ei:
        G.Context(spb=128)
    assert ei.value.status == -3   # UWSPR_ERR_UNSUPPORTED
    frames = G.synth.make_frames(3, seed=271828, snr_db=-16.0, halfbandwidth=20)
    for kw in ({"cf": 2000, "halfbandwidth": 20}, {"cf": 3000, "halfbandwidth": 20},
               {"halfbandwidth": 30, "maxfreqs": 3, "maxdrift": 1}, {"cf": 500, "halfbandwidth": 12},
               {"cf": 6000, "halfbandwidth": 20}, {"cf": 24000, "halfbandwidth": 20},
               {"fs": 400, "halfbandwidth": 12}):   # fs only moves the coarse bin width (FDR_impl.cc:93); the fine search keeps 375
        c = G.Context(**kw)
        try:
            cands, out = c.pipeline_batch(frames, max_per_frame=2)
        finally:
            c.close()
        f = oracle.FDR(**kw)
        cf = kw.get("cf", 1500)
        for b in range(3):
            exp = f.transform(frames[b])
            assert len(cands[b]) == len(exp), (kw, b)
            for j, (a, e) in enumerate(zip(cands[b], exp)):
                cand_equal(a, e)
                if j < 2:
                    d = oracle.demod_candidate(e, cf, frames[b])
                    o = out[b, j]
                    assert int(o["worth_a_try"]) == d["worth_a_try"] and int(o["shift1"]) == d["shift1"], (kw, b, j)
                    for k in ("f1", "drift1", "sync1"):
                        assert np.float32(o[k]).tobytes() == np.float32(d[k]).tobytes(), (kw, b, j, k)
                    if d["worth_a_try"]:
                        assert (o["symbols"] == d["symbols"]).all(), (kw, b, j)


@pytest.mark.parametrize("fl", [45056, 48000])
def test_other_frame_lengths(G, oracle, fl):
    """fl is a parameter of the reference's blocks: the row count of the spectrogram follows it
    (FDR_impl.cc:118), the sample bound of the fine search does NOT -- `npoints = 45000`
    (sync_and_demodulate_impl.cc:92) is what every sync_and_demodulate() call gets as np (cc:413-465),
    so with fl > 45000 the fine search ignores the samples from 45 000 on.
    Two frames longer than the flowgraph's 45 000 samples (a shorter one has fewer than the 348 rows the
    coarse search indexes: UWSPR_ERR_UNSUPPORTED; the reference reads out of bounds): spectrogram band,
    candidates and the whole schedule against the oracle (which passes np = 45000 as the reference does)."""
    base = G.synth.make_frames(3, seed=424242, snr_db=-17.0)
    if fl <= base.shape[1]:
        fr = np.ascontiguousarray(base[:, :fl])
    else:
        tail = (0.35 * np.random.default_rng(11).standard_normal((3, fl - base.shape[1], 2))).astype(np.float32)
        fr = np.ascontiguousarray(np.concatenate([base, tail], axis=1))
    c = G.Context(fl=fl)
    try:
        cands, out = c.pipeline_batch(fr, max_per_frame=2)
        ps = c.fdr_spectrum(3)[0]
        lo, w = c.info.band_lo, c.info.band_w
    finally:
        c.close()
    f = oracle.FDR(fl=fl)
    for b in range(3):
        assert ps[b].tobytes() == f.spectrogram(fr[b])[:, lo:lo + w].tobytes(), (fl, b)
        exp = f.transform(fr[b])
        assert len(cands[b]) == len(exp) >= 1, (fl, b)
        for j, (a, e) in enumerate(zip(cands[b], exp)):
            cand_equal(a, e)
            if j < 2:
                d = oracle.demod_candidate(e, 1500, fr[b])
                o = out[b, j]
                assert int(o["worth_a_try"]) == d["worth_a_try"] and int(o["shift1"]) == d["shift1"], (fl, b, j)
                for k in ("f1", "drift1", "sync1"):
                    assert np.float32(o[k]).tobytes() == np.float32(d[k]).tobytes(), (fl, b, j, k)
                if d["worth_a_try"]:
                    assert (o["symbols"] == d["symbols"]).all(), (fl, b, j)


def _late_frames(G, fl, start, n=2, seed=515151, snr_db=-14.0):
    """Frames of fl samples whose transmission starts at sample `start` (so that its last symbols lie
    beyond sample 45 000), noise everywhere."""
    base = G.synth.make_frames(n, seed=seed, snr_db=None)
    sig = base[:, G.synth.START:G.synth.START + 162 * 256]
    rng = np.random.default_rng(seed)
    fr = (G.synth.sigma_for_snr(snr_db) * rng.standard_normal((n, fl, 2))).astype(np.float32)
    fr[:, start:start + sig.shape[1]] += sig
    return np.ascontiguousarray(fr)


def test_fine_search_ignores_samples_from_45000_on(G, oracle):
    """npoints = 45000 (sync_and_demodulate_impl.cc:92, the np of cc:205's `n < np`) with fl = 48 000 and
    windows that reach past sample 45 000: a transmission starting at sample 5 000 ends at 46 472, so
    its last six symbols are (partly) invisible to the fine search.  Through every path that takes a
    lag -- the reference-signature calls (modes 0/1/2), the flat sweep, the grid sweep and the
    schedule (both forms) with hand-made candidates at shift = 5 000 -- the GPU equals the oracle run
    with np = 45 000 and DIFFERS from the oracle run with np = fl (the samples are there and are not
    zero, so the test can tell the two apart)."""
    import os
    fl, start = 48000, 5000
    fr = _late_frames(G, fl, start)
    lin = np.zeros(1, oracle.CAND_DTYPE)[0]
    # mode-2 metrics with np = 45000 / np = fl differ, or the test proves nothing
    s45, _, _, y45 = oracle.sync_and_demodulate(lin, 1500, fr[0], 0.0, 0, 0, 0.0, start, 0, 0, 1, 0.0, 50, 2)
    sfl, _, _, yfl = oracle.sync_and_demodulate(lin, 1500, fr[0], 0.0, 0, 0, 0.0, start, 0, 0, 1, 0.0, 50, 2,
                                                np_points=fl)
    assert np.float32(s45).tobytes() != np.float32(sfl).tobytes() and (y45 != yfl).any()

    c = G.Context(fl=fl)
    try:
        # (1) the reference-signature call form, all three modes, lags around the late start
        calls = np.zeros(6, G.native.CALL_DTYPE)
        k = 0
        for b in range(2):
            for mode in (0, 1, 2):
                q = calls[k]
                q["frame"] = b; q["f1"] = 0.0; q["ifmin"] = -2; q["ifmax"] = 2; q["fstep"] = 0.25
                q["shift1"] = start; q["lagmin"] = start - 128; q["lagmax"] = start + 128; q["lagstep"] = 64
                q["drift1"] = 0.0; q["symfac"] = 50; q["mode"] = mode
                k += 1
        res = c.sync_and_demodulate(fr, calls)
        for q, r in zip(calls, res):
            s, sh, f1, y = oracle.sync_and_demodulate(lin, 1500, fr[int(q["frame"])], float(q["f1"]), int(q["ifmin"]),
                                                      int(q["ifmax"]), float(q["fstep"]), int(q["shift1"]),
                                                      int(q["lagmin"]), int(q["lagmax"]), int(q["lagstep"]),
                                                      float(q["drift1"]), 50, int(q["mode"]))
            assert np.float32(r["sync"]).tobytes() == np.float32(s).tobytes(), int(q["mode"])
            assert int(r["shift1"]) == sh and np.float32(r["f1"]).tobytes() == np.float32(f1).tobytes()
            if int(q["mode"]) == 2:
                assert (r["symbols"] == y).all()
        # (2) flat sweep and grid sweep around a centre at the late start
        cent = np.zeros(2, G.native.CAND_DTYPE)
        cent["shift"] = start
        hy = G.sweep_grid(cent, [0, 1])
        sync, sym = c.sync_sweep(fr, hy, soft=True)
        for qi in range(0, len(hy), 37):
            h = hy[qi]
            s, _, _, y = oracle.sync_and_demodulate(lin, 1500, fr[int(h["frame"])], float(h["f0"]), 0, 0, 0.0,
                                                    int(h["lag"]), 0, 0, 1, float(h["drift"]), 50, 2)
            assert np.float32(sync[qi]).tobytes() == np.float32(s).tobytes() and (sym[qi] == y).all(), qi
        from gr_uwspr_amd import sweep as SW
        gs, gy = c.sync_grid(fr, cent, np.array(SW.DF_STEPS, np.float32) * np.float32(0.25),
                             np.array(SW.DRIFTS, np.float32), np.array(SW.LAGS, np.int32))
        flat = sync.reshape(2, 5, 8, 5).transpose(0, 1, 3, 2)      # flat order is (freq, lag, drift)
        assert gs.reshape(2, 5, 5, 8).tobytes() == np.ascontiguousarray(flat).tobytes()
        # (3) the schedule, both forms, from hand-made candidates at the late start
        cands = [np.zeros(1, oracle.CAND_DTYPE) for _ in range(2)]
        for b in range(2):
            cands[b][0]["shift"] = start
            cands[b][0]["sync"] = 0.5
        outs = {}
        for fused in ("1", "0"):
            c2 = G.Context(fl=fl, options={"sched": int(fused)})
            try:
                outs[fused] = c2.demod_batch(fr, cands, max_per_frame=1)
            finally:
                c2.close()
        assert outs["1"].tobytes() == outs["0"].tobytes()
        for b in range(2):
            d = oracle.demod_candidate(cands[b][0], 1500, fr[b])
            dfl = oracle.demod_candidate(cands[b][0], 1500, fr[b], np_points=fl)
            o = outs["1"][b, 0]
            assert int(o["worth_a_try"]) == d["worth_a_try"] == 1 and int(o["shift1"]) == d["shift1"]
            for k in ("f1", "drift1", "sync1"):
                assert np.float32(o[k]).tobytes() == np.float32(d[k]).tobytes(), (b, k)
            assert (o["symbols"] == d["symbols"]).all() and (o["jig_sync"] == d["jig_sync"]).all()
            assert (d["symbols"] != dfl["symbols"]).any()      # ... and np = fl would have been visible
    finally:
        c.close()


@pytest.mark.parametrize("threshold", [0, 1, 3, 10, 50, 10 ** 6])
def test_coarse_search_pruning_is_exact_for_every_threshold(G, oracle, threshold):
    """K3 evaluates the nonlinear hypotheses only for the cells before the first linear metric
    >= 1.001/threshold (no nonlinear acceptance is possible after it: |sync| <= 1, rule cc:392).
    The candidates must equal the oracle's -- which evaluates everything -- for thresholds that
    prune nothing (0, 1), some (3, 10, 50) or almost everything (1e6), on strong, weak and
    signal-free frames, without and with linear drift hypotheses (maxdrift = 4: the 1 170 linear
    sequences alone need more than one round of the kernel)."""
    fr = np.concatenate([G.synth.make_frames(3, seed=31337, snr_db=-16.0),
                         G.synth.make_frames(3, seed=31338, snr_db=-28.0),
                         (0.5 * np.random.default_rng(5).standard_normal((2, 45000, 2))).astype(np.float32)])
    for maxdrift in (0, 2, 4):
        c = G.Context(threshold=threshold, maxdrift=maxdrift)
        try:
            got = c.fdr_batch(fr)
        finally:
            c.close()
        f = oracle.FDR(threshold=threshold, maxdrift=maxdrift)
        for b in range(fr.shape[0]):
            exp = f.transform(fr[b])
            assert len(got[b]) == len(exp), (threshold, maxdrift, b)
            for a, e in zip(got[b], exp):
                cand_equal(a, e)


def test_coarse_search_tile_forms_agree(G, frames, monkeypatch):
    """K3's three tile forms (float4 per centre in LDS -- the default --, plain sqrt rows in LDS,
    sqrt rows in HBM: option k3_tile = 0 / 1 / 2) and a padded row pitch (k3_pitch): identical
    candidates and identical 16 380 metrics per candidate."""
    res = []
    for name, v in (("k3_tile", "0"), ("k3_tile", "1"), ("k3_tile", "2"), ("k3_pitch", "24")):
        monkeypatch.setenv("UWSPR_OPTIONS", name + "=" + v)      # (shape the context when it is created)
        c = G.Context()
        try:
            c.keep_syncgrid(2)
            cands = c.fdr_batch(frames)
            grid = c.fdr_syncgrid(len(frames))
        finally:
            c.close()
        res.append((cands, grid))
    ca, ga = res[0]
    for cb, gb in res[1:]:
        for b in range(len(frames)):
            assert len(ca[b]) == len(cb[b]) >= 1
            for j, (x, y) in enumerate(zip(ca[b], cb[b])):
                for k in ("m_type", "freq", "snr", "sync", "shift"):
                    assert x[k].tobytes() == y[k].tobytes(), (b, j, k)
                if int(x["m_type"]) == 1:
                    assert all(x[k] == y[k] for k in ("V1", "V2", "p1", "p2"))
                else:
                    assert x.tobytes()[24:28] == y.tobytes()[24:28]
                if j < 2:
                    assert ga[b, j].tobytes() == gb[b, j].tobytes()


@pytest.mark.parametrize("fused", ["1", "0"])
def test_lazy_tries_and_resume_equal_the_eager_schedule(G, frames, vec, monkeypatch, fused):
    """uwspr_set_tries(k) + uwspr_demod_resume: the reference stops at its first decoding try
    (cc:457-490).  Tries idt < k of the lazy pass are byte-identical to the eager ones, the
    rest is zero; after the resume the flagged records equal the eager records byte for byte
    and the unflagged ones are untouched.  Host and device pointer forms; both schedule forms
    (the staged form runs stage 5 on the k wanted tries only and is resumed by the fused kernel)."""
    import torch
    cands = [vec["cands"][b, :int(vec["npk"][b])] for b in range(4)]
    per = max(len(c) for c in cands)
    c = G.Context(options={"sched": int(fused)})
    try:
        eager = c.demod_batch(frames, cands, max_per_frame=per)
        for k in (1, 3):
            c.set_tries(k)
            lazy = c.demod_batch(frames, cands, max_per_frame=per)
            for f in ("f1", "drift1", "sync1", "shift1", "worth_a_try"):
                assert lazy[f].tobytes() == eager[f].tobytes()
            assert lazy["symbols"][:, :, :k].tobytes() == eager["symbols"][:, :, :k].tobytes()
            assert lazy["jig_sync"][:, :, :k].tobytes() == eager["jig_sync"][:, :, :k].tobytes()
            assert lazy["jig_rms"][:, :, :k].tobytes() == eager["jig_rms"][:, :, :k].tobytes()
            assert not lazy["symbols"][:, :, k:].any() and not lazy["jig_sync"][:, :, k:].any()
            need = np.zeros((4, per), np.uint8)
            need[1, :] = 1
            need[3, 0] = 1
            res = c.demod_resume(frames, need, None, max_per_frame=per)
            for b in range(4):
                for j in range(per):
                    want = eager[b, j] if need[b, j] else lazy[b, j]
                    assert res[b, j].tobytes() == want.tobytes(), (k, b, j)
        # device pointers, through the whole pipeline
        c.set_tries(1)
        dev = torch.from_numpy(frames).cuda()
        cd = torch.empty(4 * c.maxfreqs * 48, dtype=torch.uint8, device="cuda")
        nd = torch.empty(4, dtype=torch.int32, device="cuda")
        od = torch.empty(4 * G.native.DEMOD_DTYPE.itemsize, dtype=torch.uint8, device="cuda")
        c.pipeline_batch_into(dev, cd, nd, od, max_per_frame=1)
        needd = torch.ones(4, dtype=torch.uint8, device="cuda")
        c.demod_resume(dev, needd, od, max_per_frame=1)
        c.synchronize()
        got = np.frombuffer(od.cpu().numpy().tobytes(), G.native.DEMOD_DTYPE)
        c.set_tries(17)
        _, full = c.pipeline_batch(frames, max_per_frame=1)
        assert got.tobytes() == full[:, 0].tobytes()
    finally:
        c.close()


def test_resume_state_is_checked_and_eager_passes_keep_no_winner_magnitudes(G, frames, vec):
    """One context through: a lazy pass on a small batch (the winner-magnitude buffer is sized for it),
    an EAGER pass on a larger one (must not write that buffer: it would run past its end), a lazy pass
    again.  uwspr_demod_resume is refused after an eager pass (stale magnitudes), after a first pass
    whose records went somewhere the resume would not patch, and for another batch shape."""
    import torch
    big = np.concatenate([frames, frames[::-1], frames])          # 12 frames
    c = G.Context()
    try:
        _, eager_big = c.pipeline_batch(big, max_per_frame=2)
        c.set_tries(1)
        _, lazy_small = c.pipeline_batch(frames[:1], max_per_frame=1)
        c.set_tries(17)
        _, again = c.pipeline_batch(big, max_per_frame=2)          # eager, 24 slots > the 1 slot kept
        assert again.tobytes() == eager_big.tobytes()
        with pytest.raises(G.UwsprError):                          # the last schedule call was eager
            c.demod_resume(big, np.ones((12, 2), np.uint8), None, max_per_frame=2)
        c.set_tries(1)
        _, lazy_big = c.pipeline_batch(big, max_per_frame=2)
        with pytest.raises(G.UwsprError):                          # another batch shape
            c.demod_resume(big[:4], np.ones((4, 2), np.uint8), None, max_per_frame=2)
        res = c.demod_resume(big, np.ones((12, 2), np.uint8), None, max_per_frame=2)
        assert res.tobytes() == eager_big.tobytes()
        # first pass with host records (they live in the context): a device resume has nothing to patch
        dev = torch.from_numpy(big).cuda()
        od = torch.zeros(24 * G.native.DEMOD_DTYPE.itemsize, dtype=torch.uint8, device="cuda")
        c.pipeline_batch(big, max_per_frame=2)
        with pytest.raises(G.UwsprError):
            c.demod_resume(dev, torch.ones(24, dtype=torch.uint8, device="cuda"), od, max_per_frame=2)
        # first pass into the caller's device buffer: a host resume would patch records it never wrote
        cd = torch.empty(12 * c.maxfreqs * 48, dtype=torch.uint8, device="cuda")
        nd = torch.empty(12, dtype=torch.int32, device="cuda")
        c.pipeline_batch_into(dev, cd, nd, od, max_per_frame=2)
        with pytest.raises(G.UwsprError):
            c.demod_resume(big, np.ones((12, 2), np.uint8), None, max_per_frame=2)
        c.demod_resume(dev, torch.ones(24, dtype=torch.uint8, device="cuda"), od, max_per_frame=2)
        c.synchronize()
        assert od.cpu().numpy().tobytes() == eager_big.tobytes()
    finally:
        c.close()


def test_stream_push_frames_equal_whole_frame_calls(G, oracle):
    """uwspr_stream_*: a continuous stream pushed in ragged pieces, frames cut on the device every
    3375 samples (sliding_window_stream_to_pdu::work, cc:113-135): every sample is uploaded once,
    and the frames -- hence every byte the pipeline produces from them -- equal the whole-frame
    calls on host frames sliced from the same stream."""
    import torch
    hop, fl = 3375, 45000
    nfr = 6
    base = G.synth.make_frames(2, seed=4242, snr_db=-18.0)
    stream = np.concatenate([base[0], base[1][: (nfr - 1) * hop + 1000]], axis=0)   # fl + 5 hops + a bit
    want = np.stack([stream[k * hop: k * hop + fl] for k in range(nfr)])
    c = G.Context()
    try:
        ref_c, ref_o = c.pipeline_batch(want, max_per_frame=2)
        c.stream_open(hop, 4)
        got = torch.empty((nfr, fl, 2), dtype=torch.float32, device="cuda")
        rng = np.random.default_rng(1)
        pos, taken = 0, 0
        while taken < nfr:
            n = int(rng.integers(1, 9000))
            ready = c.stream_push(stream[pos: pos + n])
            pos += min(n, len(stream) - pos)
            while ready > 0 and taken < nfr:
                k = min(ready, nfr - taken, 1 + taken % 3)
                first = c.stream_take(k, got[taken: taken + k])
                assert first == taken * hop
                taken += k
                ready = c.stream_push(stream[0:0])
        c.synchronize()
        assert got.cpu().numpy().tobytes() == want.tobytes()
        cd, od = c.pipeline_batch(got, max_per_frame=2)
        for a, b in zip(ref_c, cd):
            assert a.tobytes() == b.tobytes()
        assert od.tobytes() == ref_o.tobytes()
        # frames on the device, records on the host (UWSPR_DEVICE_FRAMES)
        cands = np.zeros((nfr, c.maxfreqs), G.native.CAND_DTYPE)
        npk = np.zeros(nfr, np.int32)
        import ctypes as C
        c._chk(c.L.uwspr_fdr_batch(c.h, C.c_void_p(got.data_ptr()), nfr, G.native.DEVICE_FRAMES,
                                   C.c_void_p(cands.ctypes.data), C.c_void_p(npk.ctypes.data)))
        for b in range(nfr):
            assert cands[b, :npk[b]].tobytes() == ref_c[b].tobytes()
    finally:
        c.close()


def test_frames_in_place_equal_whole_frame_calls(G, oracle):
    """Frames read where they lie (uwspr_stream_take_view + uwspr_set_frame_stride): frame j of a take
    starts j * hop samples after the first, nothing is cut out.  A stream long enough that the ring
    moves its tail twice (the uploads run on their own copy stream), taken in ragged groups; every
    byte the pipeline produces from the views -- eager, lazy + resume, the flat sweep -- equals the
    whole-frame calls on host frames sliced from the same stream; a page-locked source rewritten right
    after uwspr_stream_push(UWSPR_HOST) returns must not disturb the stream (the call waits for its
    DMA); the same stretch handed over as ONE strided host buffer gives the same bytes as well."""
    import torch
    hop, fl, nfr, maxf = 3375, 45000, 41, 3
    base = G.synth.make_frames(5, seed=777, snr_db=-17.0)
    stream = np.concatenate([base[k][: (10 * hop if k < 4 else fl)] for k in range(5)], axis=0)
    stream = np.ascontiguousarray(stream[: fl + (nfr - 1) * hop])
    assert len(stream) == fl + (nfr - 1) * hop
    want = np.stack([stream[k * hop: k * hop + fl] for k in range(nfr)])
    c = G.Context()
    try:
        ref_c, ref_o = c.pipeline_batch(want, max_per_frame=2)
        c.stream_open(hop, maxf)
        pinned = torch.empty((6000, 2), dtype=torch.float32).pin_memory()
        rng = np.random.default_rng(2)
        pos, taken = 0, 0
        got_c, got_o = [], []
        while taken < nfr:
            n = min(int(rng.integers(1, 6000)), len(stream) - pos)
            if n > 0 and (taken % 2 == 0):
                pinned[:n] = torch.from_numpy(stream[pos: pos + n])
                ready = c.stream_push(pinned[:n].numpy())
                pinned[:n] = 7.0                      # the source is ours again when the call returns
            else:
                ready = c.stream_push(stream[pos: pos + n])
            pos += n
            while ready > 0 and taken < nfr:
                k = min(ready, nfr - taken, 1 + taken % maxf)
                ptr, stride, first = c.stream_take_view(k)
                assert first == taken * hop and stride == hop
                c.set_frame_stride(stride)
                view = G.FrameView(k, ptr=ptr)
                if taken % 4 == 1:                    # lazy first pass + resume of everything, device records
                    c.set_tries(1)
                    cands_t = torch.empty(k * c.maxfreqs * 48, dtype=torch.uint8, device="cuda")
                    npk_t = torch.empty(k, dtype=torch.int32, device="cuda")
                    out_t = torch.empty(k * 2 * G.native.DEMOD_DTYPE.itemsize, dtype=torch.uint8, device="cuda")
                    c.pipeline_batch_into(view, cands_t, npk_t, out_t, max_per_frame=2)
                    c.demod_resume(view, torch.ones((k, 2), dtype=torch.uint8, device="cuda"), out_t, max_per_frame=2)
                    c.synchronize()
                    c.set_tries(17)
                    cn = np.frombuffer(cands_t.cpu().numpy().tobytes(), G.native.CAND_DTYPE).reshape(k, -1)
                    nn = npk_t.cpu().numpy()
                    cd = [cn[b, :nn[b]].copy() for b in range(k)]
                    od = np.frombuffer(out_t.cpu().numpy().tobytes(), G.native.DEMOD_DTYPE).reshape(k, 2).copy()
                else:
                    cd, od = c.pipeline_batch(view, max_per_frame=2)
                if taken % 5 == 0:                    # the flat sweep reads the same view
                    hy = G.sweep_grid(np.array([ref_c[taken][0]]), [0])[:40]
                    s1, y1 = c.sync_sweep(view, hy, soft=True)
                    c.set_frame_stride(0)
                    s2, y2 = c.sync_sweep(want[taken: taken + 1], hy, soft=True)
                    assert s1.tobytes() == s2.tobytes() and y1.tobytes() == y2.tobytes()
                c.set_frame_stride(0)
                got_c += list(cd)
                got_o.append(od)
                taken += k
                ready = c.stream_push(stream[0:0])
        for a, b in zip(ref_c, got_c):
            assert a.tobytes() == b.tobytes()
        assert np.concatenate(got_o).tobytes() == ref_o.tobytes()
        # one strided host buffer: the span is uploaded once
        c.set_frame_stride(hop)
        cd, od = c.pipeline_batch(G.FrameView(8, host=stream[: fl + 7 * hop]), max_per_frame=2)
        c.set_frame_stride(0)
        for a, b in zip(ref_c[:8], cd):
            assert a.tobytes() == b.tobytes()
        assert od.tobytes() == ref_o[:8].tobytes()
    finally:
        c.close()


def test_two_contexts_with_different_coarse_tiles(G, oracle, frames):
    """The coarse-search kernel's dynamic-LDS limit is a property of the function, not of a context:
    a second context with a smaller tile (cf = 500) must not lower it under an earlier one (defaults),
    and a larger one (hbw = 40, maxdrift = 2) created later must work as well."""
    a = G.Context()
    b = G.Context(cf=500)
    c = G.Context(halfbandwidth=40, maxdrift=2)
    try:
        for ctx, kw in ((a, {}), (c, {"halfbandwidth": 40, "maxdrift": 2}), (b, {"cf": 500}), (a, {})):
            got = ctx.fdr_batch(frames[:2])
            f = oracle.FDR(**kw)
            for bb in range(2):
                exp = f.transform(frames[bb])
                assert len(got[bb]) == len(exp)
                for x, e in zip(got[bb], exp):
                    cand_equal(x, e)
    finally:
        a.close(); b.close(); c.close()


def test_schedule_forms_agree_on_candidates_that_stage_2_gives_a_drift(G, oracle):
    """Stage 2 (cc:421-441) can give ANY candidate a drift of +-0.5 Hz -- also one from an FDR with maxdrift = 0 -- and from
    there on its tone frequency depends on the symbol: no phasor table, per-lane recurrences in S3, S4 and S5.  Round 5's
    first wiring of the register-ring kernel left such groups to a second launch that is skipped after the context's own
    FDR with maxdrift = 0 (a shortcut that is right for S0, whose candidates have no drift yet); the exact-vs-fast test
    caught it by luck.  Here: weak and noise-only frames through uwspr_pipeline_batch -- the FDR's own candidates -- in
    every schedule form, byte for byte, the drifting records counted, and (round 6) EVERY record of the staged form -- the one
    bench.py times -- against the oracle's demod_candidate (cc:421-441, 457-468), not only against the other forms."""
    fr = np.concatenate([G.synth.make_frames(120, seed=0xD21F7, snr_db=-27.0),
                         G.synth.make_frames(120, seed=0xD21F8, snr_db=-30.0),
                         (0.5 * np.random.default_rng(77).standard_normal((60, 45000, 2))).astype(np.float32)])
    outs = {}
    for name, opts in (("fused", {"sched": 1}), ("staged", {"sched": 0}), ("staged-lds-ring", {"sched": 0, "k4_forms": 0}),
                       ("staged-flat", {"sched": 0, "stage_kernels": 0, "phasor_tables": 0})):
        c = G.Context(options=opts)
        try:
            cands, out = c.pipeline_batch(fr, max_per_frame=2)
            outs[name] = (cands, out)
        finally:
            c.close()
    ref = outs["fused"][1]
    for name in ("staged", "staged-lds-ring", "staged-flat"):
        assert outs[name][1].tobytes() == ref.tobytes(), name
    drifting = worth = 0
    for b in range(len(fr)):
        for j in range(min(2, len(outs["fused"][0][b]))):
            r = ref[b, j]
            drifting += int(float(r["drift1"]) != 0.0)
            worth += int(r["worth_a_try"]) * int(float(r["drift1"]) != 0.0)
    print("%d records with a stage-2 drift, %d of them worth a try (their S5 runs without a table)" % (drifting, worth))
    assert drifting >= 20 and worth >= 3
    cands, staged = outs["staged"]
    checked = 0
    for b in range(len(fr)):
        for j in range(min(2, len(cands[b]))):
            d = oracle.demod_candidate(cands[b][j], 1500, fr[b])
            assert oracle.record_diff(staged[b, j], d) == [], (b, j, float(staged[b, j]["drift1"]))
            checked += int(d["drift1"] != 0.0)
    assert checked == drifting
