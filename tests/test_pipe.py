"""uwspr_pipe_*: the pipelined end-to-end decoder (copy-stream ingest || lazy schedule || Fano on the
persistent host pool || resume).  Its records must be those of the sequential calls -- the whole path
with all 17 tries (uwspr_pipeline_batch) followed by uwspr_decode_batch -- field for field, message
byte for byte, in frame order: only the order in time of the stages of consecutive batches may differ
(sync_and_demodulate_impl.cc:457-490: the reference stops at its first decoding try; the lazy flow
produces the later tries only for candidates whose first one did not decode)."""
import numpy as np
import pytest

FIELDS = ("f1", "drift1", "sync1", "shift1", "worth_a_try")


def _sequential(G, ctx, frames, per):
    """-> {(frame, cand): (npk, coarse bytes, fields, decoded, idt, message bytes)}"""
    cands, out = ctx.pipeline_batch(frames, max_per_frame=per)
    recs = out.reshape(-1)
    msgs, idts, ok = G.decode_batch(recs)
    exp = {}
    for b in range(len(cands)):
        for j in range(min(per, len(cands[b]))):
            i = b * per + j
            exp[(b, j)] = (len(cands[b]), cands[b][j].tobytes(),
                           tuple(recs[i][k].tobytes() for k in FIELDS), int(ok[i]),
                           int(idts[i]) if ok[i] else -1, msgs[i].tobytes() if ok[i] else bytes(7))
    return exp


def _as_dict(recs):
    got = {}
    for r in recs:
        got[(int(r["frame"]), int(r["cand"]))] = (int(r["npk"]), r["coarse"].tobytes(),
                                                 tuple(r[k].tobytes() for k in FIELDS), int(r["decoded"]),
                                                 int(r["idt"]), r["message"].tobytes())
    return got


def _mixed_frames(G, n, seed):
    """strong, marginal (first try fails now and then) and noise-only frames, interleaved"""
    parts = [G.synth.make_frames(n, seed=seed, snr_db=-19.0),
             G.synth.make_frames(n, seed=seed + 100, snr_db=-27.5),
             G.synth.make_frames(n, seed=seed + 200, snr_db=-30.0),
             (0.4 * np.random.default_rng(seed).standard_normal((n, 45000, 2))).astype(np.float32)]
    fr = np.stack(parts, axis=1).reshape(-1, 45000, 2)
    return np.ascontiguousarray(fr)


@pytest.mark.gpu
@pytest.mark.parametrize("eager", [False, True])
def test_pipe_on_device_frames_equals_the_sequential_calls(G, eager):
    import torch
    per = 2
    frames = _mixed_frames(G, 10, seed=9090)                 # 40 frames
    ctx = G.Context()
    try:
        exp = _sequential(G, ctx, frames, per)
    finally:
        ctx.close()
    dev = torch.from_numpy(frames).cuda()
    torch.cuda.synchronize()
    pipe = G.Pipe(batch_frames=16, max_per_frame=per, lanes=2, eager=eager)
    try:
        got = []
        for s in range(0, 40, 16):
            pipe.submit_device(dev[s: s + 16])
            got.append(pipe.collect())                        # whatever is finished, without waiting
        pipe.flush()
        got.append(pipe.collect())
        st = pipe.stats()
    finally:
        pipe.close()
    recs = np.concatenate(got)
    assert (np.diff(recs["frame"] * per + recs["cand"]) > 0).all()          # frame order
    assert (recs["stream_pos"] == -1).all()
    assert _as_dict(recs) == exp
    ndec = sum(v[3] for v in exp.values())
    assert st["frames"] == 40 and st["batches"] == 3 and st["decoded"] == ndec >= 10
    late = sum(1 for v in exp.values() if v[3] and v[4] > 0)
    if eager:
        assert st["resumed"] == 0
    else:
        assert st["resumed"] >= late                           # every late decode went through the resume


@pytest.mark.gpu
@pytest.mark.parametrize("lanes", [3, 0])
def test_pipe_on_a_pushed_stream_equals_the_sequential_calls(G, lanes):
    """A 375 S/s stream pushed in ragged pieces (pageable memory through uwspr_pipe_push, page-locked
    through acquire / commit): frames every 3375 samples, batches of 8, a short last batch on flush.
    lanes = 0: the library default (three streams + six spare lanes on the same streams), with the spares opened as
    soon as the three base lanes are busy (the 2.5 ms wait for a slow host tail switched off): same records, same order."""
    spare_after_us = 1 if lanes == 0 else 0
    hop, fl, nfr, per = 3375, 45000, 27, 1
    base = G.synth.make_frames(4, seed=31415, snr_db=-18.0)
    stream = np.concatenate([base[k][: 10 * hop] for k in range(4)], axis=0)
    stream = np.ascontiguousarray(stream[: fl + (nfr - 1) * hop])
    assert len(stream) == fl + (nfr - 1) * hop
    want = np.stack([stream[k * hop: k * hop + fl] for k in range(nfr)])
    ctx = G.Context()
    try:
        exp = _sequential(G, ctx, want, per)
    finally:
        ctx.close()
    pipe = G.Pipe(hop=hop, batch_frames=8, max_per_frame=per, lanes=lanes, spare_after_us=spare_after_us)
    try:
        rng = np.random.default_rng(5)
        pos, k = 0, 0
        while pos < len(stream):
            n = min(int(rng.integers(1, 20000)), len(stream) - pos)
            if k % 2:
                pipe.push(stream[pos: pos + n])
            else:
                buf = pipe.acquire(n)
                buf[:] = stream[pos: pos + n]
                pipe.commit(n)
            pos += n
            k += 1
        pipe.flush()
        recs = pipe.collect()
        st = pipe.stats()
    finally:
        pipe.close()
    assert st["frames"] == nfr and st["batches"] == 4
    assert (recs["stream_pos"] == recs["frame"] * hop).all()
    assert _as_dict(recs) == exp
    assert sum(v[3] for v in exp.values()) >= 3                # the transmissions decode in their own windows


@pytest.mark.gpu
def test_pipe_argument_errors(G):
    with pytest.raises(G.UwsprError):
        G.Pipe(hop=50000)
    pipe = G.Pipe(batch_frames=4)
    try:
        with pytest.raises(G.UwsprError):
            pipe.acquire(4 * 3375 + 1)
    finally:
        pipe.close()
