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
def test_pipe_argument_errors_are_not_sticky(G):
    """An argument error fails the call that made it (status + message) and nothing else: the pipe goes on and
    gives the records of the sequential calls."""
    import torch
    with pytest.raises(G.UwsprError):
        G.Pipe(hop=50000)
    fr = G.synth.make_frames(4, seed=99, snr_db=-18.0)
    ctx = G.Context()
    try:
        exp = _sequential(G, ctx, fr, 1)
    finally:
        ctx.close()
    pipe = G.Pipe(batch_frames=4)
    try:
        with pytest.raises(G.UwsprError) as e:
            pipe.acquire(4 * 45000 + 1)
        assert e.value.status == -6 and "at most" in str(e.value)      # UWSPR_ERR_ARG
        dev = torch.from_numpy(fr).cuda()
        with pytest.raises(G.UwsprError) as e:
            pipe.submit_device(dev, stride=-5)
        assert e.value.status == -6
        with pytest.raises(G.UwsprError):
            pipe.commit(10)                                   # no matching acquire
        pipe.submit_device(dev)                               # ... and the pipe still works
        pipe.flush()
        assert _as_dict(pipe.collect()) == exp
    finally:
        pipe.close()


@pytest.mark.gpu
@pytest.mark.parametrize("where", [0, 1])
def test_pipe_runtime_failure_mid_stream(G, where):
    """A runtime failure on batch 2 of 5 (injected at its launch / in its host tail): batches 0 and 1 emit their records
    as the sequential calls give them, batch 2 emits nothing, the status is sticky -- submit, flush and (once the good
    records are gone) collect return it -- and close returns."""
    import torch
    fr = G.synth.make_frames(20, seed=4711, snr_db=-18.0)
    ctx = G.Context()
    try:
        exp = _sequential(G, ctx, fr[:8], 1)
    finally:
        ctx.close()
    dev = torch.from_numpy(fr).cuda()
    pipe = G.Pipe(batch_frames=4, lanes=3)
    try:
        pipe.inject_failure(2, where)
        failed_at = None
        for k in range(5):
            try:
                pipe.submit_device(dev[4 * k: 4 * k + 4])
            except G.UwsprError as e:
                failed_at = k
                assert e.status == -4 and "injected" in str(e)
                break
        assert failed_at is None or failed_at >= 2                # (where = 1: the producer learns of it a call or two later, or in flush)
        assert where == 1 or failed_at == 2
        with pytest.raises(G.UwsprError) as e:
            pipe.flush()
        assert e.value.status == -4
        recs = pipe.collect()                                     # what was emitted before the failure stays collectable
        got = _as_dict(recs)
        assert set(int(f) for f in recs["frame"]) >= set(range(8)) and not (set(int(f) for f in recs["frame"]) & {8, 9, 10, 11})
        assert {k: v for k, v in got.items() if k[0] < 8} == exp
        with pytest.raises(G.UwsprError):
            pipe.collect(wait=True)                               # nothing left: the status
        with pytest.raises(G.UwsprError):
            pipe.submit_device(dev[:4])
    finally:
        pipe.close()                                              # returns


@pytest.mark.gpu
def test_pipe_stream_longer_than_the_device_ring(G):
    """A pushed stream several times the ring's capacity (batches of 2 frames, 3 lanes: (3 + 3) * 2 * 3375 + 45000 =
    85 500 samples per buffer; 320 frames = 1.12 M samples): the tail moves between the two buffers a dozen times
    while earlier views are still being read, marginal transmissions are resumed from frames the ring has moved
    past -- byte-equal to the sequential calls on the same windows."""
    hop, fl, nfr, per = 3375, 45000, 320, 1
    base = G.synth.make_frames(28, seed=2718, snr_db=-26.5)
    stream = np.concatenate([base[k][: 12 * hop] for k in range(28)], axis=0)
    stream = np.ascontiguousarray(stream[: fl + (nfr - 1) * hop])
    assert len(stream) == fl + (nfr - 1) * hop
    want = np.stack([stream[k * hop: k * hop + fl] for k in range(nfr)])
    ctx = G.Context()
    try:
        exp = _sequential(G, ctx, want, per)
    finally:
        ctx.close()
    pipe = G.Pipe(hop=hop, batch_frames=2, max_per_frame=per, lanes=3)
    try:
        rng = np.random.default_rng(17)
        pos = 0
        while pos < len(stream):
            n = min(int(rng.integers(500, 2 * hop)), len(stream) - pos)
            pipe.push(stream[pos: pos + n])
            pos += n
        pipe.flush()
        recs = pipe.collect()
        st = pipe.stats()
    finally:
        pipe.close()
    assert st["frames"] == nfr and st["batches"] == nfr // 2
    assert (recs["stream_pos"] == recs["frame"] * hop).all()
    assert _as_dict(recs) == exp
    assert st["resumed"] >= 1 and st["decoded"] >= 10


@pytest.mark.gpu
def test_pipe_options_reach_every_lane(G):
    """uwspr_pipe_set_option: fast_search through the pipe (staged lanes) keeps every decision and message of the exact
    pipe on the same frames (what the option promises: tests/test_gpu_fast_search.py); it is refused while batches are
    in flight and for "sched", and an unknown name is an argument error that leaves the pipe usable."""
    import torch
    frames = _mixed_frames(G, 8, seed=6161)                  # 32 frames
    dev = torch.from_numpy(frames).cuda()
    torch.cuda.synchronize()

    def run(fast):
        pipe = G.Pipe(batch_frames=16, max_per_frame=1, lanes=3, sched="staged")
        try:
            if fast:
                pipe.set_option("fast_search", 1)
            with pytest.raises(G.UwsprError):
                pipe.set_option("sched", 1)
            with pytest.raises(G.UwsprError):
                pipe.set_option("no_such_option", 1)
            pipe.submit_device(dev[:16])
            pipe.submit_device(dev[16:])
            pipe.flush()
            recs = pipe.collect()
            pipe.set_option("reuse", 1)                      # nothing in flight any more: accepted
            return recs
        finally:
            pipe.close()

    exact, fast = run(False), run(True)
    assert len(exact) == len(fast) > 20
    for k in ("frame", "cand", "npk", "shift1", "worth_a_try", "decoded", "idt"):
        assert (exact[k] == fast[k]).all(), k
    assert exact["message"].tobytes() == fast["message"].tobytes()
    assert exact["f1"].tobytes() == fast["f1"].tobytes() and exact["drift1"].tobytes() == fast["drift1"].tobytes()
    assert np.all(np.abs(fast["sync1"] - exact["sync1"]) <= 1e-5 * np.maximum(np.abs(exact["sync1"]), 0.1))
    assert int(exact["decoded"].sum()) >= 8
