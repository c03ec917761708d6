"""Full BASELINE.json sizes on the GPU, checked through size-independent properties
and oracle digests (the oracle itself needs minutes at these sizes, so its
outputs travel as sha256 digests made by tests/golden/make_config3_sha.py).

  configs[2]  1024 frames x 200 (freq, lag, drift) hypotheses: flat and grid forms,
              symbol bytes and metrics against the oracle's digest
  configs[3]  65536 frames sharded round-robin over 8 (virtual) ranks, slabs
              gathered and put back in order == the unsharded run
"""
import hashlib
import json
import os
from concurrent.futures import ThreadPoolExecutor

import numpy as np
import pytest

from conftest import GOLDEN

pytestmark = pytest.mark.gpu
RTOL = 1e-5


@pytest.fixture(scope="module")
def ctx(G):
    c = G.Context()
    yield c
    c.close()


def test_config3_full_size_against_oracle_digest(ctx, G):
    import torch
    dig = json.load(open(os.path.join(GOLDEN, "config3_digest.json")))
    B = dig["frames"]

    def gen(b0):
        return G.synth.make_frames(64, seed=dig["seed"], snr_db=dig["snr_db"], first=b0, return_meta=True)

    with ThreadPoolExecutor(8) as ex:
        parts = list(ex.map(gen, range(0, B, 64)))
    frames = np.concatenate([p[0] for p in parts])
    meta = [m for p in parts for m in p[1]]
    cent = np.zeros(B, G.native.CAND_DTYPE)
    cent["freq"] = np.array([np.float32(m["f_off"]) for m in meta], np.float32)
    cent["shift"] = 368
    hy = G.sweep.sweep_grid(cent)
    assert hy.size == dig["hypotheses"]
    fr_t = torch.from_numpy(frames).cuda()

    # flat form: uwspr_sync_sweep, the oracle's hypothesis order
    sync, sym = ctx.sync_sweep(fr_t, hy, soft=True)
    assert hashlib.sha256(sym.tobytes()).hexdigest() == dig["symbols_sha256"]
    np.testing.assert_allclose(sync[::97], np.array(dig["sync_every_97th"], np.float32), rtol=RTOL, atol=0)
    assert hashlib.sha256(sync.tobytes()).hexdigest() == dig["sync_sha256"]   # in practice bit-exact

    # grid form: uwspr_sync_grid, [f][drift][lag] order -> reorder to the flat (f, lag, drift)
    df = np.array(G.sweep.DF_STEPS, np.float32) * np.float32(0.25)
    dd = np.array(G.sweep.DRIFTS, np.float32)
    dl = np.array(G.sweep.LAGS, np.int32)
    gsync, gsym = ctx.sync_grid(fr_t, cent, df, dd, dl, soft=True)
    gsync = gsync.transpose(0, 1, 3, 2).reshape(-1)
    gsym = gsym.transpose(0, 1, 3, 2, 4).reshape(-1, 162)
    assert gsync.tobytes() == sync.tobytes()
    assert hashlib.sha256(np.ascontiguousarray(gsym).tobytes()).hexdigest() == dig["symbols_sha256"]


def test_config4_round_robin_65536_frames(ctx, G):
    """configs[3] at full size on one card: every virtual rank r of 8 runs its
    round-robin shard frames[r::8]; the gathered slabs, restored to global order,
    equal the slabs of the same frames run in natural order."""
    import torch
    from gr_uwspr_amd import dist as D
    N = G.native
    total, world, chunk = 65536, 8, 8192
    dev = torch.device("cuda", 0)
    frames = G.synth.make_frames_torch(total, dev, seed=4242, snr_db=-20.0)
    cands = torch.empty(chunk * ctx.maxfreqs * 48, dtype=torch.uint8, device=dev)
    npk = torch.empty(chunk, dtype=torch.int32, device=dev)
    out = torch.empty(chunk * N.DEMOD_DTYPE.itemsize, dtype=torch.uint8, device=dev)

    def run(fr):
        slab = torch.empty((fr.shape[0], D.SLAB_BYTES), dtype=torch.uint8, device=dev)
        ctx.pipeline_batch_into(fr, cands, npk, out, max_per_frame=1)
        ctx.pack_slabs_into(fr.shape[0], D.SLAB_K, slab)
        ctx.synchronize()
        return slab

    natural = torch.cat([run(frames[s:s + chunk]) for s in range(0, total, chunk)])
    shards = []
    for r in range(world):
        idx = torch.from_numpy(D.shard_indices(total, r, world)).to(dev)
        assert idx.numel() == D.local_count(total, r, world) == chunk
        shards.append(run(frames.index_select(0, idx).contiguous()))
    restored = D.restore_order(torch.stack(shards), total)
    assert torch.equal(restored, natural)
    # the batch is not degenerate: nearly every -20 dB frame has a candidate that passes the gates
    nat = natural.cpu().numpy()
    have = np.frombuffer(nat[:, :4].tobytes(), np.int32)
    assert (have > 0).mean() > 0.99
    sync1 = np.frombuffer(nat[:, 16 + D.SLAB_K * 48 + 8:16 + D.SLAB_K * 48 + 12].tobytes(), np.float32)
    assert (sync1 > 0.12).mean() > 0.95
