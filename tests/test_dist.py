"""N>1 path on CPU: world_size-2 gloo processes exercise the round-robin frame
sharding, the fixed-size slab packing and the gather-to-rank-0 + order restore
that bench.py uses over RCCL on GPUs (gr-uwspr_amd/dist.py).  No data-path
collective exists: frames are independent."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import ROOT


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, total, q):
    sys.path.insert(0, ROOT)
    import gr_uwspr_amd as G
    from gr_uwspr_amd import dist as D
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    N = G.native
    mine = D.shard_indices(total, rank, world)
    bl = (total + world - 1) // world             # equal-sized shards (padded)
    maxfreqs, per = 4, 1
    cands = np.zeros((bl, maxfreqs), N.CAND_DTYPE)
    npk = np.zeros(bl, np.int32)
    demod = np.zeros((bl, per), N.DEMOD_DTYPE)
    for li, b in enumerate(mine):                 # fabricate results that encode the global index
        npk[li] = 1 + b % 3
        cands[li, :npk[li]]["freq"] = b + 0.25
        cands[li, :npk[li]]["shift"] = 128 * (b % 26)
        demod[li, 0]["f1"] = b + 0.5
        demod[li, 0]["shift1"] = 1000 + b
    slab = D.pack_slabs(torch.from_numpy(np.frombuffer(cands.tobytes(), np.uint8).copy()),
                        torch.from_numpy(npk), torch.from_numpy(np.frombuffer(demod.tobytes(), np.uint8).copy()),
                        maxfreqs, per, N.DEMOD_DTYPE.itemsize)
    g = D.gather_slabs(slab, dst=0)
    if rank == 0:
        allf = D.restore_order(g, total).numpy()
        ok = allf.shape == (total, D.SLAB_BYTES)
        for b in range(total):
            n, cs, f1, sh, dr, sy = D.unpack_slab(allf[b], N.CAND_DTYPE)
            ok &= n == 1 + b % 3 and len(cs) == n and float(cs[0]["freq"]) == b + 0.25
            ok &= int(cs[0]["shift"]) == 128 * (b % 26) and f1 == b + 0.5 and sh == 1000 + b
        q.put(bool(ok))
    else:
        assert g is None
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("total", [10, 7])
def test_gloo_world2_shard_gather_restore(total):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, total, q)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    assert q.get(timeout=5) is True


def test_shard_indices_cover_everything_once():
    from gr_uwspr_amd import dist as D
    for total in (1, 8, 65536 // 64 + 3):
        for world in (1, 2, 4, 8):
            allb = np.concatenate([D.shard_indices(total, r, world) for r in range(world)])
            assert sorted(allb.tolist()) == list(range(total))
            assert sum(D.local_count(total, r, world) for r in range(world)) == total


def test_gather_is_identity_without_a_process_group():
    from gr_uwspr_amd import dist as D
    x = torch.arange(12, dtype=torch.uint8).reshape(3, 4)
    assert torch.equal(D.gather_slabs(x)[0], x)


@pytest.mark.gpu
def test_c_abi_gather_equals_the_torch_gather_on_one_rank(G, monkeypatch):
    """uwspr_dist_*: the C-ABI gather (RCCL point-to-point transfers to the root) with one rank is the
    identity, byte-equal to dist.gather_slabs; with option dist_force_comm = 1 a real one-rank RCCL
    communicator is created and destroyed in this process (dlopen + the eight entry points), next to
    whatever collective library PyTorch brought."""
    import torch
    from gr_uwspr_amd import dist as D
    frames = G.synth.make_frames(6, seed=77, snr_db=-18.0)
    c = G.Context()
    try:
        dev = torch.from_numpy(frames).cuda()
        cd = torch.empty(6 * c.maxfreqs * 48, dtype=torch.uint8, device="cuda")
        nd = torch.empty(6, dtype=torch.int32, device="cuda")
        od = torch.empty(6 * G.native.DEMOD_DTYPE.itemsize, dtype=torch.uint8, device="cuda")
        c.pipeline_batch_into(dev, cd, nd, od, max_per_frame=1)
        slab = torch.zeros((6, D.SLAB_BYTES), dtype=torch.uint8, device="cuda")
        c.pack_slabs_into(6, D.SLAB_K, slab)
        c.synchronize()
        want = D.gather_slabs(slab)                         # [1, 6, 416]
        for force in ("0", "1"):
            c.set_option("dist_force_comm", int(force))
            uid = G.Context.dist_unique_id()
            assert len(uid) == 128 and any(uid)
            c.dist_init(0, 1, uid)
            got = torch.zeros_like(slab)
            c.dist_gather(slab, got, root=0)
            c.synchronize()
            assert torch.equal(got.view(1, 6, D.SLAB_BYTES), want)
            c.dist_finalize()
        with pytest.raises(G.UwsprError):
            c.dist_gather(slab, slab, root=0)                # not initialised
    finally:
        c.close()


def _bench_line(extra, timeout=420):
    """One `bench.py` run as the driver starts it (a child process; rank 0's JSON line on stdout)."""
    import json
    import subprocess
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", MASTER_ADDR="127.0.0.1")
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT"):
        env.pop(k, None)
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "3", "--warmup", "1", "--repeats", "1",
           "--total-frames", "50", "--sched", "staged", "--streams", "2", "--no-cpu", "--no-sweep", "--no-lazy",
           "--no-host-legs"] + extra
    r = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=timeout, cwd=ROOT)
    assert r.returncode == 0, r.stderr.decode()[-2000:]
    lines = [ln for ln in r.stdout.decode().splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout.decode()[-2000:]
    return json.loads(lines[0])


@pytest.mark.gpu
def test_bench_two_rank_rehearsal_gathers_the_one_rank_slabs():
    """The first thing the driver's multi-GPU run executes and nothing else did: `bench.py --gpus 2` spawning
    its ranks (here both on the one device: --rehearsal, gloo), round-robin shards of 50 frames per step (25 + 25;
    with three ranks the shards would be padded), the end-of-region slab gather and the order restore.  The
    frames of a small --total-frames run are a function of their global index, so rank 0's gathered slabs in
    global frame order must be byte for byte the one-rank run's."""
    import torch
    if torch.cuda.device_count() != 1:
        pytest.skip("the rehearsal shares ONE device between the ranks")
    one = _bench_line(["--gpus", "1", "--configs4-seeds", "4"])
    two = _bench_line(["--gpus", "2", "--rehearsal", "--configs4-seeds", "4"])
    assert one["n_gpus"] == 1 and two["n_gpus"] == 2
    assert one["config"]["gather"] == "none (one rank)"
    assert two["config"]["gather"] == "torch.distributed (gloo)"
    assert two["config"]["backend"] == "gloo"
    assert two["config"]["rehearsal_ranks_share_devices"] is True
    assert one["config"]["rehearsal_ranks_share_devices"] is False
    assert two["config"]["parallelism"] == "dp2" and two["scaling"] == "strong"
    assert two["config"]["frames_per_gpu"] == 25 and one["config"]["frames_per_gpu"] == 50
    sha = one["config"]["gathered_slabs_sha256"]
    assert sha and len(sha) == 64
    assert two["config"]["gathered_slabs_sha256"] == sha
    # BASELINE configs[4] with N > 1: the noisy copies of every SNR sharded round-robin, each rank on its half of the
    # host's CPUs, counts added up -- the same decodes as the one-rank sweep over the same copies
    c1, c2 = one["configs4_n1"], two["configs4"]
    assert c2["ranks"] == 2 and c2["ranks_on_this_host"] == 2
    assert c2["host_threads_per_rank"] == max(1, c1["host_threads"] // 2)
    assert [(r["snr_db"], r["frames"], r["decoded"], r["other_decodes"]) for r in c2["rows"]] == \
           [(r["snr_db"], r["frames"], r["decoded"], r["other_decodes"]) for r in c1["rows"]]
    assert all(r["frames"] == 4 and r["gpu_equals_lazy"] for r in c2["rows"]) and c2["rows"][0]["decoded"] == 4
    # three ranks: 50 frames = 17 + 17 + 16, the last shard padded to 17 rows that the order restore drops
    three = _bench_line(["--gpus", "3", "--rehearsal"])
    assert three["n_gpus"] == 3 and three["config"]["frames_per_gpu"] == 17
    assert three["config"]["gathered_slabs_sha256"] == sha
    # and what the timed lanes left in their buffers is the oracle's (bench.py: parity_spot_check), on the staged form
    for line in (one, two, three):
        sc = line["parity_spot_check"]
        assert sc["equal"] is True and sc["frames"] == 8 and sc["form"] == "staged" and sc["mismatches"] == [], sc
        assert line["chosen_streams"] == 2 and line["chosen_sched"] == "staged"
    # without --rehearsal more ranks than devices is an error, not a silent time-slice
    import subprocess
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0",
                        "--no-cpu", "--no-sweep", "--no-lazy", "--no-host-legs"], env=env, stdout=subprocess.PIPE,
                       stderr=subprocess.PIPE, timeout=300, cwd=ROOT)
    assert r.returncode == 2 and b"--rehearsal" in r.stderr


# ---- uwspr_dist_gather with world > 1 on ONE GPU: a test double of librccl (tests/host/fake_rccl.cc) ----------------
FAKE_DIR = os.path.join(ROOT, "tests", "host", "_fake_rccl")


@pytest.fixture(scope="module")
def fake_rccl():
    """librccl.so.1 (the eight entry points csrc/dist.hip resolves, over UNIX sockets + hipMemcpy) and the C-ABI worker,
    built with hipcc where the test runs.  The worker has no PyTorch in its process, so the library's
    dlopen("librccl.so.1") finds the double first on LD_LIBRARY_PATH."""
    import subprocess
    import gr_uwspr_amd as G
    G.native.build()
    os.makedirs(FAKE_DIR, exist_ok=True)
    hipcc = "/opt/rocm/bin/hipcc"
    lib = os.path.join(FAKE_DIR, "librccl.so.1")
    exe = os.path.join(FAKE_DIR, "dist_worker")
    hsrc = os.path.join(ROOT, "tests", "host")
    if not os.path.exists(lib) or os.path.getmtime(lib) < os.path.getmtime(os.path.join(hsrc, "fake_rccl.cc")):
        subprocess.run([hipcc, "-O1", "-shared", "-fPIC", "-o", lib, os.path.join(hsrc, "fake_rccl.cc")], check=True)
    if not os.path.exists(exe) or os.path.getmtime(exe) < max(os.path.getmtime(os.path.join(hsrc, "dist_worker.cc")),
                                                              os.path.getmtime(G.native.LIBPATH)):
        subprocess.run([hipcc, "-O1", "-I" + os.path.join(ROOT, "include"), os.path.join(hsrc, "dist_worker.cc"), "-o", exe,
                        "-L" + G.native.LIBDIR, "-luwspr_hip", "-Wl,-rpath," + G.native.LIBDIR], check=True)
    return exe


def _run_world(exe, world, root, rows, mode, tmp_path, timeout_s=60):
    import subprocess
    import time
    tag = "w%d_r%d_%s" % (world, root, mode)
    import gr_uwspr_amd as G
    env = dict(os.environ, LD_LIBRARY_PATH=os.pathsep.join([FAKE_DIR, G.native.LIBDIR, os.environ.get("LD_LIBRARY_PATH", "")]),
               FAKE_RCCL_DIR=str(tmp_path), FAKE_RCCL_LOG=str(tmp_path / ("ops_" + tag)), FAKE_RCCL_TIMEOUT_S=str(timeout_s))
    env.pop("UWSPR_OPTIONS", None)
    uidfile = str(tmp_path / ("uid_" + tag))
    t0 = time.time()
    procs = [subprocess.Popen([exe, str(r), str(world), str(root), str(rows), uidfile, mode], env=env,
                              stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True) for r in range(world)]
    outs = []
    for p in procs:
        o, e = p.communicate(timeout=240)
        outs.append((p.returncode, o, e))
    ops = []
    for r in range(world):
        path = "%s.%d" % (env["FAKE_RCCL_LOG"], r)
        ops.append([ln.split() for ln in open(path).read().splitlines()] if os.path.exists(path) else [])
    return outs, ops, time.time() - t0


@pytest.mark.gpu
@pytest.mark.parametrize("world,root,rows", [(2, 0, 25), (2, 1, 25), (3, 0, 17), (3, 2, 17), (4, 1, 64)])
def test_dist_gather_between_processes(fake_rccl, tmp_path, world, root, rows):
    """uwspr_dist_unique_id / init / gather / finalize between `world` processes sharing the one GPU (rows = a shard of
    configs[3] in small: 50 frames over 2 ranks, over 3 ranks padded to 17 rows): the root -- also a root that is not rank 0
    -- holds rank p's bytes at recv + p * bytes, its own shard included (the self copy); every other rank sends once to the
    root and receives nothing; the root posts world - 1 receives and sends nothing; the communicator serves a second
    gather; a root out of range and a root without a receive buffer are argument errors."""
    outs, ops, _ = _run_world(fake_rccl, world, root, rows, "ok", tmp_path)
    for r, (rc, o, e) in enumerate(outs):
        assert rc == 0 and ("GATHER_OK rank %d of %d root %d rows %d" % (r, world, root, rows)) in o, (r, rc, o, e[-800:])
    nbytes = str(rows * 416)
    for r in range(world):
        if r == root:
            assert sorted(ops[r]) == sorted([["recv", str(p), nbytes] for p in range(world) if p != root] * 2), ops[r]
        else:
            assert ops[r] == [["send", str(root), nbytes]] * 2, ops[r]


@pytest.mark.gpu
def test_dist_gather_reports_a_missing_peer_as_a_status_code(fake_rccl, tmp_path):
    """A peer that leaves after uwspr_dist_init: the root's gather comes back with UWSPR_ERR_HIP and the collective
    library's message within the transport's time-out -- a status code, never exit() or an exception across the C ABI.
    (Over real RCCL a vanished peer makes the receive WAIT; bench.py bounds that with its --comm-timeout watchdog.)"""
    outs, ops, dt = _run_world(fake_rccl, 3, 0, 17, "exit_early", tmp_path, timeout_s=3)
    rc, o, e = outs[0]
    assert rc == 0 and "GATHER_STATUS -4 RCCL gather: unhandled system error" in o, (rc, o, e[-800:])
    for r in (1, 2):
        assert outs[r][0] == 0 and "EXIT_EARLY" in outs[r][1]
        assert ops[r] == []
    assert dt < 60.0
