#!/usr/bin/env python3
"""bench.py -- frames/s of the coarse (FDR) + fine (sync_and_demodulate) path.

    python bench.py --gpus N --steps K --warmup W [--total-frames T] [--repeats R]

One step = one pass of the whole hot path (K1 spectrogram, K2 spectrum/peaks,
K3 coarse search + selection over ALL candidates, then the S0..S5 refinement
schedule incl. the 17 soft-symbol vectors for the top candidate of every frame,
and the packing of the per-frame candidate slabs) over one batch of synthetic
frames that is already resident in HBM.  The slabs of all K steps go to rank 0 in
ONE gather at the end of the timed region (inside it).

Workload = BASELINE.json configs[1]: 256 synthetic 375 Hz / 45000-sample frames
per GPU at -20 dB, flowgraph-default FDR grid, single candidate per frame.  The
timed steps rotate over FIVE distinct batches (460 MB, more than the 256 MiB
Infinity Cache), so every step ingests frames that are not cache-resident.
  weak scaling (default)   every rank runs its own 256 frames per step;
  --total-frames T         strong scaling (BASELINE configs[3]: T = 65536): the T frames
                           of a step are sharded round-robin, T / N per rank.
Frames shard with no data-path collective; the one exchange is the final gather.

`--gpus N` run bare (no WORLD_SIZE in the environment) starts the N ranks itself, one
process per GPU, BEFORE anything touches the GPU in the parent.  More ranks than visible
devices is an error (exit 2) unless --rehearsal is given: the ranks then share the devices
and talk gloo -- a functional check, flagged in the JSON, not a measurement.  Under
torch.distributed.run the environment's ranks are used.  With RCCL the slab gather goes
through the library's own communicator (uwspr_dist_*): unique id broadcast and
ncclCommInitRank on the main thread, a watchdog that ends the rank with exit code 3 when
the communicator or its first gather hangs (--comm-timeout), a clean failure falls back to
torch.distributed's gather; the line carries devices, device_name, rccl_version, the
gather that was used and whether it equalled torch's gather byte for byte.

The timed region (K steps between barrier + synchronize) is repeated R times
(default 5); `value` is the MEDIAN repeat, all repeats are listed.

Prints ONE JSON line (rank 0).  Extra objects:
  roofline      the dominant kernel, the tone-correlation schedule (k6_sched, or the K4 family
                of the staged form).  It is FP32-VALU bound (no FMA: the reference's
                two-rounding sums), so `bound` = "valu_fp32_nofma": achieved = binary32
                operations the launches' hypotheses need / kernel time, peak = 78.6 T op/s
                (256 CU x 4 SIMD x 32 lanes x 2.4 GHz), frac = achieved / peak <= 1.
                Secondary: the north-star "algorithmic bytes" rate (331950 B per hypothesis
                as if every hypothesis streamed its windows from HBM) under its own names,
                PMC-measured fabric traffic per launch, VALU issue utilisation from the
                committed PMC pass.
  cpu_baseline  the CPU restatement (oracle/, kind "port") on a bounded sample.
  fast_search   `value`'s timed region with option fast_search = 1 (FMA + shuffle-tree stages; never `value`).
  end_to_end_decoded  uwspr_pipe_*: decoded frames/s on the host, frames in HBM (its `fast_search`: the same leg with the
                option set on every lane through uwspr_pipe_set_option).
  configs3_n1   BASELINE configs[3] (65 536 frames) on this one GPU: the strong-scaling reference point.
  kernels       HIP-event time per kernel family per step (single stream).
  lazy_s5       the same step with uwspr_set_tries(1): only the first jiggered shift.
  sweep         (N=1) BASELINE configs[2]: 1024 frames x 200 (freq,lag,drift) hypotheses.
  configs4_n1   (N=1) BASELINE configs[4]: the reference's recording + AWGN, -20 .. -30 dB, decoded fraction and frames/s
                end to end (K0 front-end + search + host Fano), GPU beside the CPU path (tools/snr_sweep.py).
  host_pointer_legs  the batch handed over as host buffers (pageable / page-locked): 10 calls each, min / median / max.
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
for _p in (ROOT, os.path.join(ROOT, "oracle")):
    if _p not in sys.path:
        sys.path.insert(0, _p)

HYP_BYTES = 162 * 256 * 8 + 162 + 12      # 331950 B per fine hypothesis
OPS_MAC = 162 * 4 * 256 * 8               # binary32 ops of one hypothesis' correlation (cc:206-207)
OPS_PHASOR = 162 * 4 * 256 * 6            # ... of its per-symbol phasor recurrences (cc:193-195)
HBM_PEAK_GBS = 8000.0
FP32_NOFMA_PEAK_TOPS = 78.6               # 256 CU x 4 SIMD x 32 lanes x 2.4 GHz, one op/lane/clk
NBATCH = 5                                # distinct batches the steps rotate over


def cpu_model():
    try:
        for ln in open("/proc/cpuinfo"):
            if ln.startswith("model name"):
                return ln.split(":", 1)[1].strip()
    except OSError:
        pass
    return None


def cpu_baseline(frames_np, budget_s=12.0):
    """Oracle (CPU restatement) on a bounded sample, timed per stage as SURVEY 8(d) asks: spectrogram (+ spectrum
    statistics and peak pick), coarse search over ALL candidates, fine schedule S0..S5 with its 17 soft-symbol vectors
    for the top candidate.  One frame per worker thread (the C code is re-entrant and ctypes releases the GIL)."""
    from concurrent.futures import ThreadPoolExecutor
    import oracle_py as O
    O.lib()
    O.pr3()
    import gr_uwspr_amd as _G
    ncores = max(1, min(16, _G.host_threads()))
    fdrs = [O.FDR() for _ in range(ncores)]

    def one(args):
        w, b = args
        f = fdrs[w]
        t0 = time.perf_counter()
        ps = f.spectrogram(frames_np[b])
        cands = f.peaks(f.stats(ps)[2])
        t1 = time.perf_counter()
        found = [f.search(ps, c)[0] for c in cands]
        t2 = time.perf_counter()
        if found:
            O.demod_candidate(found[0], 1500, frames_np[b])
        t3 = time.perf_counter()
        return t1 - t0, t2 - t1, t3 - t2

    t0 = time.time()
    one((0, 0))
    per = max(time.time() - t0, 1e-3)
    nb = frames_np.shape[0]
    n = int(max(ncores, budget_s * ncores / per))      # ~budget_s seconds of wall time
    t0 = time.time()
    with ThreadPoolExecutor(ncores) as ex:
        parts = list(ex.map(one, [(i % ncores, i % nb) for i in range(n)]))
    dt = time.time() - t0
    st = np.array(parts).sum(axis=0)
    return {"value": n / dt, "unit": "frames/s", "cores": ncores, "kind": "port", "cpu_model": cpu_model(),
            "frames_per_s_per_core": n / float(st.sum()),
            "stage_core_seconds": {"spectrogram_and_peaks": float(st[0]), "coarse_search": float(st[1]),
                                   "fine_schedule": float(st[2])},
            "stage_ms_per_frame_one_core": {"spectrogram_and_peaks": 1e3 * float(st[0]) / n,
                                            "coarse_search": 1e3 * float(st[1]) / n,
                                            "fine_schedule": 1e3 * float(st[2]) / n},
            "sample": "%d frame-passes over %d of the benchmark's frames (oracle spectrogram + peaks, coarse search over "
                      "all candidates, S0..S5 schedule with 17 soft-symbol vectors for the top candidate), %d threads, "
                      "%.1f s wall = %.0f core-seconds" % (n, nb, ncores, dt, float(st.sum()))}


def parity_spot_check(snap, D, N, form):
    """Part 2 of the spot check: the oracle (the CHECKER, outside every timed region) on the frames the timed lanes read in
    their last steps, against what those lanes left in their buffers: every candidate of the frame (FDR_impl.cc:293-409,
    PDU fields, binary32 by bytes), the top candidate's record (sync_and_demodulate_impl.cc:403-482: f1 / shift1 / drift1 /
    sync1 bytes, all 17 soft-symbol vectors, jiggered shifts / sync / rms) and the slab row that step packed."""
    import oracle_py as O
    O.lib()
    O.pr3()
    fdr = O.FDR()
    bad = []
    for s in snap:
        where = "lane %d step %d frame %d" % (s["lane"], s["step"], s["frame"])
        exp = fdr.transform(s["iq"])
        if len(exp) != s["npk"]:
            bad.append("%s: npk %d != %d" % (where, s["npk"], len(exp)))
            continue
        for j, e in enumerate(exp):
            for k in O.cand_diff(s["cands"][j], e):
                bad.append("%s: candidate %d %s" % (where, j, k))
        npk_s, cands_s, f1, shift1, drift1, sync1 = D.unpack_slab(s["slab"], N.CAND_DTYPE)
        if npk_s != s["npk"] or cands_s.tobytes() != s["cands"][:min(s["npk"], D.SLAB_K)].tobytes():
            bad.append("%s: slab candidates" % where)
        if len(exp):
            d = O.demod_candidate(exp[0], 1500, s["iq"])
            for k in O.record_diff(s["out"][0], d):
                bad.append("%s: record %s" % (where, k))
            got = np.array([f1, drift1, sync1], np.float32).tobytes() + np.int32(shift1).tobytes()
            want = np.array([d["f1"], d["drift1"], d["sync1"]], np.float32).tobytes() + np.int32(d["shift1"]).tobytes()
            if got != want:
                bad.append("%s: slab record" % where)
    return {"frames": len(snap), "equal": not bad, "mismatches": bad[:10], "form": form,
            "lanes": sorted({s["lane"] for s in snap}),
            "what": "oracle (CPU restatement) vs the buffers the TIMED lanes wrote in their last steps of the last timed "
                    "region: all candidates (PDU fields, binary32 by bytes), the top candidate's f1 / shift1 / drift1 / "
                    "sync1 bytes + 17 soft-symbol vectors + jiggered shifts, and the packed slab row; outside the timed "
                    "regions"}


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--repeats", type=int, default=5, help="repeats of the timed K-step region")
    ap.add_argument("--frames", type=int, default=256, help="frames per GPU per step (weak scaling)")
    ap.add_argument("--total-frames", type=int, default=0,
                    help="strong scaling: frames per step over ALL ranks (configs[3]: 65536)")
    ap.add_argument("--snr", type=float, default=-20.0)
    ap.add_argument("--no-sweep", action="store_true")
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--no-spot-check", action="store_true", help="skip parity_spot_check (the oracle on 8 frames of the timed lanes' last batches)")
    ap.add_argument("--no-lazy", action="store_true", help="skip the lazy-S5 leg (profiling runs: only full-work launches)")
    ap.add_argument("--no-host-legs", action="store_true", help="skip the host-pointer and stream-ingest legs (profiling runs)")
    ap.add_argument("--sweep-frames", type=int, default=1024)
    ap.add_argument("--sched", choices=("auto", "fused", "staged"), default="auto",
                    help="schedule form: k6_sched (one workgroup per candidate), the staged K4/K5 launches, "
                         "or whichever measures faster in a short trial (recorded in config.sched)")
    ap.add_argument("--gather", choices=("auto", "abi", "torch"), default="auto",
                    help="final slab gather: the library's own RCCL gather (uwspr_dist_*), torch.distributed, or "
                         "auto = the former when its communicator comes up (checked once against the latter)")
    ap.add_argument("--rehearsal", action="store_true",
                    help="allow more ranks than devices (ranks time-slice the GPUs over gloo: functional check, not a "
                         "measurement); without it such a launch is an error")
    ap.add_argument("--comm-timeout", type=float, default=120.0,
                    help="seconds the library's RCCL communicator + its first gather may take before the rank exits 3")
    ap.add_argument("--trial-steps", type=int, default=100,
                    help="steps per region of the (schedule form, streams) trial: max(this, --steps), whatever --steps is")
    ap.add_argument("--configs4-seeds", type=int, default=-1,
                    help="noisy copies per SNR of the configs[4] leg (recording + AWGN, end to end incl. host Fano), sharded "
                         "round-robin over the ranks; 0 skips it; default: 64 unless --no-host-legs / --total-frames "
                         "(and, at one rank, --no-cpu) is given")
    ap.add_argument("--streams", type=int, default=0,
                    help="HIP streams (each with its own context and scratch) the steps rotate over; 0 = per --sched trial")
    return ap.parse_args()


def rccl_version(torch):
    try:
        v = torch.cuda.nccl.version()
        return ".".join(str(x) for x in v) if isinstance(v, (tuple, list)) else str(v)
    except Exception:                                    # noqa: BLE001 -- informational only
        return None


def spawn_ranks(args):
    """`--gpus N` run bare: one child process per rank, started before the parent touches the
    GPU.  Rank 0's stdout (the JSON line) is passed through."""
    # (the parent makes no GPU-runtime call at all: every rank counts the devices it sees itself and
    # falls back to the gloo rehearsal when there are fewer devices than ranks)
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    procs = []
    for r in range(args.gpus):
        env = os.environ.copy()
        env.update(RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(args.gpus), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                      stdout=subprocess.PIPE if r == 0 else subprocess.DEVNULL))
    out, _ = procs[0].communicate()
    rc = procs[0].returncode
    for p in procs[1:]:
        rc = p.wait() or rc
    sys.stdout.write(out.decode())
    sys.stdout.flush()
    sys.exit(rc)


def main():
    args = parse_args()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        spawn_ranks(args)

    import torch
    import torch.distributed as dist
    import gr_uwspr_amd as G
    from gr_uwspr_amd import dist as D
    N = G.native

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    assert torch.cuda.is_available(), "bench.py needs a GPU (there is no CPU fallback)"
    ndev = torch.cuda.device_count()
    rehearsal = ndev < world                  # several ranks on one GPU (gloo)
    if rehearsal and not args.rehearsal:
        sys.stderr.write("bench.py: %d ranks but %d visible device(s); pass --rehearsal to time-slice them over gloo "
                         "(a functional check, not a measurement)\n" % (world, ndev))
        sys.exit(2)
    local = local % ndev
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    backend = None
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        backend = os.environ.get("UWSPR_BENCH_BACKEND", "gloo" if rehearsal else "nccl")
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)
        world = dist.get_world_size()          # the ranks the collective library actually sees

    # one process per GPU: the ranks of this node share the host's CPUs -- the Fano pool of every rank is sized by its
    # share (uwspr_host_set_ranks), not by the whole host (configs[4] is host-bound)
    local_world = int(os.environ.get("LOCAL_WORLD_SIZE", str(world)))
    if local_world > 1:
        G.host_set_ranks(local_world)

    strong = args.total_frames > 0
    B = D.local_count(args.total_frames, rank, world) if strong else args.frames
    Bmax = D.local_count(args.total_frames, 0, world) if strong else B     # equal-sized (padded) shards
    nb = NBATCH if B * 360000 * NBATCH < 40e9 else max(1, int(40e9 // (B * 360000)))
    # Small strong-scaling runs (the rehearsal test, tests/test_dist.py): frame b of a batch is a function of
    # its GLOBAL index, whatever the number of ranks, so the gathered slabs of an N-rank run can be compared
    # byte for byte with the one-rank run's (`gathered_slabs_sha256`).  At configs[3]'s size every rank makes
    # its own shard from its own seed (a global batch of 65 536 frames is 23.6 GB).
    global_frames = strong and args.total_frames <= 4096
    if global_frames:
        batches = [G.synth.make_frames_torch(args.total_frames, dev, seed=0xC0FFEE + 104729 * k, snr_db=args.snr)
                   [rank::world].contiguous() for k in range(nb)]
    else:
        batches = [G.synth.make_frames_torch(B, dev, seed=0xC0FFEE + 7919 * rank + 104729 * k, snr_db=args.snr)
                   for k in range(nb)]
    K = args.steps

    # The HIP streams are created (and used) ONCE, before any context: a stream created after other
    # streams were destroyed (every closed context destroys its own) can end up sharing a hardware
    # queue with another lane -- the same three-lane configuration then measured 12 % slower
    # (tools/stream_probe.py).  Trials and the timed run use these same streams.
    max_lanes = 3 if args.streams <= 0 else args.streams
    lane_streams = [torch.cuda.Stream(device=dev) for _ in range(max_lanes)]
    for st in lane_streams:
        with torch.cuda.stream(st):
            torch.zeros(1, device=dev)
    torch.cuda.synchronize()

    # the end-to-end pipes (their contexts, lane streams and copy streams) are created here as well, for
    # the same reason; they idle until their legs run
    host_legs = rank == 0 and world == 1 and not args.no_host_legs and args.total_frames <= 0
    pipes_e2e = {f: G.Pipe(batch_frames=B, max_per_frame=1, lanes=0, sched=f) for f in ("fused", "staged")} if host_legs else {}
    pipe_st = G.Pipe(hop=3375, batch_frames=B, max_per_frame=1, lanes=0) if host_legs else None

    def make_lanes(ns, fused, extra=None):
        lanes = []
        for k in range(ns):
            st = lane_streams[k]
            cx = G.Context(device=local, options=dict({"sched": 1 if fused else 0}, **(extra or {})))
            cx.set_stream(st.cuda_stream)
            lanes.append({"stream": st, "ctx": cx,
                          "cands": torch.empty(B * cx.maxfreqs * 48, dtype=torch.uint8, device=dev),
                          "npk": torch.empty(B, dtype=torch.int32, device=dev),
                          "out": torch.empty(B * N.DEMOD_DTYPE.itemsize, dtype=torch.uint8, device=dev)})
        return lanes

    def close_lanes(lanes):
        torch.cuda.synchronize()
        for ln in lanes:
            ln["ctx"].close()

    # slabs of every step of a timed region: ONE gather at its end (off the per-step path)
    slab_ring = torch.zeros((K, Bmax, D.SLAB_BYTES), dtype=torch.uint8, device=dev)

    separate_pack = os.environ.get("UWSPR_BENCH_SEPARATE_PACK", "0") == "1"

    def step(lanes, i):
        ln = lanes[i % len(lanes)]
        with torch.cuda.stream(ln["stream"]):
            if separate_pack:                                        # (A/B: the packing kernel as its own launch)
                ln["ctx"].pipeline_batch_into(batches[i % nb], ln["cands"], ln["npk"], ln["out"], max_per_frame=1)
                ln["ctx"].pack_slabs_into(B, D.SLAB_K, slab_ring[i % K])
                return
            ln["ctx"].pipeline_slabs(D.SLAB_K, slab_ring[i % K])     # this batch's slabs from the schedule's last kernel
            ln["ctx"].pipeline_batch_into(batches[i % nb], ln["cands"], ln["npk"], ln["out"], max_per_frame=1)

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    # ---- the gather: the library's RCCL gather (C ABI) when its communicator comes up, else torch's ----
    gather_how = "none (one rank)" if world == 1 else "torch.distributed (%s)" % backend
    abi_ctx = None
    gather_recv = None
    if world > 1 and backend == "nccl" and args.gather in ("auto", "abi"):
        import threading
        gctx = G.Context(device=local)
        state = {"ok": False, "err": None}
        # The unique id travels on the MAIN thread and every rank enters the broadcast whatever happened on rank 0
        # (zeros = rank 0 could not make one): torch's process group sees the same collectives in the same order on
        # every rank.  ncclCommInitRank and the first gather run on the main thread too, before torch's group is used
        # for anything else; a watchdog ends THIS rank with exit code 3 if they hang (the launcher then fails the
        # job: a hung communicator is not something to fall back from silently).
        uid_t = torch.zeros(128, dtype=torch.uint8, device=dev)
        if rank == 0:
            try:
                uid_t.copy_(torch.frombuffer(bytearray(G.Context.dist_unique_id()), dtype=torch.uint8))
            except Exception as e:                   # noqa: BLE001
                state["err"] = "uwspr_dist_unique_id: %r" % (e,)
        dist.broadcast(uid_t, src=0)
        uid = bytes(uid_t.cpu().numpy().tobytes())
        if any(uid):
            def _hung():
                sys.stderr.write("bench.py rank %d: RCCL communicator / first gather not up after %.0f s: exit 3\n"
                                 % (rank, args.comm_timeout))
                sys.stderr.flush()
                os._exit(3)
            wd = threading.Timer(args.comm_timeout, _hung)
            wd.daemon = True
            wd.start()
            try:
                gctx.dist_init(rank, world, uid)
                # a first, small gather under the same watchdog: every rank sends 4 KB of its rank number
                probe = torch.full((4096,), rank, dtype=torch.uint8, device=dev)
                got = torch.zeros((world, 4096), dtype=torch.uint8, device=dev) if rank == 0 else None
                torch.cuda.synchronize()
                gctx.dist_gather(probe, got, root=0)
                gctx.synchronize()
                if rank == 0:
                    want = torch.arange(world, dtype=torch.uint8, device=dev)[:, None].expand(world, 4096)
                    if not torch.equal(got, want):
                        raise RuntimeError("probe gather returned wrong bytes")
                state["ok"] = True
            except Exception as e:                       # noqa: BLE001 -- a clean failure means: use torch's gather
                state["err"] = repr(e)
            finally:
                wd.cancel()
        elif state["err"] is None:
            state["err"] = "rank 0 could not create a unique id"
        okt = torch.tensor([1 if state["ok"] else 0], dtype=torch.int32, device=dev)
        dist.all_reduce(okt, op=dist.ReduceOp.MIN)
        if int(okt.item()) == 1:
            abi_ctx = gctx
            gather_how = "uwspr_dist_gather (RCCL send/recv to the root, C ABI)"
        else:
            gather_how += "; C-ABI RCCL communicator not used: %s" % (state["err"] or "failed on another rank")

    def gather_to_root(flat):
        """flat: [n, SLAB_BYTES] uint8 on the device -> [world, n, SLAB_BYTES] on rank 0 (None elsewhere)."""
        nonlocal gather_recv
        if abi_ctx is None:
            return D.gather_slabs(flat, dst=0)
        if rank == 0 and (gather_recv is None or gather_recv.numel() != world * flat.numel()):
            gather_recv = torch.empty((world,) + tuple(flat.shape), dtype=torch.uint8, device=dev)
        abi_ctx.dist_gather(flat, gather_recv if rank == 0 else None, root=0)
        abi_ctx.synchronize()
        return gather_recv if rank == 0 else None

    def region(lanes, steps, gather=True):
        """K steps between barrier + synchronize (+ the one gather); seconds, max over ranks."""
        barrier()
        t0 = time.perf_counter()
        for i in range(steps):
            step(lanes, i)
        t_enq = time.perf_counter() - t0
        torch.cuda.synchronize()
        g = gather_to_root(slab_ring.view(K * Bmax, D.SLAB_BYTES)) if gather else None
        barrier()
        dt = time.perf_counter() - t0
        if world > 1:
            tt = torch.tensor([dt], dtype=torch.float64, device="cpu" if backend == "gloo" else dev)
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            dt = float(tt.item())
        return dt, t_enq, g

    # ---- which schedule form / how many streams: a trial whose length does NOT depend on --steps, same on every rank ----
    # (round 5's driver run trialled 20-step regions -- 7 ms -- and picked 2 streams on a box where the 100-step trial
    # gives 3 streams 5.5 % more: the trial is now >= 100 steps, two regions per pair after a warm one, and a smaller
    # stream count is taken only when it wins by more than 2 %; the table and the choice are top-level scalars of the line)
    trials = {}
    KT = max(K, args.trial_steps * 256 // max(B, 1))          # (>= 100 steps of the 256-frame batch; configs[3]'s 65 536-frame steps: K)
    if args.sched == "auto" or args.streams <= 0:
        cands = []
        for fused in ((True, False) if args.sched == "auto" else ((args.sched == "fused"),)):
            for ns in ((1, 2, 3) if args.streams <= 0 else (args.streams,)):
                cands.append((fused, ns))
        for fused, ns in cands:
            lanes = make_lanes(ns, fused)
            region(lanes, min(KT, 20), gather=False)
            trials[(fused, ns)] = min(region(lanes, KT, gather=False)[0] for _ in range(2))
            close_lanes(lanes)
        tmin = min(trials.values())
        best = max((c for c in trials if trials[c] <= 1.02 * tmin), key=lambda c: (c[1], -trials[c]))
        if world > 1:   # every rank takes rank 0's choice
            ch = torch.tensor([int(best[0]), best[1]], dtype=torch.int64, device="cpu" if backend == "gloo" else dev)
            dist.broadcast(ch, src=0)
            best = (bool(ch[0].item()), int(ch[1].item()))
        fused, ns = best
    else:
        fused, ns = args.sched == "fused", args.streams
    lanes = make_lanes(ns, fused)
    ctx = lanes[0]["ctx"]

    for i in range(args.warmup):
        step(lanes, i)
    # warm up until two consecutive K-step regions agree within 2 % (clocks, caches, the lanes' rotation over
    # the distinct batches: round 2's first timed repeat was 11 % below the others), at most 8 regions
    warm_regions = []
    for _ in range(8):
        warm_regions.append(region(lanes, K, gather=False)[0])
        if len(warm_regions) >= 2 and abs(warm_regions[-1] - warm_regions[-2]) <= 0.02 * warm_regions[-2]:
            break
    # ---- the timed region, R times ----
    reps = []
    for _ in range(max(1, args.repeats)):
        dt, t_enq, gathered = region(lanes, K)
        reps.append((dt, t_enq))
    # ---- parity spot check, part 1 (outside the timed regions): freeze what the TIMED lanes wrote in their last steps --
    # the candidate / record buffers of every lane, the slab row of that step and the frames it read -- on the host; the
    # oracle runs on them at the end of the run (part 2), so the line says whether the bytes `value` was measured on are
    # the reference's (round 5's k4_jig bug sat in exactly this path, unseen by the forms-against-forms tests)
    spot_snap = []
    if rank == 0 and not args.no_spot_check:
        SPOT = 8
        live_lanes = [l for l in range(len(lanes)) if l < K]
        for q in range(SPOT):
            l = live_lanes[q % len(live_lanes)]
            i_last = max(i for i in range(K) if i % len(lanes) == l)       # that lane's last step of the region
            b = (q * 37 + 5) % B
            ln = lanes[l]
            spot_snap.append({"lane": l, "step": i_last, "frame": b,
                              "iq": batches[i_last % nb][b].cpu().numpy().copy(),
                              "npk": int(ln["npk"][b].item()),
                              "cands": np.frombuffer(ln["cands"].view(B, -1)[b].cpu().numpy().tobytes(), N.CAND_DTYPE).copy(),
                              "out": np.frombuffer(ln["out"].view(B, -1)[b].cpu().numpy().tobytes(), N.DEMOD_DTYPE).copy(),
                              "slab": slab_ring[i_last % K, b].cpu().numpy().copy()})
    gathered_sha = None
    if rank == 0 and global_frames and gathered is not None:
        # [world, K * Bmax, SLAB] -> per step, global frame order (b = local * G + rank), padding rows dropped
        import hashlib
        gs = gathered.view(world, K, Bmax, D.SLAB_BYTES).cpu()
        hh = hashlib.sha256()
        for k in range(K):
            hh.update(D.restore_order(gs[:, k], args.total_frames).contiguous().numpy().tobytes())
        gathered_sha = hh.hexdigest()
    gather_checked = None
    if abi_ctx is not None:          # once, outside the timed regions: the two gathers give the same bytes
        ref = D.gather_slabs(slab_ring.view(K * Bmax, D.SLAB_BYTES), dst=0)
        if rank == 0:
            gather_checked = bool(torch.equal(ref.to(dev), gathered))
    order = sorted(range(len(reps)), key=lambda r: reps[r][0])
    dt, t_enq = reps[order[len(order) // 2]]
    frames_per_step = args.total_frames if strong else world * B
    rates = [frames_per_step * K / r[0] for r in reps]

    # ---- option "fast_search" (FMA + shuffle-tree stages S0..S4: NOT the reference's arithmetic; never `value`) and
    # ---- BASELINE configs[3] at N = 1 (65 536 frames through the same lanes: SCALE's strong-scaling reference point)
    fast_leg = None
    c3_leg = None
    if world == 1 and not strong and not args.no_lazy:
        nfl = min(3, max_lanes)                 # --streams 1 / 2 leaves fewer lane streams than the default three
        fl_lanes = make_lanes(nfl, False, {"fast_search": 1})
        region(fl_lanes, min(K, 10), gather=False)
        fr = sorted(region(fl_lanes, K)[0] for _ in range(3))
        close_lanes(fl_lanes)
        fast_leg = {"frames_per_s": B * K / fr[1], "ms_per_step": 1e3 * fr[1] / K, "min": B * K / fr[2], "max": B * K / fr[0],
                    "repeats": 3, "sched": "staged", "streams": nfl,
                    "what": "the timed region of `value` with uwspr_set_option(fast_search, 1): stages S0..S4 with fused "
                            "multiply-adds and wavefront shuffle-tree sums (sync metrics agree with the exact path to "
                            "~1e-6, integer results and soft symbols identical: tests/test_gpu_fast_search.py); NOT the "
                            "reference's arithmetic, so never `value`"}
        T3 = 65536
        n3 = T3 // B
        ring3 = torch.zeros((n3, B, D.SLAB_BYTES), dtype=torch.uint8, device=dev)

        def c3_region():
            barrier()
            t0 = time.perf_counter()
            for i in range(n3):
                ln = lanes[i % len(lanes)]
                with torch.cuda.stream(ln["stream"]):
                    ln["ctx"].pipeline_slabs(D.SLAB_K, ring3[i])
                    ln["ctx"].pipeline_batch_into(batches[i % nb], ln["cands"], ln["npk"], ln["out"], max_per_frame=1)
            torch.cuda.synchronize()
            D.gather_slabs(ring3.view(n3 * B, D.SLAB_BYTES), dst=0)
            barrier()
            return time.perf_counter() - t0
        c3_region()
        t3s = sorted(c3_region() for _ in range(3))
        del ring3
        c3_leg = {"total_frames": n3 * B, "frames_per_s": n3 * B / t3s[1], "ms_total": 1e3 * t3s[1],
                  "min": n3 * B / t3s[2], "max": n3 * B / t3s[0], "repeats": 3,
                  "what": "BASELINE configs[3] on ONE GPU: %d frames as %d batches of %d through the lanes of `value`, one "
                          "gather of all %d slabs at the end (what `--gpus N --total-frames 65536` shards round-robin)"
                          % (n3 * B, n3, B, n3 * B)}

    # ---- per-kernel HIP-event times: single stream, same rotating batches (untimed for `value`) ----
    ctx.prof_enable(True)
    ctx.prof_read()
    n1 = min(K, 20)
    for i in range(n1):
        step(lanes[:1], i)
    barrier()
    prof = ctx.prof_read()
    ctx.prof_enable(False)
    # the dominant kernel alone (events around it only: every recorded event is a marker packet)
    ctx.prof_enable(("tonecorr",))
    ctx.prof_read()
    for i in range(n1):
        step(lanes[:1], i)
    barrier()
    k4 = ctx.prof_read()["tonecorr"]
    ctx.prof_enable(False)

    # ---- what the batch contained (outside the timed region) ----------------
    lanes[0]["ctx"].pipeline_batch_into(batches[0], lanes[0]["cands"], lanes[0]["npk"], lanes[0]["out"], max_per_frame=1)
    torch.cuda.synchronize()
    cands = np.frombuffer(lanes[0]["cands"].cpu().numpy().tobytes(), N.CAND_DTYPE).reshape(B, -1)
    npk = lanes[0]["npk"].cpu().numpy()
    out = np.frombuffer(lanes[0]["out"].cpu().numpy().tobytes(), N.DEMOD_DTYPE).reshape(B, 1)
    top_lin = np.array([cands[b, 0]["m_type"] == 0 if npk[b] > 0 else False for b in range(B)])
    live = npk > 0
    worth = out[:, 0]["worth_a_try"] > 0
    nlive, nlin, nworth = int(live.sum()), int((live & top_lin).sum()), int(worth.sum())
    # hypotheses the reference resolves per candidate: S0 5, S1 5, S2 2 (linear), S3 5, S4 5, S5 17
    fine_hyps = 10 * nlive + 2 * nlin + 27 * nworth
    # correlations the kernels run: the stage winner repeated by S1/S3/S4 and by try 0 of S5 is carried
    # (both forms since round 3), S0's last lag is its first one symbol later
    reuse_on = ctx.get_option("reuse") != 0
    fine_corr = fine_hyps - ((nlive + 3 * nworth) if reuse_on else 0) - nlive
    # binary32 operations those correlations need: 8 per sample, tone and hypothesis; the per-symbol
    # phasor recurrences (6) only where the algorithm cannot share them: the two drift tries of S2, and since round 4
    # ONE recurrence per mirrored symbol pair of the two (k4_pair.hip: 163 pair-rows for 2 x 162 symbols; the candidates
    # of this workload come from an FDR with maxdrift = 0, so every linear one is mirrored)
    ops_step = float(fine_corr) * OPS_MAC + (163.0 / 162.0) * nlin * OPS_PHASOR

    # host tail alone: de-interleave + Fano on the persistent pool (all the cores this process may use),
    # one call per 256-record batch as the pipeline makes them
    nthr = G.host_threads()                  # affinity and cgroup CPU quota applied
    G.decode_batch(out[:8, 0], nthreads=nthr)
    _, _, okv = G.decode_batch(out[:, 0], nthreads=nthr)
    ht = []
    for _ in range(7):
        t_h = time.perf_counter()
        for _k in range(8):
            G.decode_batch(out[:, 0], nthreads=nthr)
        ht.append(8 * B / (time.perf_counter() - t_h))
    host_tail = {"records": int(B), "decoded": int(okv.sum()), "threads": nthr, "calls_per_repeat": 8,
                 "records_per_s": float(np.median(ht)), "min": min(ht), "max": max(ht)}

    # ---- lazy jiggered shifts: only try 0 (fused form) ----
    lazy = None
    if rank == 0 and not args.no_lazy:
        torch.cuda.synchronize()
        lazy_by_lanes = {}
        lz = None
        for lf in (False, True):                    # both schedule forms ...
            for nl in range(1, max_lanes + 1):      # ... with 1, 2, 3 batches in flight
                if lz is not None:
                    close_lanes(lz)
                lz = make_lanes(nl, lf)
                for ln in lz:
                    ln["ctx"].set_tries(1)
                for i in range(6):
                    step(lz, i)
                torch.cuda.synchronize()
                t2 = time.perf_counter()
                for i in range(K):
                    step(lz, i)
                torch.cuda.synchronize()
                lazy_by_lanes[(lf, nl)] = time.perf_counter() - t2
        (lbest_fused, nbest) = min(lazy_by_lanes, key=lazy_by_lanes.get)
        dl = lazy_by_lanes[(lbest_fused, nbest)]
        lz[0]["ctx"].pipeline_batch_into(batches[0], lz[0]["cands"], lz[0]["npk"], lz[0]["out"], max_per_frame=1)
        torch.cuda.synchronize()
        out1 = np.frombuffer(lz[0]["out"].cpu().numpy().tobytes(), N.DEMOD_DTYPE).reshape(B, 1)
        _, _, ok1 = G.decode_batch(out1[:, 0], nthreads=nthr)
        # the records try 0 did not decode get their other tries from uwspr_demod_resume
        need = torch.from_numpy((~ok1.astype(bool)).astype(np.uint8)).to(dev)
        lz[0]["ctx"].demod_resume(batches[0], need, lz[0]["out"], max_per_frame=1)   # first call: allocations
        torch.cuda.synchronize()
        t3 = time.perf_counter()
        lz[0]["ctx"].demod_resume(batches[0], need, lz[0]["out"], max_per_frame=1)
        torch.cuda.synchronize()
        t_res = time.perf_counter() - t3
        lazy = {"frames_per_s": B * K / dl, "ms_per_step": 1e3 * dl / K, "tries": 1, "streams": nbest,
                "sched": "fused" if lbest_fused else "staged",
                "ms_per_step_by_form_and_streams": {("%s x%d" % ("fused" if f else "staged", k)): 1e3 * v / K
                                                    for (f, k), v in lazy_by_lanes.items()},
                "decoded_by_try_0": int(ok1.sum()), "of": int(B), "resume_ms_for_the_rest": 1e3 * t_res,
                "note": "uwspr_set_tries(1), this rank: the reference stops at its first decoding try "
                        "(cc:457-490); the rest is produced on demand by uwspr_demod_resume"}
        close_lanes(lz)
    if world > 1:
        dist.barrier()

    # a host copy of one batch: for the host-pointer legs and for cpu_baseline (which --no-host-legs alone does not switch off)
    frames_cpu = batches[0].cpu().numpy() if (rank == 0 and world == 1 and not (args.no_host_legs and args.no_cpu)) else None
    # the same batch handed over as HOST buffers (what a GNU Radio block does per PDU batch):
    # H2D of the frames + D2H of every result, PCIe inclusive; never `value`
    host_rate = None
    host_pinned_rate = None
    host_legs_detail = None
    if frames_cpu is not None and not args.no_host_legs:
        def host_leg(buf, calls=10):
            ctx.pipeline_batch(buf, max_per_frame=1)          # untimed: allocations, first touch
            ts = []
            for _ in range(calls):
                t2 = time.perf_counter()
                ctx.pipeline_batch(buf, max_per_frame=1)
                ts.append(time.perf_counter() - t2)
            r = sorted(B / t for t in ts)
            return {"calls": calls, "frames_per_s": {"min": r[0], "median": float(np.median(r)), "max": r[-1]},
                    "GB_per_s_h2d_median": B * 360000 / float(np.median(ts)) / 1e9}
        pageable = host_leg(frames_cpu)
        # ... from PAGE-LOCKED host memory: torch's pinned allocator, and the library's own uwspr_host_alloc
        pinned = torch.from_numpy(frames_cpu).pin_memory()
        pinned_torch = host_leg(pinned.numpy())
        hp = G.host_alloc(frames_cpu.nbytes) if hasattr(G, "host_alloc") else None
        pinned_lib = None
        if hp is not None:
            hview = np.frombuffer(hp, dtype=np.float32).reshape(frames_cpu.shape)
            hview[:] = frames_cpu
            pinned_lib = host_leg(hview)
            del hview
            G.host_free(hp)
        host_rate = pageable["frames_per_s"]["median"]
        host_pinned_rate = pinned_torch["frames_per_s"]["median"]
        numa = {}
        try:
            import glob
            numa["gpu_numa_nodes"] = sorted({open(q).read().strip() for q in glob.glob("/sys/class/drm/card*/device/numa_node")})
            numa["cpus_allowed"] = len(os.sched_getaffinity(0))
            numa["numa_nodes_online"] = open("/sys/devices/system/node/online").read().strip()
        except OSError:
            pass
        host_legs_detail = {"pageable": pageable, "pinned_torch": pinned_torch, "pinned_uwspr_host_alloc": pinned_lib,
                            "placement": numa,
                            "what": "uwspr_pipeline_batch on a HOST buffer of %d frames (92 MB up, records back), PCIe "
                                    "inclusive; 10 timed calls each; a page-locked buffer is read by one DMA (uwspr_api.hip: "
                                    "upload), a pageable one goes through the runtime's staging" % B}
    # ---- end to end through the pipelined C-ABI driver (uwspr_pipe_*): lazy schedule || D2H of the records
    # || Fano on the persistent host pool || resume of what try 0 did not decode; results collected on the
    # host.  (a) decoded: the HBM-resident benchmark batches, every frame decodable -- "frames decoded/s";
    # (b) stream: a continuous 375 S/s stream, every sample crossing PCIe once from page-locked staging
    # buffers (uwspr_pipe_acquire / commit; the buffers are pre-filled, as a driver DMA-ing samples into
    # them would leave them: producing the samples is not part of the path), frames every 3375 samples.
    e2e = None
    stream_leg = None
    if host_legs:
        torch.cuda.synchronize()
        REP, KS = 5, 500      # (round 3's 100-step repeats of these thread-pool + three-lane legs spread by +-19 % on the driver's box)
        by_form = {}
        for form, pipe in pipes_e2e.items():
            try:
                for i in range(3 * nb):
                    pipe.submit_device(batches[i % nb], B)
                pipe.flush()
                pipe.collect()
                prates = []
                for _ in range(REP):
                    t2 = time.perf_counter()
                    for i in range(KS):
                        pipe.submit_device(batches[i % nb], B)
                        if i % 8 == 7:
                            pipe.collect()
                    pipe.flush()
                    pipe.collect()
                    prates.append(KS * B / (time.perf_counter() - t2))
                st = pipe.stats()
                frates = []
                if form == "staged":      # the same leg with the fast_search option on every lane (NOT the reference's arithmetic)
                    pipe.set_option("fast_search", 1)
                    for _ in range(3):
                        t2 = time.perf_counter()
                        for i in range(KS):
                            pipe.submit_device(batches[i % nb], B)
                            if i % 8 == 7:
                                pipe.collect()
                        pipe.flush()
                        pipe.collect()
                        frates.append(KS * B / (time.perf_counter() - t2))
                    st_fast = pipe.stats()
            finally:
                pipe.close()
            by_form[form] = {"frames_per_s": float(np.median(prates)), "min": min(prates), "max": max(prates),
                             "decoded_fraction": st["decoded"] / max(st["candidates"], 1),
                             "resumed_fraction": st["resumed"] / max(st["candidates"], 1),
                             "fano_calls_per_frame": st["fano_calls"] / max(st["frames"], 1),
                             "fano_timeouts_per_frame": st["fano_timeouts"] / max(st["frames"], 1),
                             "coordinator_s": {k: st[k] for k in ("gpu_wait_s", "fano_s", "resume_s")}}
            if frates:
                by_form[form]["fast_search"] = {"frames_per_s": float(np.median(frates)), "min": min(frates), "max": max(frates),
                                                "decoded_fraction": (st_fast["decoded"] - st["decoded"]) / max(st_fast["candidates"] - st["candidates"], 1)}
        best = max(by_form, key=lambda f: by_form[f]["frames_per_s"])
        e2e = dict(by_form[best])
        if "fast_search" not in e2e and "fast_search" in by_form.get("staged", {}):
            e2e["fast_search"] = by_form["staged"]["fast_search"]
        e2e.update({"sched": best, "by_sched": {f: v["frames_per_s"] for f, v in by_form.items()}, "repeats": REP,
                    "steps_per_repeat": KS, "frames_per_step": B, "lanes": "library default: 3 streams + 6 spare lanes (opened only under long host tails)", "host_threads": max(1, G.host_threads() - 2),
                    "what": "uwspr_pipe_submit_device: frames resident in HBM (the same rotating batches as `value`), "
                            "FDR + lazy S0..S5 on 3 lanes, records to the host, Fano for every frame on the persistent "
                            "host pool, resume + Fano for what try 0 did not decode, messages collected in frame order"})
        # (b) the pushed stream, twice: a QUIET stream (noise, a transmission only every 40th frame length:
        # a window without one never reaches Fano -- the ingest rate) and a BUSY one (a transmission in every
        # window: 12 of 13 windows see it outside the search range, a few of their tries pass the gates
        # (cc:470) and run Fano to its 10000-cycles-per-bit time-out on the host -- the reference's cost)
        hop = 3375
        pipe = pipe_st
        stream_leg = {}
        try:
            rng = np.random.default_rng(3)
            one = batches[0][:20].cpu().numpy()                     # 20 transmissions to sprinkle over the stream
            sig = one[:, 375:375 + 162 * 256] - 0.0
            for name, every in (("quiet", 40), ("busy", 1)):
                for k in range(4):                                 # the four staging buffers, filled once
                    buf = pipe.acquire(B * hop)
                    buf[:] = (G.synth.sigma_for_snr(args.snr) * rng.standard_normal((B * hop, 2))).astype(np.float32)
                    for t in range(0, B * hop // 45000 - 1, every):
                        s0 = t * 45000 + int(rng.integers(0, 3000))
                        buf[s0:s0 + sig.shape[1]] += sig[t % 20]
                    pipe.commit(B * hop)
                pipe.flush()
                pipe.collect()
                st0 = pipe.stats()
                prates = []
                ks = KS if name == "quiet" else 40
                for _ in range(REP if name == "quiet" else 3):
                    t2 = time.perf_counter()
                    f0 = pipe.stats()["frames"]
                    for i in range(ks):
                        pipe.acquire(B * hop)
                        pipe.commit(B * hop)
                        if i % 8 == 7:
                            pipe.collect()
                    pipe.flush()
                    pipe.collect()
                    prates.append((pipe.stats()["frames"] - f0) / (time.perf_counter() - t2))
                st = pipe.stats()
                nf = max(st["frames"] - st0["frames"], 1)
                stream_leg[name] = {"frames_per_s": float(np.median(prates)), "min": min(prates), "max": max(prates),
                                    "repeats": len(prates), "steps_per_repeat": ks,
                                    "decoded_per_frame": (st["decoded"] - st0["decoded"]) / nf,
                                    "fano_timeouts_per_frame": (st["fano_timeouts"] - st0["fano_timeouts"]) / nf}
        finally:
            pipe.close()
        stream_leg.update({"new_samples_per_step": B * hop, "bytes_uploaded_per_step": B * hop * 8,
                           "what": "uwspr_pipe_acquire/commit: PCIe-inclusive (every sample uploaded once on the copy "
                                   "stream from page-locked staging buffers, frames read in place at stride 3375, all "
                                   "records back to the host, gates + Fano on the host pool); never `value`"})
    if args.no_cpu:
        frames_cpu = None
    result = None
    pmc = None
    tpath = os.path.join(ROOT, "profiles", "k4_traffic.json")
    traffic_stale = None
    if os.path.exists(tpath):                 # PMC passes taken with rocprofv3 on this workload (tools/run_profiles.sh)
        pmc = json.load(open(tpath))
        if pmc.get("library_sources_sha256") != N.source_digest():
            # counters of ANOTHER build say nothing about this one: loud, and no figure rather than a stale one
            traffic_stale = ("profiles/k4_traffic.json was taken on other sources (%s...) than this build (%s...): "
                             "re-run tools/run_profiles.sh + tools/make_final_profile.sh"
                             % (str(pmc.get("library_sources_sha256"))[:12], N.source_digest()[:12]))
            if rank == 0:
                print("WARNING: " + traffic_stale, file=sys.stderr)
            pmc = None
    if rank == 0:
        k4_ms_step = k4["ms"] / n1
        k4_launch_ms = k4["ms"] / max(k4["launches"], 1)
        achieved_tops = ops_step / (k4_ms_step * 1e-3) / 1e12 if k4_ms_step > 0 else 0.0
        eff_gbs = fine_corr * HYP_BYTES / (k4_ms_step * 1e-3) / 1e9 if k4_ms_step > 0 else 0.0
        kern = {k: {"ms_per_step": v["ms"] / n1, "launches_per_step": v["launches"] / n1} for k, v in prof.items()}
        kern["tonecorr"] = {"ms_per_step": k4_ms_step, "launches_per_step": k4["launches"] / n1}
        pm = (pmc or {}).get("fused" if fused else "staged", {}) if pmc else {}
        result = {
            "metric": "2-min WSPR frames decoded/sec (coarse+sync)",
            "value": frames_per_step * K / dt,
            "unit": "frames/s",
            "n_gpus": world, "steps": K, "warmup": args.warmup,
            "ms_per_step": 1e3 * dt / K,
            "higher_is_better": True, "scaling": "strong" if strong else "weak", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "chosen": {"sched": "fused" if fused else "staged", "streams": ns},
            "chosen_sched": "fused" if fused else "staged", "chosen_streams": ns, "trial_steps": KT if trials else None,
            "trial_rule": "min of two %d-step regions per (form, streams); the largest stream count within 2 %% of the best" % KT,
            "single_stream": ({"frames_per_s": frames_per_step * KT / min(v for (f, s_), v in trials.items() if s_ == 1),
                               "sched": "fused" if min(((v, f) for (f, s_), v in trials.items() if s_ == 1))[1] else "staged",
                               "note": "the same K-step region on ONE HIP stream (best schedule form); `value` overlaps "
                                       "%d batches on %d streams" % (ns, ns)}
                              if any(s_ == 1 for (_f, s_) in trials) else None),
            "warmup_regions_until_stable": len(warm_regions),
            "repeats": {"n": len(reps), "frames_per_s": rates, "min": min(rates), "median": float(np.median(rates)),
                        "max": max(rates), "value_is": "median"},
            "config": {"workload": "BASELINE configs[%d]: %s synthetic .c2 frames (375 Hz, 45000 samples, SNR %.0f dB)"
                                   " per step, flowgraph-default FDR grid (hbw=10, maxdrift=0, threshold=10, all "
                                   "candidates searched), S0..S5 schedule + 17 soft-symbol vectors for the top "
                                   "candidate of each frame, slab packing every step, one slab gather to rank 0 per "
                                   "timed region" % (3 if strong else 1,
                                                     ("%d (sharded round-robin, %d per GPU)" % (args.total_frames, Bmax))
                                                     if strong else ("%d per GPU" % B), args.snr),
                       "frames_per_gpu": B, "distinct_batches": nb, "bytes_of_distinct_frames": nb * B * 360000,
                       "candidates_per_frame_mean": float(npk.mean()),
                       "fine_hypotheses_per_step": fine_hyps,
                       "fine_correlations_run_per_step": fine_corr,
                       "coarse_hypotheses_per_step": int(npk.sum()) * 130 * ctx.info.cell_hyps,
                       "top_candidate_decodes": int(okv.sum()), "parallelism": "dp%d" % world,
                       "sched": "fused (k6_sched)" if fused else "staged (k4_* + k5_fold_step)",
                       "streams_per_gpu": ns,
                       "trial_ms_per_step": {("%s x%d streams" % ("fused" if f else "staged", s)): 1e3 * v / KT
                                             for (f, s), v in trials.items()},
                       "rehearsal_ranks_share_devices": bool(rehearsal), "backend": backend, "devices": min(ndev, world),
                       "devices_visible": ndev, "device_name": torch.cuda.get_device_name(local),
                       "rccl_version": rccl_version(torch),
                       "gather": gather_how, "gather_equals_torch_gather": gather_checked,
                       "gathered_slabs_sha256": gathered_sha},
            "roofline": {"kernel": "k6_sched" if fused else "k4_* (6 launches)", "bound": "valu_fp32_nofma",
                         "achieved": achieved_tops, "peak": FP32_NOFMA_PEAK_TOPS, "unit": "Top/s",
                         "frac": achieved_tops / FP32_NOFMA_PEAK_TOPS,
                         "traffic": pm.get("bytes_per_step"), "traffic_source": pm.get("source"),
                         "traffic_stale": traffic_stale,
                         "valu_issue_utilisation_pmc": pm.get("valu_issue_utilisation"),
                         "ops_per_step_algorithmic": ops_step,
                         "kernel_ms_per_step": k4_ms_step, "avg_launch_ms": k4_launch_ms,
                         "launches_per_step": k4["launches"] / n1,
                         "accounting": "single stream, HIP events in the kernel's own dispatch packet, steps over "
                                       "%d distinct batches; ops = 8 per sample, tone and hypothesis of the "
                                       "correlations the launches run (config.fine_correlations_run_per_step) + 6 "
                                       "for the per-symbol phasors of S2's two drift tries, one recurrence per "
                                       "mirrored symbol pair (the count of rounds 2-3 was two: 2.1 %% more ops)" % nb,
                         "north_star_algorithmic": {
                             "effective_sample_rate_GBs": eff_gbs, "x_hbm_peak": eff_gbs / HBM_PEAK_GBS,
                             "note": "331950 B per correlation as if every hypothesis streamed its symbol windows "
                                     "from HBM (SURVEY 8(d)); not an HBM utilisation: the measured fabric traffic is "
                                     "`traffic`"}},
            "kernels": kern,
            "fast_search": fast_leg,
            "configs3_n1": c3_leg,
            "lazy_s5": lazy,
            "host_pointer_frames_per_s_pcie_inclusive": host_rate,
            "host_pinned_pointer_frames_per_s_pcie_inclusive": host_pinned_rate,
            "host_pointer_legs": host_legs_detail,
            "end_to_end_decoded": e2e,
            "host_stream_pcie_inclusive": stream_leg,
            "host_tail_fano": host_tail,
            "host_enqueue_ms_per_step": 1e3 * t_enq / K,
        }

    if rank == 0:
        for (f, s_), v in sorted(trials.items()):       # top-level scalars: the driver's record keeps them
            result["trial_%s_x%d_ms" % ("fused" if f else "staged", s_)] = 1e3 * v / KT

    # ---- configs[2]: the (freq, lag, drift) sweep, N=1 only -----------------
    if rank == 0 and world == 1 and not args.no_sweep and not strong:
        from gr_uwspr_amd import sweep as SW
        Bs = args.sweep_frames
        del batches
        torch.cuda.empty_cache()
        fr2 = G.synth.make_frames_torch(Bs, dev, seed=99, snr_db=args.snr)
        H = Bs * 200
        cent = np.zeros(Bs, N.CAND_DTYPE)
        cent["freq"] = 0.0
        cent["shift"] = 368
        cent_t = torch.from_numpy(np.frombuffer(cent.tobytes(), np.uint8).copy()).to(dev)
        df = np.array(SW.DF_STEPS, np.float32) * np.float32(0.25)
        dd = np.array(SW.DRIFTS, np.float32)
        dl = np.array(SW.LAGS, np.int32)
        sync_t = torch.empty(H, dtype=torch.float32, device=dev)
        sym_t = torch.empty(H * 162, dtype=torch.uint8, device=dev)
        sweep = {"workload": "BASELINE configs[2]: %d frames x 200 (freq,drift,lag) hypotheses, sync + 162 "
                             "soft symbols each" % Bs, "hypotheses": H, "north_star_bar_ms": 28.3}

        def timed(fn, ops_per_hyp, reps=3):
            fn()
            torch.cuda.synchronize()
            ctx.prof_enable(True)
            ctx.prof_read()
            t0 = time.perf_counter()
            for _ in range(reps):
                fn()
            torch.cuda.synchronize()
            dts = (time.perf_counter() - t0) / reps
            p2 = ctx.prof_read()
            ctx.prof_enable(False)
            k4ms = p2["tonecorr"]["ms"] / reps
            tops = H * ops_per_hyp / (k4ms * 1e-3) / 1e12
            return {"ms_total": 1e3 * dts, "k4_ms": k4ms, "k5_ms": p2["fold"]["ms"] / reps,
                    "k4_launches": p2["tonecorr"]["launches"] / reps, "hyps_per_s": H / dts,
                    "fp32_tops_k4": tops, "frac_of_valu_peak_k4": tops / FP32_NOFMA_PEAK_TOPS,
                    "effective_sample_rate_GBs_k4": H * HYP_BYTES / (k4ms * 1e-3) / 1e9,
                    "x_hbm_peak_k4": H * HYP_BYTES / (k4ms * 1e-3) / 1e9 / HBM_PEAK_GBS}

        # grid form (uwspr_sync_grid): shared symbol windows, phasors shared by the 8 lags of a (freq, drift)
        sweep["grid"] = timed(lambda: ctx.sync_grid(fr2, cent_t, df, dd, dl, into=(sync_t, sym_t)),
                              OPS_MAC + OPS_PHASOR / 8.0)
        grid_sync = sync_t.clone()
        # flat form (uwspr_sync_sweep) on the same 204800 hypotheses, same order
        hy = G.sweep_grid_uniform(Bs, f_c=0.0, shift_c=368)
        hy = hy.reshape(Bs, 5, 8, 5).transpose(0, 1, 3, 2).reshape(-1).copy()
        hy_t = torch.from_numpy(np.frombuffer(hy.tobytes(), np.uint8).copy()).to(dev)
        sweep["flat"] = timed(lambda: ctx.sync_sweep_into(fr2, hy_t, H, sync_t, sym_t), OPS_MAC + OPS_PHASOR)
        sweep["grid_equals_flat_bitwise"] = bool(torch.equal(grid_sync, sync_t))
        result["sweep"] = sweep

    # ---- BASELINE configs[4]: the reference's recording + AWGN at -20 .. -30 dB, end to end (host audio -> K0 = the
    # flowgraph's front-end chain -> FDR -> S0..S5 -> records to the host -> Fano + unpack), decoded fraction and frames/s.
    # N = 1: the CPU path (oracle search + the same host tail, no front-end) beside it on a bounded sample.  N > 1: the
    # noisy copies of every SNR are sharded round-robin over the ranks (copy s on rank s mod N), every rank decodes its
    # shard on ITS share of the host's CPUs (uwspr_host_set_ranks), and the counts are added up: SUM of frames / decodes,
    # MAX of the ranks' times -- no data-path exchange, three small all-reduces at the end.
    c4_seeds = args.configs4_seeds
    if c4_seeds < 0:
        c4_seeds = 64 if (not args.no_host_legs and not strong and (world > 1 or not args.no_cpu)) else 0
    if c4_seeds > 0:
        sys.path.insert(0, os.path.join(ROOT, "tools"))
        sw = None
        c4_err = None
        try:
            import snr_sweep
            sw = snr_sweep.run(seeds=c4_seeds, cpu_seeds=(8 if world == 1 and not args.no_cpu else 0), quiet=True,
                               rank=rank, world=world)
        except Exception as e:                     # noqa: BLE001 -- every rank still enters the reductions below
            c4_err = repr(e)
        nsn = 6
        cnt = np.zeros((nsn, 4), np.float64)       # frames, decoded, other decodes, lazy records resumed
        tim = np.zeros((nsn, 2), np.float64)       # eager seconds, lazy seconds
        okf = np.array([0.0 if c4_err else 1.0] + [1.0] * nsn)
        if sw is not None:
            for i, r in enumerate(sw["rows"]):
                cnt[i] = (r["frames"], r["decoded"], r["other_decodes"], r["lazy_records_resumed"])
                tim[i] = (r["gpu_s"], r["gpu_lazy_s"])
                okf[1 + i] = 1.0 if r["gpu_equals_cpu"] else 0.0
        if world > 1:
            rdev = "cpu" if backend == "gloo" else dev
            tc_, tt_, to_ = (torch.from_numpy(v).to(rdev) for v in (cnt, tim, okf))
            dist.all_reduce(tc_, op=dist.ReduceOp.SUM)
            dist.all_reduce(tt_, op=dist.ReduceOp.MAX)
            dist.all_reduce(to_, op=dist.ReduceOp.MIN)
            cnt, tim, okf = tc_.cpu().numpy(), tt_.cpu().numpy(), to_.cpu().numpy()
        if rank == 0:
            if okf[0] == 0.0 or sw is None:
                result["configs4_n1" if world == 1 else "configs4"] = {"error": c4_err or "failed on another rank"}
            elif world == 1:
                result["configs4_n1"] = {
                    "what": "BASELINE configs[4] at N = 1: examples/150613_1920.wav (tests/golden/150613_1920_int16.npz) + AWGN, "
                            "%d noisy copies per SNR handed over as HOST audio (5.76 MB each, PCIe inclusive), up to 4 candidates per "
                            "frame through the schedule, Fano + unpack on the host pool; `gpu_lazy` = uwspr_set_tries(1) + "
                            "uwspr_demod_resume (the reference's early exit); CPU = oracle FDR + schedule + the same host tail on the "
                            "first 8 frames of each SNR, threads as stated, no front-end; never `value`" % c4_seeds,
                    "native_snr_db": sw["native_snr_db"], "front_end": sw["front_end"], "host_threads": sw["host_threads"],
                    "rows": [{k: r[k] for k in ("snr_db", "frames", "decoded", "other_decodes", "gpu_equals_cpu", "gpu_frames_per_s",
                                                 "gpu_lazy_frames_per_s", "cpu_frames_per_s", "cpu_frames", "cpu_threads",
                                                 "lazy_records_resumed")} for r in sw["rows"]]}
            else:
                result["configs4"] = {
                    "what": "BASELINE configs[4] over %d ranks: examples/150613_1920.wav + AWGN, %d noisy copies per SNR sharded "
                            "round-robin (copy s on rank s mod %d, its noise a function of s), each rank: HOST audio in (PCIe "
                            "inclusive), K0 front-end, FDR, S0..S5 for up to 4 candidates per frame, Fano + unpack on ITS share "
                            "of the host's CPUs; frames / decodes summed over the ranks, seconds = the slowest rank's; "
                            "`gpu_equals_lazy` = eager and lazy (uwspr_set_tries(1) + uwspr_demod_resume) decode sets equal "
                            "on every rank; never `value`" % (world, c4_seeds, world),
                    "ranks": world, "ranks_on_this_host": local_world, "host_threads_per_rank": sw["host_threads"],
                    "native_snr_db": sw["native_snr_db"], "front_end": sw["front_end"],
                    "rows": [{"snr_db": snr_sweep.SNRS[i], "frames": int(cnt[i, 0]), "decoded": int(cnt[i, 1]),
                              "other_decodes": int(cnt[i, 2]), "lazy_records_resumed": int(cnt[i, 3]),
                              "gpu_equals_lazy": bool(okf[1 + i] == 1.0),
                              "gpu_frames_per_s": cnt[i, 0] / max(tim[i, 0], 1e-9),
                              "gpu_lazy_frames_per_s": cnt[i, 0] / max(tim[i, 1], 1e-9)} for i in range(nsn)]}
    if rank == 0:
        result["parity_spot_check"] = parity_spot_check(spot_snap, D, N, "fused" if fused else "staged") if spot_snap else None
        result["cpu_baseline"] = cpu_baseline(frames_cpu[:256]) if frames_cpu is not None else None
        print(json.dumps(result))
        sys.stdout.flush()
    if dist.is_initialized():
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
