#!/usr/bin/env python3
"""bench.py -- frames/s of the coarse (FDR) + fine (sync_and_demodulate) path.

    python bench.py --gpus N --steps K --warmup W

One step = one pass of the whole hot path (K1 spectrogram, K2 spectrum/peaks,
K3 coarse search + selection over ALL candidates, then the S0..S5 refinement
schedule incl. the 17 soft-symbol vectors for the top candidate of every frame)
over one batch of synthetic frames that is already resident in HBM, followed by
the gather of the per-frame candidate slabs to rank 0.  Workload = BASELINE.json
configs[1]: 256 synthetic 375 Hz / 45000-sample frames per GPU at -20 dB, flowgraph
default FDR grid, single candidate per frame.  For N>1 every rank runs its own
256 frames (weak scaling); frames shard with no data-path collective.

Prints ONE JSON line (rank 0).  Extra objects:
  roofline      the dominant kernel (K4 tone correlation sweep): algorithmic
                bytes/launch (331950 B per hypothesis) / HIP-event
                time per launch, against the 8 TB/s HBM peak.
  cpu_baseline  the CPU restatement (oracle/, kind "port") on a bounded sample
                of the same workload on this box's host cores.
  kernels       HIP-event time per kernel family per step.
  sweep         (N=1) BASELINE configs[2]: 1024 frames x 200 (freq,lag,drift)
                hypotheses through uwspr_sync_sweep, the north_star's roofline case.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
for _p in (ROOT, os.path.join(ROOT, "oracle")):
    if _p not in sys.path:
        sys.path.insert(0, _p)

HYP_BYTES = 162 * 256 * 8 + 162 + 12      # 331950 B per fine hypothesis
HYP_FLOP = 162 * 4 * 256 * 14             # binary32 ops per hypothesis in K4
HBM_PEAK_GBS = 8000.0
FP32_NOFMA_PEAK_TOPS = 78.6               # 256 CU x 4 SIMD x 32 lanes x 2.4 GHz, one op/lane/clk


def cpu_baseline(frames_np, budget_s=12.0):
    """Oracle (CPU restatement) on a bounded sample: FDR over all candidates +
    the schedule for the top candidate, one frame per worker thread (the C code
    is re-entrant and ctypes releases the GIL)."""
    from concurrent.futures import ThreadPoolExecutor
    import oracle_py as O
    O.lib()
    O.pr3()
    ncores = max(1, min(16, len(os.sched_getaffinity(0))))
    fdrs = [O.FDR() for _ in range(ncores)]

    def one(args):
        w, b = args
        c = fdrs[w].transform(frames_np[b])
        if len(c):
            O.demod_candidate(c[0], 1500, frames_np[b])
        return 1

    t0 = time.time()
    one((0, 0))
    per = max(time.time() - t0, 1e-3)
    nb = frames_np.shape[0]
    n = int(max(ncores, budget_s * ncores / per))      # ~budget_s seconds of wall time
    t0 = time.time()
    with ThreadPoolExecutor(ncores) as ex:
        list(ex.map(one, [(i % ncores, i % nb) for i in range(n)]))
    dt = time.time() - t0
    return {"value": n / dt, "unit": "frames/s", "cores": ncores, "kind": "port",
            "sample": "%d frame-passes over the benchmark's %d frames (oracle FDR over all candidates + "
                      "S0..S5 schedule with 17 soft-symbol vectors for the top candidate), %d threads, "
                      "%.1f s wall = %.0f core-seconds" % (n, nb, ncores, dt, dt * ncores)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--frames", type=int, default=256, help="frames per GPU per step")
    ap.add_argument("--snr", type=float, default=-20.0)
    ap.add_argument("--no-sweep", action="store_true")
    ap.add_argument("--prof-steps", type=int, default=0,
                    help="timed steps whose K4 launches are bracketed by HIP events (0 = all)")
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--sweep-frames", type=int, default=1024)
    ap.add_argument("--streams", type=int, default=3,
                    help="HIP streams (each with its own context and scratch) the steps rotate over, "
                         "so the tail of one batch overlaps the head of the next")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist
    import gr_uwspr_amd as G
    from gr_uwspr_amd import dist as D
    N = G.native

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    assert torch.cuda.is_available(), "bench.py needs a GPU (there is no CPU fallback)"
    local = min(local, torch.cuda.device_count() - 1)   # rehearsal: several ranks on one GPU (gloo)
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    if world > 1 or os.environ.get("UWSPR_BENCH_FORCE_PG"):
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        backend = os.environ.get("UWSPR_BENCH_BACKEND", "nccl")
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)

    B = args.frames
    frames = G.synth.make_frames_torch(B, dev, seed=0xC0FFEE + 7919 * rank, snr_db=args.snr)
    ns = max(1, args.streams)
    lanes = []
    for k in range(ns):
        st = torch.cuda.Stream(device=dev)
        cx = G.Context(device=local)
        cx.set_stream(st.cuda_stream)
        lanes.append({"stream": st, "ctx": cx,
                      "cands": torch.empty(B * cx.maxfreqs * 48, dtype=torch.uint8, device=dev),
                      "npk": torch.empty(B, dtype=torch.int32, device=dev),
                      "out": torch.empty(B * N.DEMOD_DTYPE.itemsize, dtype=torch.uint8, device=dev),
                      "slab": torch.empty((B, D.SLAB_BYTES), dtype=torch.uint8, device=dev)})
    ctx, cands_t, npk_t, out_t = lanes[0]["ctx"], lanes[0]["cands"], lanes[0]["npk"], lanes[0]["out"]
    torch.cuda.synchronize()
    step_no = [0]

    def step():
        ln = lanes[step_no[0] % ns]
        step_no[0] += 1
        with torch.cuda.stream(ln["stream"]):
            ln["ctx"].pipeline_batch_into(frames, ln["cands"], ln["npk"], ln["out"], max_per_frame=1)
            ln["ctx"].pack_slabs_into(B, D.SLAB_K, ln["slab"])
            return D.gather_slabs(ln["slab"], dst=0)

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    barrier()
    # timed region: HIP events only around the dominant kernel (K4); a second,
    # untimed pass of the same K steps records every family for the breakdown
    def prof_all(which):
        for ln in lanes:
            ln["ctx"].prof_enable(which)
            ln["ctx"].prof_read()

    def prof_sum():
        tot = None
        for ln in lanes:
            p = ln["ctx"].prof_read()
            if tot is None:
                tot = p
            else:
                for k in p:
                    for f in p[k]:
                        tot[k][f] += p[k][f]
        return tot

    prof_all(("tonecorr",))
    epoch = torch.cuda.Event(enable_timing=True)
    epoch.record()
    # the K4 launches of the first `prof_steps` timed steps carry HIP-event stamps (each stamped
    # launch costs the queue a ~5 us bubble on either side: tools/trace_gaps.py); the roofline
    # is taken over that window, `value` over all K steps
    prof_steps = args.steps if args.prof_steps <= 0 else min(args.steps, args.prof_steps)
    t0 = time.perf_counter()
    for i in range(args.steps):
        if i == prof_steps:
            for ln in lanes:
                ln["ctx"].prof_enable(False)
        gathered = step()
    t_enq = time.perf_counter() - t0     # host time to enqueue the K steps (before any wait)
    barrier()
    dt = time.perf_counter() - t0
    # K4 launches of different lanes may overlap in time: the family's busy time is
    # the union of their [start, stop] intervals (HIP events, common epoch)
    iv = []
    for ln in lanes:
        a, b = ln["ctx"].prof_intervals("tonecorr", epoch.cuda_event)
        iv += list(zip(a, b))
    iv.sort()
    k4_busy_ms, cur_a, cur_b = 0.0, None, None
    for a, b in iv:
        if cur_b is None or a > cur_b:
            if cur_b is not None:
                k4_busy_ms += cur_b - cur_a
            cur_a, cur_b = a, b
        else:
            cur_b = max(cur_b, b)
    if cur_b is not None:
        k4_busy_ms += cur_b - cur_a
    prof_k4 = prof_sum()
    prof_all(True)
    t1 = time.perf_counter()
    for _ in range(args.steps):
        step()
    barrier()
    dt_all_events = time.perf_counter() - t1
    prof = prof_sum()
    prof["tonecorr"] = prof_k4["tonecorr"]
    prof_all(False)
    # single-stream pass (untimed for `value`): K4 launch durations without other streams'
    # kernels sharing the CUs; this is the figure a `--streams 1` rocprofv3 trace shows
    k4_single = None
    if ns > 1:
        lanes[0]["ctx"].prof_enable(("tonecorr",))
        lanes[0]["ctx"].prof_read()
        n1 = min(args.steps, 20)
        for _ in range(n1):
            step_no[0] = 0
            step()
        barrier()
        k4_single = lanes[0]["ctx"].prof_read()["tonecorr"]
        k4_single["steps"] = n1
        lanes[0]["ctx"].prof_enable(False)
    if world > 1:
        tt = torch.tensor([dt], dtype=torch.float64,
                          device="cpu" if dist.get_backend() == "gloo" else dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())

    # ---- what the batch contained (outside the timed region) ----------------
    cands = np.frombuffer(cands_t.cpu().numpy().tobytes(), N.CAND_DTYPE).reshape(B, -1)
    npk = npk_t.cpu().numpy()
    out = np.frombuffer(out_t.cpu().numpy().tobytes(), N.DEMOD_DTYPE).reshape(B, 1)
    top_lin = np.array([cands[b, 0]["m_type"] == 0 if npk[b] > 0 else False for b in range(B)])
    live = npk > 0
    worth = out[:, 0]["worth_a_try"] > 0
    fine_hyps = int(10 * live.sum() + 2 * (live & top_lin).sum() + 27 * worth.sum())
    # the library skips the repeat of the stage winner in S1, S3 and S4 (stage-winner
    # reuse): hypotheses resolved per step stay the reference's 39 per candidate, correlations run are fewer
    reuse_on = os.environ.get("UWSPR_K4_REUSE", "1") != "0"
    fine_corr = fine_hyps - (int(live.sum() + 2 * worth.sum()) if reuse_on else 0)
    # host tail: deinterleave + Fano of the batch's top candidates on the
    # host cores; reported beside `value`, never inside the timed region
    nthr = min(16, len(os.sched_getaffinity(0)))
    G.decode_batch(out[:8, 0], nthreads=nthr)
    t_h = time.perf_counter()
    _, _, okv = G.decode_batch(out[:, 0], nthreads=nthr)
    host_tail = {"records": int(B), "decoded": int(okv.sum()), "threads": nthr,
                 "records_per_s": B / (time.perf_counter() - t_h)}
    decoded = int(okv[:64].sum())

    frames_cpu = frames.cpu().numpy() if (rank == 0 and world == 1) else None
    # the same batch handed over as HOST buffers (what a GNU Radio block would do):
    # pageable H2D of the frames + D2H of every result, PCIe inclusive; never `value`
    host_rate = None
    if frames_cpu is not None:
        ctx.pipeline_batch(frames_cpu, max_per_frame=1)
        t2 = time.perf_counter()
        for _ in range(3):
            ctx.pipeline_batch(frames_cpu, max_per_frame=1)
        host_rate = 3 * B / (time.perf_counter() - t2)
    if args.no_cpu:
        frames_cpu = None
    result = None
    traffic = None
    tpath = os.path.join(ROOT, "profiles", "k4_traffic.json")
    if os.path.exists(tpath) and B == 256:   # PMC-measured with rocprofv3 on this workload (separate passes)
        traffic = json.load(open(tpath))
    if rank == 0:
        k4 = prof["tonecorr"]
        k4_launch_ms = k4["ms"] / max(k4["launches"], 1)
        k4_bytes_per_launch = fine_corr * HYP_BYTES / 6.0      # 6 K4 launches per step; correlations actually run
        # achieved = algorithmic bytes of all K4 launches / time during which K4 was running
        achieved = (fine_corr * HYP_BYTES * prof_steps) / (k4_busy_ms * 1e-3) / 1e9 if k4_busy_ms > 0 else 0.0
        kern = {k: {"ms_per_step": v["ms"] / (prof_steps if k == "tonecorr" else args.steps),
                    "launches_per_step": v["launches"] / (prof_steps if k == "tonecorr" else args.steps)}
                for k, v in prof.items()}
        result = {
            "metric": "2-min WSPR frames decoded/sec (coarse+sync)",
            "value": world * B * args.steps / dt,
            "unit": "frames/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": 1e3 * dt / args.steps,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "config": {"workload": "BASELINE configs[1]: %d synthetic .c2 frames (375 Hz, 45000 samples, "
                                   "SNR %.0f dB) per GPU, flowgraph-default FDR grid (hbw=10, maxdrift=0, "
                                   "threshold=10, all candidates searched), S0..S5 schedule + 17 soft-symbol "
                                   "vectors for the top candidate of each frame, slab gather to rank 0"
                                   % (B, args.snr),
                       "frames_per_gpu": B, "candidates_per_frame_mean": float(npk.mean()),
                       "fine_hypotheses_per_step": fine_hyps,
                       "fine_correlations_run_per_step": fine_corr,
                       "coarse_hypotheses_per_step": int(npk.sum()) * 130 * ctx.info.cell_hyps,
                       "top_candidate_decodes_in_first_64": decoded, "parallelism": "dp%d" % world,
                       "streams_per_gpu": ns},
            "roofline": {"kernel": "k4_tonecorr", "bound": "hbm", "achieved": achieved,
                         "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                         "traffic": traffic["bytes_per_launch"] if traffic else None,
                         "traffic_source": traffic["source"] if traffic else None,
                         "bytes_per_launch_algorithmic": k4_bytes_per_launch,
                         "avg_launch_ms": k4_launch_ms,
                         "k4_busy_ms_per_step": k4_busy_ms / prof_steps,
                         "steps_with_k4_events": prof_steps,
                         "k4_sum_of_launch_ms_per_step": k4["ms"] / prof_steps,
                         "accounting": "launches from %d streams may overlap: achieved = bytes / union of the "
                                       "launches' HIP-event intervals; bytes = 331950 B x the correlations the "
                                       "launches actually run (config.fine_correlations_run_per_step), not the "
                                       "reference's 39 per candidate" % ns,
                         "single_stream": None if not k4_single else {
                             "avg_launch_ms": k4_single["ms"] / max(k4_single["launches"], 1),
                             "k4_ms_per_step": k4_single["ms"] / k4_single["steps"],
                             "achieved": fine_corr * HYP_BYTES * k4_single["steps"] / (k4_single["ms"] * 1e-3) / 1e9,
                             "frac": fine_corr * HYP_BYTES * k4_single["steps"] / (k4_single["ms"] * 1e-3) / 1e9 / HBM_PEAK_GBS,
                             "note": "same step on one stream: launch durations comparable with "
                                     "profiles/*_streams1 rocprofv3 kernel trace"},
                         "fp32_tops": fine_corr * HYP_FLOP * prof_steps / (k4_busy_ms * 1e-3) / 1e12 if k4_busy_ms > 0 else 0.0,
                         "fp32_nofma_peak_tops": FP32_NOFMA_PEAK_TOPS},
            "kernels": kern,
            "host_pointer_frames_per_s_pcie_inclusive": host_rate,
            "host_tail_fano": host_tail,
            "host_enqueue_ms_per_step": 1e3 * t_enq / args.steps,
            "ms_per_step_with_events_on_every_kernel": 1e3 * dt_all_events / args.steps,
        }

    # ---- configs[2]: the (freq, lag, drift) sweep, N=1 only -----------------
    if rank == 0 and world == 1 and not args.no_sweep:
        from gr_uwspr_amd import sweep as SW
        Bs = args.sweep_frames
        del frames
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

        def timed(fn, reps=3):
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
            return {"ms_total": 1e3 * dts, "k4_ms": k4ms, "k5_ms": p2["fold"]["ms"] / reps,
                    "k4_launches": p2["tonecorr"]["launches"] / reps, "hyps_per_s": H / dts,
                    "algorithmic_GBs_k4": H * HYP_BYTES / (k4ms * 1e-3) / 1e9,
                    "frac_of_hbm_peak_k4": H * HYP_BYTES / (k4ms * 1e-3) / 1e9 / HBM_PEAK_GBS}

        # grid form (uwspr_sync_grid): shared symbol windows + shared tone phasors
        sweep["grid"] = timed(lambda: ctx.sync_grid(fr2, cent_t, df, dd, dl, into=(sync_t, sym_t)))
        grid_sync = sync_t.clone()
        # flat form (uwspr_sync_sweep) on the same 204800 hypotheses, same order
        hy = G.sweep_grid_uniform(Bs, f_c=0.0, shift_c=368)
        hy = hy.reshape(Bs, 5, 8, 5).transpose(0, 1, 3, 2).reshape(-1).copy()
        hy_t = torch.from_numpy(np.frombuffer(hy.tobytes(), np.uint8).copy()).to(dev)
        sweep["flat"] = timed(lambda: ctx.sync_sweep_into(fr2, hy_t, H, sync_t, sym_t))
        sweep["flat"]["fp32_tops_k4"] = H * HYP_FLOP / (sweep["flat"]["k4_ms"] * 1e-3) / 1e12
        sweep["grid_equals_flat_bitwise"] = bool(torch.equal(grid_sync, sync_t))
        result["sweep"] = sweep

    if rank == 0:
        result["cpu_baseline"] = cpu_baseline(frames_cpu) if frames_cpu is not None else None
        print(json.dumps(result))
    if dist.is_initialized():
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
