"""Thin Python handle on a uwspr_ctx (include/uwspr_hip.h) for tests and bench.

Arrays in, arrays out; every number comes from the HIP library.  Frames may be
numpy arrays (host path: staged by the library) or torch CUDA tensors (device
path: pointers are passed through untouched).
"""
import ctypes as C

import numpy as np

from . import native as N


def _is_torch(x):
    return hasattr(x, "data_ptr") and hasattr(x, "is_cuda")


class FrameView:
    """B frames that are not a contiguous [B, fl, 2] array: either a raw device pointer (what
    stream_take_view returns: frames in place in the stream buffer) or a host array holding a stretch
    of the stream.  Frame b starts at 2*stride*b floats; pair with Context.set_frame_stride(stride)."""

    def __init__(self, B, ptr=None, host=None, device=0):
        self.B = int(B)
        self.ptr = ptr
        self.host = None if host is None else np.ascontiguousarray(host, dtype=np.float32)
        self._dev = device

    @property
    def device(self):
        import torch
        return torch.device("cuda", self._dev)


class Context:
    """Parameters mirror gr::uwspr::FDR::make / sync_and_demodulate::make
    (include/uwspr/FDR.h:49-50, include/uwspr/sync_and_demodulate.h:49)."""

    def __init__(self, fs=375, fl=45000, spb=256, maxdrift=0, maxfreqs=200, halfbandwidth=10,
                 cf=1500, threshold=10, device=0, options=None):
        """options: {name: int} for uwspr_set_option ("sched", "stage_kernels", "reuse", "phasor_tables",
        "fast_search", ...: include/uwspr_hip.h)."""
        self.L = N.lib()
        self.params = N.Params(fs, fl, spb, maxdrift, maxfreqs, halfbandwidth, cf, threshold)
        self.h = C.c_void_p()
        rc = self.L.uwspr_ctx_create(C.byref(self.params), device, C.byref(self.h))
        if rc != 0:
            msg = self.L.uwspr_last_error(self.h).decode() if self.h else \
                self.L.uwspr_status_string(rc).decode()
            if self.h:
                self.L.uwspr_ctx_destroy(self.h)
                self.h = C.c_void_p()
            raise N.UwsprError(rc, msg)
        self.info = N.Info()
        self._chk(self.L.uwspr_get_info(self.h, C.byref(self.info)))
        self.fl, self.maxfreqs = fl, maxfreqs
        self._keep = []
        self._stream_ptr = None
        for k, v in (options or {}).items():
            self.set_option(k, v)

    def set_option(self, name, value):
        self._chk(self.L.uwspr_set_option(self.h, name.encode(), int(value)))

    def get_option(self, name):
        v = C.c_int32()
        self._chk(self.L.uwspr_get_option(self.h, name.encode(), C.byref(v)))
        return v.value

    def close(self):
        if getattr(self, "h", None):
            self.L.uwspr_ctx_destroy(self.h)
            self.h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _chk(self, rc):
        if rc != 0:
            raise N.UwsprError(rc, self.L.uwspr_last_error(self.h).decode())

    # -- helpers -----------------------------------------------------------
    def _frames(self, frames):
        """-> (pointer, B, where, keepalive)"""
        if isinstance(frames, FrameView):      # B frames at a raw pointer, pitch per set_frame_stride
            if frames.host is not None:
                return C.c_void_p(frames.host.ctypes.data), frames.B, N.HOST, frames.host
            return C.c_void_p(frames.ptr), frames.B, N.DEVICE, frames
        if _is_torch(frames):
            assert frames.is_cuda and frames.is_contiguous() and frames.dtype.is_floating_point
            assert frames.element_size() == 4
            B = frames.numel() // (2 * self.fl)
            if self._stream_ptr is None:
                # the library runs on the context's own stream: whatever torch queued on
                # its current stream to produce `frames` must have finished first
                import torch
                torch.cuda.current_stream(frames.device).synchronize()
            return C.c_void_p(frames.data_ptr()), B, N.DEVICE, frames
        a = np.ascontiguousarray(frames, dtype=np.float32).reshape(-1, self.fl, 2)
        return C.c_void_p(a.ctypes.data), a.shape[0], N.HOST, a

    def set_stream(self, stream_ptr):
        """Run on a caller's hipStream_t (e.g. torch.cuda.Stream().cuda_stream); the caller
        then owns the ordering against its own work on that stream.  None/0 = own stream."""
        self._chk(self.L.uwspr_set_stream(self.h, C.c_void_p(stream_ptr)))
        self._stream_ptr = stream_ptr or None

    def synchronize(self):
        self._chk(self.L.uwspr_synchronize(self.h))

    def debug_snr_db(self, x):
        """diagnostics: 10 * log10f(x) as K2 computes a candidate's `snr` (FDR_impl.cc:303); x, result: float32 CUDA tensors"""
        import torch
        out = torch.empty_like(x)
        f = self.L.uwspr_debug_snr_db
        f.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_longlong]
        f.restype = C.c_int
        torch.cuda.current_stream().synchronize()
        self._chk(f(self.h, C.c_void_p(x.data_ptr()), C.c_void_p(out.data_ptr()), x.numel()))
        self.synchronize()
        return out

    # -- front-end ---------------------------------------------------------
    def frontend(self, audio):
        """12 kS/s real audio [B, nin] -> frames [B, fl, 2] at 375 S/s (uwspr_frontend_batch)."""
        if _is_torch(audio):
            import torch
            B, nin = audio.shape
            out = torch.empty((B, self.fl, 2), dtype=torch.float32, device=audio.device)
            if self._stream_ptr is None:
                torch.cuda.current_stream(audio.device).synchronize()
            self._chk(self.L.uwspr_frontend_batch(self.h, C.c_void_p(audio.data_ptr()), B, nin, N.DEVICE,
                                                  C.c_void_p(out.data_ptr())))
            if self._stream_ptr is None:
                self.synchronize()
            return out
        a = np.ascontiguousarray(audio, dtype=np.float32)
        a = a.reshape(1, -1) if a.ndim == 1 else a
        out = np.empty((a.shape[0], self.fl, 2), np.float32)
        self._chk(self.L.uwspr_frontend_batch(self.h, C.c_void_p(a.ctypes.data), a.shape[0], a.shape[1],
                                              N.HOST, C.c_void_p(out.ctypes.data)))
        return out

    # -- FDR ---------------------------------------------------------------
    def fdr_batch(self, frames):
        """-> list (per frame) of candidate record arrays, FDR_impl.cc:214-456."""
        p, B, where, keep = self._frames(frames)
        if where == N.DEVICE:
            import torch
            cands = torch.empty(B * self.maxfreqs * 48, dtype=torch.uint8, device=frames.device)
            npk = torch.empty(B, dtype=torch.int32, device=frames.device)
            self._chk(self.L.uwspr_fdr_batch(self.h, p, B, where, C.c_void_p(cands.data_ptr()),
                                             C.c_void_p(npk.data_ptr())))
            self.synchronize()
            cands = np.frombuffer(cands.cpu().numpy().tobytes(), N.CAND_DTYPE).reshape(B, -1)
            npk = npk.cpu().numpy()
        else:
            cands = np.zeros((B, self.maxfreqs), N.CAND_DTYPE)
            npk = np.zeros(B, np.int32)
            self._chk(self.L.uwspr_fdr_batch(self.h, p, B, where, C.c_void_p(cands.ctypes.data),
                                             C.c_void_p(npk.ctypes.data)))
        return [cands[b, :npk[b]].copy() for b in range(B)]

    def fdr_spectrum(self, B):
        i = self.info
        ps = np.empty((B, i.n, i.band_w), np.float32)
        psavg = np.empty((B, i.band_w), np.float32)
        smraw = np.empty((B, i.finpb), np.float32)
        smspec = np.empty((B, i.finpb), np.float32)
        noise = np.empty(B, np.float32)
        self._chk(self.L.uwspr_fdr_read_spectrum(self.h, B, *[C.c_void_p(a.ctypes.data) for a in
                                                              (ps, psavg, smraw, smspec, noise)]))
        return ps, psavg, smraw, smspec, noise

    def keep_syncgrid(self, ncand_cap):
        self._chk(self.L.uwspr_fdr_keep_syncgrid(self.h, ncand_cap))
        self._grid_cap = ncand_cap

    def fdr_syncgrid(self, B):
        g = np.empty((B, self._grid_cap, N.NIFR, N.NK0, self.info.cell_hyps), np.float32)
        self._chk(self.L.uwspr_fdr_read_syncgrid(self.h, B, C.c_void_p(g.ctypes.data)))
        return g

    # -- fine sweep --------------------------------------------------------
    def sync_sweep(self, frames, hyps, soft=True):
        """hyps: HYP_DTYPE array. -> (sync[H], symbols[H,162] or None)."""
        p, B, where, keep = self._frames(frames)
        hyps = np.ascontiguousarray(hyps, dtype=N.HYP_DTYPE)
        H = hyps.size
        if where == N.DEVICE:
            import torch
            dh = torch.from_numpy(np.frombuffer(hyps.tobytes(), np.uint8).copy()).to(frames.device)
            sync = torch.empty(H, dtype=torch.float32, device=frames.device)
            sym = torch.empty(H * N.NSYM, dtype=torch.uint8, device=frames.device) if soft else None
            self._chk(self.L.uwspr_sync_sweep(self.h, p, B, C.c_void_p(dh.data_ptr()), H, where,
                                              C.c_void_p(sync.data_ptr()),
                                              C.c_void_p(sym.data_ptr()) if soft else None))
            self.synchronize()
            return sync.cpu().numpy(), (sym.cpu().numpy().reshape(H, N.NSYM) if soft else None)
        sync = np.empty(H, np.float32)
        sym = np.empty((H, N.NSYM), np.uint8) if soft else None
        self._chk(self.L.uwspr_sync_sweep(self.h, p, B, C.c_void_p(hyps.ctypes.data), H, where,
                                          C.c_void_p(sync.ctypes.data),
                                          C.c_void_p(sym.ctypes.data) if soft else None))
        return sync, sym

    def sync_grid(self, frames, centres, df, ddrift, dlag, soft=True, into=None):
        """(freq, drift, lag) grid around one centre per frame (uwspr_sync_grid).
        -> sync [B,nf,ndrift,nlag], symbols [B,nf,ndrift,nlag,162] (numpy), or, with
        device frames and into=(sync_t, sym_t), results left in those torch tensors."""
        p, B, where, keep = self._frames(frames)
        df = np.ascontiguousarray(df, np.float32)
        ddrift = np.ascontiguousarray(ddrift, np.float32)
        dlag = np.ascontiguousarray(dlag, np.int32)
        shape = (B, df.size, ddrift.size, dlag.size)
        H = int(np.prod(shape))
        args = (df.size, C.c_void_p(df.ctypes.data), ddrift.size, C.c_void_p(ddrift.ctypes.data),
                dlag.size, C.c_void_p(dlag.ctypes.data))
        if where == N.DEVICE:
            import torch
            if _is_torch(centres):
                cent_t = centres
            else:
                cent = np.ascontiguousarray(centres, dtype=N.CAND_DTYPE)
                cent_t = torch.from_numpy(np.frombuffer(cent.tobytes(), np.uint8).copy()).to(frames.device)
            if into is None:
                sync_t = torch.empty(H, dtype=torch.float32, device=frames.device)
                sym_t = torch.empty(H * N.NSYM, dtype=torch.uint8, device=frames.device) if soft else None
            else:
                sync_t, sym_t = into
            self._chk(self.L.uwspr_sync_grid(self.h, p, B, where, C.c_void_p(cent_t.data_ptr()), *args,
                                             C.c_void_p(sync_t.data_ptr()),
                                             C.c_void_p(sym_t.data_ptr()) if sym_t is not None else None))
            if into is not None:
                return None
            self.synchronize()
            return (sync_t.cpu().numpy().reshape(shape),
                    sym_t.cpu().numpy().reshape(shape + (N.NSYM,)) if soft else None)
        cent = np.ascontiguousarray(centres, dtype=N.CAND_DTYPE)
        sync = np.empty(shape, np.float32)
        sym = np.empty(shape + (N.NSYM,), np.uint8) if soft else None
        self._chk(self.L.uwspr_sync_grid(self.h, p, B, where, C.c_void_p(cent.ctypes.data), *args,
                                         C.c_void_p(sync.ctypes.data),
                                         C.c_void_p(sym.ctypes.data) if soft else None))
        return sync, sym

    def sync_and_demodulate(self, frames, calls):
        """calls: CALL_DTYPE array, one per sync_and_demodulate() invocation
        (sync_and_demodulate_impl.cc:126-131). -> RESULT_DTYPE array."""
        p, B, where, keep = self._frames(frames)
        calls = np.ascontiguousarray(calls, dtype=N.CALL_DTYPE)
        res = np.zeros(calls.size, N.RESULT_DTYPE)
        self._chk(self.L.uwspr_sync_and_demodulate_batch(self.h, p, B, where,
                                                         C.c_void_p(calls.ctypes.data), calls.size,
                                                         C.c_void_p(res.ctypes.data)))
        return res

    # -- schedule ----------------------------------------------------------
    def demod_batch(self, frames, cands_per_frame, max_per_frame=1):
        p, B, where, keep = self._frames(frames)
        assert where == N.HOST, "demod_batch with device frames: use pipeline_batch"
        stride = max(1, max(len(c) for c in cands_per_frame))
        cands = np.zeros((B, stride), N.CAND_DTYPE)
        npk = np.zeros(B, np.int32)
        for b, c in enumerate(cands_per_frame):
            cands[b, :len(c)] = c
            npk[b] = len(c)
        out = np.zeros((B, max_per_frame), N.DEMOD_DTYPE)
        self._chk(self.L.uwspr_demod_batch(self.h, p, B, where, C.c_void_p(cands.ctypes.data),
                                           C.c_void_p(npk.ctypes.data), stride, max_per_frame,
                                           C.c_void_p(out.ctypes.data)))
        return out

    # ---- overlap-aware stream ingest (uwspr_stream_*) ----
    def stream_open(self, hop=3375, max_frames=256):
        self._chk(self.L.uwspr_stream_open(self.h, int(hop), int(max_frames)))

    def stream_push(self, iq):
        """Append (I,Q) samples (numpy [n,2] float32, or a torch CUDA tensor); -> frames ready."""
        n = C.c_int(0)
        if _is_torch(iq):
            self._chk(self.L.uwspr_stream_push(self.h, C.c_void_p(iq.data_ptr()), iq.numel() // 2, N.DEVICE, C.byref(n)))
        else:
            a = np.ascontiguousarray(iq, np.float32)
            self._chk(self.L.uwspr_stream_push(self.h, C.c_void_p(a.ctypes.data), a.size // 2, N.HOST, C.byref(n)))
        return n.value

    def stream_take(self, nframes, into):
        """The next nframes frames into a torch CUDA float32 tensor [nframes, fl, 2]; -> stream
        index of the first frame's first sample."""
        pos = C.c_longlong(0)
        fr = C.c_void_p()
        self._chk(self.L.uwspr_stream_take(self.h, int(nframes), C.c_void_p(into.data_ptr()), C.byref(fr), C.byref(pos)))
        return pos.value

    def stream_take_view(self, nframes):
        """The next nframes frames IN PLACE -> (device pointer, stride in samples, stream index of
        the first frame).  Frame j starts at pointer + 8*stride*j bytes; valid until the next take.
        Use with set_frame_stride(stride) and frames_ptr=... of the *_into calls."""
        pos = C.c_longlong(0)
        fr = C.c_void_p()
        st = C.c_int(0)
        self._chk(self.L.uwspr_stream_take_view(self.h, int(nframes), C.byref(fr), C.byref(st), C.byref(pos)))
        return fr.value, st.value, pos.value

    def stream_wait_uploads(self):
        self._chk(self.L.uwspr_stream_wait_uploads(self.h))

    def set_frame_stride(self, stride=0):
        """Frame pitch in samples of the `frames` argument of the calls that follow (0 = fl)."""
        self._chk(self.L.uwspr_set_frame_stride(self.h, int(stride)))

    def set_tries(self, ntries):
        """Mode-2 tries per candidate the schedule calls produce (17 = all; fewer = lazy)."""
        self._chk(self.L.uwspr_set_tries(self.h, int(ntries)))

    def demod_resume(self, frames, need, out, max_per_frame=1):
        """Produce all 17 tries for the slots flagged in `need` (uint8 [B, max_per_frame]) of the
        last schedule call.  Host form: returns the full record array; device form (torch
        tensors for need / out): in place."""
        p, B, where, keep = self._frames(frames)
        if where == N.DEVICE:
            self._chk(self.L.uwspr_demod_resume(self.h, p, B, where, C.c_void_p(need.data_ptr()),
                                                max_per_frame, C.c_void_p(out.data_ptr())))
            return out
        need = np.ascontiguousarray(need, np.uint8).reshape(B, max_per_frame)
        res = np.zeros((B, max_per_frame), N.DEMOD_DTYPE)
        self._chk(self.L.uwspr_demod_resume(self.h, p, B, where, C.c_void_p(need.ctypes.data),
                                            max_per_frame, C.c_void_p(res.ctypes.data)))
        return res

    def pipeline_batch(self, frames, max_per_frame=1, fetch=True):
        """FDR + refinement schedule. -> (cands list, demod_out[B,max_per_frame]) or None."""
        p, B, where, keep = self._frames(frames)
        if where == N.DEVICE:
            import torch
            if not fetch:
                self._chk(self.L.uwspr_pipeline_batch(self.h, p, B, where, max_per_frame, None,
                                                      None, None))
                return None
            dev = frames.device
            cands = torch.empty(B * self.maxfreqs * 48, dtype=torch.uint8, device=dev)
            npk = torch.empty(B, dtype=torch.int32, device=dev)
            out = torch.empty(B * max_per_frame * N.DEMOD_DTYPE.itemsize, dtype=torch.uint8,
                              device=dev)
            self._chk(self.L.uwspr_pipeline_batch(self.h, p, B, where, max_per_frame,
                                                  C.c_void_p(cands.data_ptr()),
                                                  C.c_void_p(npk.data_ptr()),
                                                  C.c_void_p(out.data_ptr())))
            self.synchronize()
            cands = np.frombuffer(cands.cpu().numpy().tobytes(), N.CAND_DTYPE).reshape(B, -1)
            npk = npk.cpu().numpy()
            out = np.frombuffer(out.cpu().numpy().tobytes(), N.DEMOD_DTYPE).reshape(B, -1)
        else:
            cands = np.zeros((B, self.maxfreqs), N.CAND_DTYPE)
            npk = np.zeros(B, np.int32)
            out = np.zeros((B, max_per_frame), N.DEMOD_DTYPE)
            self._chk(self.L.uwspr_pipeline_batch(self.h, p, B, where, max_per_frame,
                                                  C.c_void_p(cands.ctypes.data),
                                                  C.c_void_p(npk.ctypes.data),
                                                  C.c_void_p(out.ctypes.data)))
        return [cands[b, :npk[b]].copy() for b in range(B)], out.copy()

    def pipeline_batch_into(self, frames, cands_t, npk_t, out_t, max_per_frame=1):
        """Device-resident form used by bench.py: frames and the three output
        buffers are torch CUDA tensors; nothing is copied to the host and the call
        returns as soon as the work is enqueued on the context's stream."""
        p, B, where, keep = self._frames(frames)
        assert where == N.DEVICE
        assert cands_t.numel() * cands_t.element_size() >= B * self.maxfreqs * 48
        assert out_t.numel() * out_t.element_size() >= B * max_per_frame * N.DEMOD_DTYPE.itemsize
        self._chk(self.L.uwspr_pipeline_batch(self.h, p, B, where, max_per_frame,
                                              C.c_void_p(cands_t.data_ptr()),
                                              C.c_void_p(npk_t.data_ptr()),
                                              C.c_void_p(out_t.data_ptr())))

    def pipeline_slabs(self, K, slab_t):
        """The next pipeline_batch* call also writes its frames' gather slabs into slab_t (torch CUDA uint8
        [B, 32+48K]); None cancels."""
        self._chk(self.L.uwspr_pipeline_slabs(self.h, int(K), C.c_void_p(slab_t.data_ptr()) if slab_t is not None else None))

    def pack_slabs_into(self, B, K, slab_t):
        """Per-frame gather slabs of the last pipeline batch, written into a torch
        CUDA uint8 tensor [B, 32+48K] (or a numpy array for the host form)."""
        if _is_torch(slab_t):
            self._chk(self.L.uwspr_pack_slabs(self.h, B, K, C.c_void_p(slab_t.data_ptr()), N.DEVICE))
        else:
            self._chk(self.L.uwspr_pack_slabs(self.h, B, K, C.c_void_p(slab_t.ctypes.data), N.HOST))

    def sync_sweep_into(self, frames, hyps_t, H, sync_t, sym_t):
        p, B, where, keep = self._frames(frames)
        assert where == N.DEVICE
        self._chk(self.L.uwspr_sync_sweep(self.h, p, B, C.c_void_p(hyps_t.data_ptr()), H, where,
                                          C.c_void_p(sync_t.data_ptr()),
                                          C.c_void_p(sym_t.data_ptr()) if sym_t is not None else None))

    # -- multi-GPU gather over RCCL (uwspr_dist_*) ---------------------------
    @staticmethod
    def dist_unique_id():
        """rank 0: the 128 bytes every rank passes to dist_init (ncclGetUniqueId)."""
        buf = (C.c_char * 128)()
        rc = N.lib().uwspr_dist_unique_id(buf)
        if rc != 0:
            raise N.UwsprError(rc, "uwspr_dist_unique_id: RCCL unavailable")
        return bytes(buf)

    def dist_init(self, rank, world, uid=None):
        self._chk(self.L.uwspr_dist_init(self.h, int(rank), int(world), uid))

    def dist_gather(self, send_t, recv_t=None, root=0):
        """send_t: torch CUDA tensor (this rank's slabs); recv_t: [world * send bytes] on the root."""
        nbytes = send_t.numel() * send_t.element_size()
        self._chk(self.L.uwspr_dist_gather(self.h, C.c_void_p(send_t.data_ptr()), nbytes,
                                           C.c_void_p(recv_t.data_ptr()) if recv_t is not None else None,
                                           int(root), N.DEVICE))

    def dist_finalize(self):
        self._chk(self.L.uwspr_dist_finalize(self.h))

    # -- measurement -------------------------------------------------------
    def prof_enable(self, which=True):
        """which: True = every kernel family, False/0 = off, or an iterable of
        family names from native.K_NAMES (e.g. ("tonecorr",))."""
        if which is True:
            mask = 0x3F
        elif not which:
            mask = 0
        else:
            mask = sum(1 << N.K_NAMES.index(k) for k in which)
        self._chk(self.L.uwspr_prof_enable(self.h, mask))

    def prof_intervals(self, kind, epoch_event_ptr, cap=4096):
        """(start_ms, stop_ms) arrays of the recorded launches of one family, relative
        to a caller-recorded hipEvent (e.g. torch.cuda.Event(enable_timing=True).cuda_event)."""
        a = np.zeros(cap, np.float64)
        b = np.zeros(cap, np.float64)
        n = C.c_int(0)
        self._chk(self.L.uwspr_prof_intervals(self.h, N.K_NAMES.index(kind), C.c_void_p(epoch_event_ptr),
                                              C.c_void_p(a.ctypes.data), C.c_void_p(b.ctypes.data),
                                              cap, C.byref(n)))
        return a[:n.value].copy(), b[:n.value].copy()

    def prof_read(self):
        p = N.Prof()
        self._chk(self.L.uwspr_prof_read(self.h, C.byref(p)))
        return {k: {"ms": p.ms[i], "launches": p.launches[i], "units": p.units[i]}
                for i, k in enumerate(N.K_NAMES)}


# ---- host tail (no device needed) -------------------------------------------
def deinterleave(symbols):
    s = np.array(symbols, dtype=np.uint8).copy()
    N.lib().uwspr_deinterleave(C.c_void_p(s.ctypes.data))
    return s


def fano_decode(symbols, delta=60, maxcycles=10000):
    s = np.ascontiguousarray(symbols, dtype=np.uint8)
    data = np.zeros(11, np.uint8)
    metric, cycles, maxnp = C.c_uint32(), C.c_uint32(), C.c_uint32()
    rc = N.lib().uwspr_fano_decode(C.c_void_p(s.ctypes.data), C.c_void_p(data.ctypes.data),
                                   C.byref(metric), C.byref(cycles), C.byref(maxnp), delta,
                                   maxcycles)
    return rc, data, metric.value, cycles.value


def fano_encode(data_bytes):
    d = np.ascontiguousarray(data_bytes, dtype=np.uint8)
    out = np.zeros(d.size * 16, np.uint8)
    N.lib().uwspr_fano_encode(C.c_void_p(out.ctypes.data), C.c_void_p(d.ctypes.data), d.size)
    return out


def decode_candidate(demod_rec):
    """demod_rec: one DEMOD_DTYPE record. -> (message7 int8 array, idt) or None."""
    rec = np.ascontiguousarray(np.array(demod_rec, dtype=N.DEMOD_DTYPE).reshape(1))
    msg = np.zeros(7, np.int8)
    idt = C.c_int32(-1)
    ok = N.lib().uwspr_decode_candidate(C.c_void_p(rec.ctypes.data), C.c_void_p(msg.ctypes.data),
                                        C.byref(idt))
    return (msg, idt.value) if ok else None


def host_threads():
    """CPUs this process may keep busy (affinity and cgroup quota applied)."""
    return int(N.lib().uwspr_host_threads())


def host_set_ranks(ranks):
    """The `ranks` processes of a job that share this host (LOCAL_WORLD_SIZE) share its CPUs: host_threads() and the
    process-wide Fano pool become that share.  Before the first decode_batch / Pipe of the process."""
    rc = N.lib().uwspr_host_set_ranks(int(ranks))
    if rc != 0:
        raise N.UwsprError(rc, "uwspr_host_set_ranks(%d): %s" % (ranks, "the host pool already exists" if rc == -3 else "ranks < 1"))


def decode_batch(demod_recs, nthreads=0):
    """demod_recs: DEMOD_DTYPE array (any shape). -> (messages [n,7] int8, idt [n] int32,
    decoded [n] bool), record order kept; Fano runs on `nthreads` host threads."""
    rec = np.ascontiguousarray(np.asarray(demod_recs, dtype=N.DEMOD_DTYPE).reshape(-1))
    n = rec.size
    msg = np.zeros((n, 7), np.int8)
    idt = np.full(n, -1, np.int32)
    ok = np.zeros(n, np.uint8)
    rc = N.lib().uwspr_decode_batch(C.c_void_p(rec.ctypes.data), n, nthreads, C.c_void_p(msg.ctypes.data),
                                    C.c_void_p(idt.ctypes.data), C.c_void_p(ok.ctypes.data))
    if rc < 0:
        raise N.UwsprError(rc, "uwspr_decode_batch")
    return msg, idt, ok.astype(bool)


def unpack_message(message7):
    m = np.ascontiguousarray(message7, dtype=np.int8)
    buf = C.create_string_buffer(32)
    rc = N.lib().uwspr_unpack_message(C.c_void_p(m.ctypes.data), buf, 32)
    return rc, buf.value.decode("ascii", "replace")


FRONTEND_GRC, FRONTEND_COMPACT = 0, 1


def host_alloc(nbytes):
    """uwspr_host_alloc: page-locked host memory as a writable ctypes byte array (np.frombuffer it); release with
    host_free."""
    ptr = C.c_void_p()
    rc = N.lib().uwspr_host_alloc(C.c_size_t(nbytes), C.byref(ptr))
    if rc:
        raise N.UwsprError(rc, "uwspr_host_alloc(%d)" % nbytes)
    buf = (C.c_uint8 * nbytes).from_address(ptr.value)
    buf._uwspr_ptr = ptr.value
    return buf


def host_free(buf):
    N.lib().uwspr_host_free(C.c_void_p(buf._uwspr_ptr))


def frontend_design(mode=FRONTEND_GRC, stage=0):
    """uwspr_frontend_design: stage 0 -> (complex128 composite taps g, read-ahead D) of the K0 front-end
    y[m] = sum_k g[k] x[32 m + D - k]; stages 1..3 (grc mode) -> the band-pass, low-pass and resampler designs."""
    d = C.c_int32(0)
    n = N.lib().uwspr_frontend_design(mode, stage, None, 0, C.byref(d))
    if n < 0:
        raise N.UwsprError(n, "uwspr_frontend_design(mode=%d, stage=%d)" % (mode, stage))
    g = np.zeros(n * (2 if stage == 0 else 1), np.float64)
    N.lib().uwspr_frontend_design(mode, stage, C.c_void_p(g.ctypes.data), n, C.byref(d))
    if stage == 0:
        return g[0::2] + 1j * g[1::2], int(d.value)
    return g


def c2_read(path):
    iq = np.zeros((45000, 2), np.float32)
    freq, typ = C.c_double(), C.c_int32()
    rc = N.lib().uwspr_c2_read(path.encode(), C.c_void_p(iq.ctypes.data), C.byref(freq),
                               C.byref(typ))
    if rc != 0:
        raise N.UwsprError(rc, "cannot read %s" % path)
    return iq, freq.value, typ.value


class Pipe:
    """uwspr_pipe_*: the pipelined end-to-end decoder (stream ingest on a copy stream, lazy schedule,
    Fano on the persistent host pool under the next batch's kernels, resume of what try 0 did not
    decode).  Results are DECODE_DTYPE records in frame order."""

    def __init__(self, fs=375, fl=45000, spb=256, maxdrift=0, maxfreqs=200, halfbandwidth=10, cf=1500,
                 threshold=10, device=0, hop=3375, batch_frames=256, max_per_frame=1, lanes=0,
                 host_threads=0, eager=False, sched=None, spare_after_us=0):
        self.L = N.lib()
        self.h = C.c_void_p()
        self.fl = fl
        p = N.Params(fs, fl, spb, maxdrift, maxfreqs, halfbandwidth, cf, threshold)
        o = N.PipeOpts(hop, batch_frames, max_per_frame, lanes, host_threads, 1 if eager else 0,
                       {None: 0, "fused": 1, "staged": 2}[sched], int(spare_after_us))
        rc = self.L.uwspr_pipe_open(C.byref(p), device, C.byref(o), C.byref(self.h))
        if rc != 0:
            msg = self.L.uwspr_pipe_last_error(self.h).decode() if self.h else ""
            if self.h:
                self.L.uwspr_pipe_close(self.h)
            self.h = None
            raise N.UwsprError(rc, msg)
        self.batch_frames, self.hop = batch_frames, hop

    def _chk(self, rc):
        if rc < 0:
            raise N.UwsprError(rc, self.L.uwspr_pipe_last_error(self.h).decode())
        return rc

    def close(self):
        if self.h:
            self.L.uwspr_pipe_close(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def push(self, iq):
        a = np.ascontiguousarray(iq, np.float32)
        self._chk(self.L.uwspr_pipe_push(self.h, C.c_void_p(a.ctypes.data), a.size // 2))

    def acquire(self, nsamples):
        """-> numpy view [nsamples, 2] of the page-locked staging buffer to fill; then commit(nsamples)."""
        ptr = C.c_void_p()
        self._chk(self.L.uwspr_pipe_acquire(self.h, int(nsamples), C.byref(ptr)))
        buf = (C.c_float * (2 * int(nsamples))).from_address(ptr.value)
        return np.frombuffer(buf, np.float32).reshape(-1, 2)

    def commit(self, nsamples):
        self._chk(self.L.uwspr_pipe_commit(self.h, int(nsamples)))

    def submit_device(self, frames, B=None, stride=0):
        """frames: torch CUDA tensor [B, fl, 2] (or a raw pointer with B given)."""
        if _is_torch(frames):
            B = frames.numel() // (2 * self.fl) if B is None else B
            ptr = frames.data_ptr()
        else:
            ptr = int(frames)
        self._chk(self.L.uwspr_pipe_submit_device(self.h, C.c_void_p(ptr), int(B), int(stride)))

    def flush(self):
        self._chk(self.L.uwspr_pipe_flush(self.h))

    def collect(self, cap=65536, wait=False):
        out = np.zeros(cap, N.DECODE_DTYPE)
        n = self._chk(self.L.uwspr_pipe_collect(self.h, C.c_void_p(out.ctypes.data), cap, 1 if wait else 0))
        return out[:n].copy()

    def inject_failure(self, batch, where):
        """test hook: batch number `batch` fails at its launch (where = 0) or in its host tail (where = 1)"""
        self._chk(self.L.uwspr_pipe_inject_failure(self.h, int(batch), int(where)))

    def set_option(self, name, value):
        """uwspr_pipe_set_option: an option of every lane's context; only while nothing is in flight"""
        self._chk(self.L.uwspr_pipe_set_option(self.h, name.encode(), int(value)))

    def stats(self):
        st = N.PipeStats()
        self._chk(self.L.uwspr_pipe_get_stats(self.h, C.byref(st)))
        return {k: getattr(st, k) for k, _ in N.PipeStats._fields_}
