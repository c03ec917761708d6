/*
 * uwspr_hip.h -- C ABI of the MI355X (gfx950) implementation of gr-uwspr's
 * coarse (FDR) + fine (sync_and_demodulate) search path.
 *
 * This is the drop-in boundary: plain pointers and sizes, no C++/torch types.
 * A GNU Radio block shim (gr-uwspr_amd/host/) or any FFI (ctypes, cgo, JNI)
 * binds exactly these symbols.  Each entry point cites the reference interface
 * it replaces (paths relative to the upstream gr-uwspr tree).
 *
 * Conventions
 *   - every function returns 0 (UWSPR_OK) or a negative uwspr_status; nothing
 *     here calls exit() or throws (the reference does: lib/FDR_impl.cc:85-90).
 *   - `where` says whether the bulk pointers of that call are host
 *     (UWSPR_HOST) or device (UWSPR_DEVICE) memory; a call never mixes them.
 *   - frames are interleaved (I,Q) binary32 pairs, `fl` pairs per frame,
 *     frame b at frames + 2*fl*b (or 2*stride*b, uwspr_set_frame_stride): the payload of the PDU that
 *     sliding_window_stream_to_pdu emits (lib/sliding_window_stream_to_pdu_impl.cc:113-135)
 *     narrowed from complex<double> to the gr_complex it was built from (cc:109).
 *   - a context is thread-compatible: one context per host thread; no globals.
 */
#ifndef UWSPR_HIP_H
#define UWSPR_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define UWSPR_ABI_VERSION 5   /* 2: frame stride, in-place stream views, uwspr_pipe_*, uwspr_dist_*, uwspr_host_threads;
                                 3: uwspr_set_option / uwspr_get_option, uwspr_pipe_set_option, uwspr_pipe_inject_failure,
                                    uwspr_pipe_opts.spare_after_us (was reserved);
                                 4: option "frontend" (the flowgraph's GNU Radio chain is the default front-end),
                                    uwspr_frontend_design replaces uwspr_frontend_taps;
                                 5: uwspr_host_set_ranks (the host's CPU share divided between the ranks of a node) */

typedef enum {
  UWSPR_OK = 0,
  UWSPR_ERR_PARAM = -1,       /* halfbandwidth > fs/2 (reference exit(-1)s: FDR_impl.cc:85-90) */
  UWSPR_ERR_RANGE = -2,       /* pass band + search reach leaves [0,size) (reference reads OOB) */
  UWSPR_ERR_UNSUPPORTED = -3, /* geometry this build does not implement (spb != 256, ...) */
  UWSPR_ERR_HIP = -4,         /* HIP runtime error, see uwspr_last_error() */
  UWSPR_ERR_NOMEM = -5,
  UWSPR_ERR_ARG = -6,         /* bad pointer / size / ordering in a call */
  UWSPR_ERR_NODEVICE = -7     /* no usable gfx950 device: there is NO CPU fallback */
} uwspr_status;

/* UWSPR_DEVICE_FRAMES (uwspr_fdr_batch, uwspr_demod_batch, uwspr_pipeline_batch, uwspr_demod_resume): the frames are device
 * memory (e.g. what uwspr_stream_take returned), every other pointer of the call is host memory. */
enum { UWSPR_HOST = 0, UWSPR_DEVICE = 1, UWSPR_DEVICE_FRAMES = 2,
       UWSPR_HOST_ASYNC = 3 /* uwspr_stream_push only: see there */ };
enum { UWSPR_LINEAR = 0, UWSPR_NONLINEAR = 1 };   /* enum Modes, lib/candidate_t.h:36 */

#define UWSPR_NSYM 162      /* symbols per frame */
#define UWSPR_NSLM 125      /* straight-line-model instances, lib/slm.cc:79-87 */
#define UWSPR_NK0 26        /* half-symbol start offsets, FDR_impl.cc:346 */
#define UWSPR_NIFR 5        /* tuned bins if0-2..if0+2, FDR_impl.cc:344 */
#define UWSPR_NJIG 17       /* jiggered shifts, sync_and_demodulate_impl.cc:460 */

/* Constructor arguments of the two blocks:
 * gr::uwspr::FDR::make(fs,fl,spb,maxdrift,maxfreqs,halfbandwidth,cf,threshold)
 *   include/uwspr/FDR.h:49-50
 * gr::uwspr::sync_and_demodulate::make(fs,fl,spb,maxdrift,maxfreqs,cf)
 *   include/uwspr/sync_and_demodulate.h:49  (a subset of the above) */
typedef struct uwspr_params {
  int32_t fs;             /* 375 */
  int32_t fl;             /* 45000 */
  int32_t spb;            /* 256 */
  int32_t maxdrift;       /* 0 */
  int32_t maxfreqs;       /* 200 */
  int32_t halfbandwidth;  /* 10 in the flowgraphs (the GRC default 187 is rejected: UWSPR_ERR_RANGE) */
  int32_t cf;             /* 1500 */
  int32_t threshold;      /* 10 */
} uwspr_params;

/* candidate_t, lib/candidate_t.h:27-50: same 48-byte layout, so a slab of
 * these can be handed to code compiled against the reference header. */
typedef struct uwspr_candidate {
  float freq;
  float snr;
  float drift;     /* unused by the reference as well */
  float sync;
  int32_t shift;
  int32_t m_type;  /* UWSPR_LINEAR / UWSPR_NONLINEAR */
  union {
    struct { float drift; } m_linear;
    struct { double V1, V2; int32_t p1, p2; } m_nonlinear;
  };
} uwspr_candidate;

/* One point of the fine (freq, time-lag, drift) grid: one pass of the
 * ifreq/lag loop body of sync_and_demodulate_impl.cc:163-232. */
typedef struct uwspr_hyp {
  int32_t frame;   /* index into the frames of the call; <0 = skip */
  int32_t m_type;
  float f0;        /* Hz, the f0 of cc:164 */
  int32_t lag;     /* samples, the lag of cc:165 */
  float drift;     /* linear: *drift1 of cc:173 */
  int32_t p1, p2;  /* nonlinear: candidate.m_nonlinear */
  int32_t _pad;
  double V1, V2;
} uwspr_hyp;

/* Result of the per-candidate refinement schedule
 * (sync_and_demodulate_impl.cc:403-482) up to, not including, Fano: every
 * one of the 17 jiggered mode-2 soft-symbol vectors is produced, in the order
 * the reference tries them, with the sync and rms it gates Fano on. */
typedef struct uwspr_demod_out {
  float f1;
  float drift1;
  float sync1;
  int32_t shift1;
  int32_t worth_a_try;
  float jig_sync[UWSPR_NJIG];
  float jig_rms[UWSPR_NJIG];
  int32_t jig_shift[UWSPR_NJIG];
  uint8_t symbols[UWSPR_NJIG][UWSPR_NSYM];
  uint8_t _pad[2];
} uwspr_demod_out;

/* Derived constants (FDR_impl.cc:81-141) and buffer geometry. */
typedef struct uwspr_info {
  int32_t abi_version;
  int32_t size, m, hpbm, n, finpb, noiseidx;
  float df, min_snr;
  int32_t band_lo, band_w;     /* spectrogram columns kept: [band_lo, band_lo+band_w) */
  int32_t cell_hyps;           /* (2*maxdrift+1) + 125 */
  int32_t off_min, off_max;    /* range of ifd-ifr over the whole search */
  int32_t device;
  char device_name[64];
} uwspr_info;

typedef struct uwspr_ctx uwspr_ctx;

/* ---- lifetime ---------------------------------------------------------- */
/* Replaces FDR_impl::FDR_impl / sync_and_demodulate_impl ctor set-up
 * (FDR_impl.cc:48-151, sync_and_demodulate_impl.cc:62-111). */
int uwspr_ctx_create(const uwspr_params *p, int device, uwspr_ctx **out);
void uwspr_ctx_destroy(uwspr_ctx *ctx);
const char *uwspr_last_error(const uwspr_ctx *ctx);
const char *uwspr_status_string(int status);
int uwspr_get_info(const uwspr_ctx *ctx, uwspr_info *info);
/* Run on a caller-owned hipStream_t (NULL = the context's own, non-blocking
 * stream).  All calls are asynchronous on that stream when where==UWSPR_DEVICE
 * and complete before returning when where==UWSPR_HOST.  Device inputs written
 * by work on ANOTHER stream must be complete (or ordered by the caller, e.g. by
 * passing that stream here) before the call: the library adds no cross-stream
 * dependency of its own. */
int uwspr_set_stream(uwspr_ctx *ctx, void *hip_stream);
int uwspr_synchronize(uwspr_ctx *ctx);

/* ---- front-end -------------------------------------- */
/* 12 kS/s real audio -> fl complex samples at 375 S/s.  In the reference this is the flowgraph's chain of GNU Radio
 * blocks (examples/WaveFilePlusNoiseDecode.grc): float_to_complex (:527) -> freq_xlating_fft_filter_ccc with
 * firdes.band_pass(1, 12000, 1490, 1510, 10, WIN_HAMMING) real taps at centre 0 (:303-352, 840-893) ->
 * freq_xlating_fft_filter_ccc with firdes.low_pass(1, 12000, 1510, 10, WIN_HAMMING) at centre 1500 Hz (:358-400,
 * 903-956) -> rational_resampler_ccc(1, 32) with its own Kaiser design (:1767-1808).
 *   option "frontend" = 0 (UWSPR_FRONTEND_GRC, default): that chain, its taps designed from GNU Radio 3.7's published
 *     formulas and folded into ONE 6831-tap complex polyphase decimator (every stage is linear, the mixer's period
 *     divides the decimation): y[m] = sum_k g[k] x[32 m - k];
 *   option "frontend" = 1 (UWSPR_FRONTEND_COMPACT): mix by -1500 Hz, 1025-tap Hamming low-pass (100 Hz), decimate:
 *     y[m] = sum_k g[k] x[32 m + 512 - k] -- a sixth of the arithmetic, not the reference's pass band.
 * GNU Radio is third-party, unpinned and absent here: **parity unpinned**; oracle/frontend_grc.py restates the chain
 * stage by stage in binary64.  audio [B][nin] (zero outside the record), frames_out [B][fl] (I,Q) pairs.
 * uwspr_frontend_design: stage 0 = the composite complex taps g (cap counts (re, im) pairs) and *delay = D of
 * y[m] = sum_k g[k] x[32 m + D - k]; stages 1, 2, 3 (grc mode) = the band-pass, low-pass and resampler designs
 * (real taps).  Returns the tap count (taps may be NULL), < 0 on a bad mode / stage. */
enum { UWSPR_FRONTEND_GRC = 0, UWSPR_FRONTEND_COMPACT = 1 };
int uwspr_frontend_batch(uwspr_ctx *ctx, const float *audio, int B, int nin, int where,
                         float *frames_out);
int uwspr_frontend_design(int mode, int stage, double *taps, int cap, int *delay);

/* ---- overlap-aware stream ingest ----------------------- */
/* sliding_window_stream_to_pdu::work (lib/sliding_window_stream_to_pdu_impl.cc:97-138) cuts the
 * 375 S/s stream into frames of fl samples that start every hop = shift*fs = 3375 samples: 41625 of
 * a frame's 45000 samples are its predecessor's.  Handing whole frames to the calls below uploads
 * every sample 13 times; through this interface every sample crosses PCIe ONCE, on a copy stream of
 * its own (uploads overlap the search kernels of the batches before), and the frames are read where
 * they lie:
 *   uwspr_stream_open(ctx, hop, max_frames)        hop in samples; at most max_frames per take
 *   uwspr_stream_push(ctx, iq, n, where, &nready)  append n (I,Q) pairs; nready = complete frames waiting.
 *       where = UWSPR_HOST: on return iq may be reused (pageable memory is staged; a page-locked buffer
 *       is read by the DMA itself and the call waits for that transfer -- not for any kernel);
 *       UWSPR_HOST_ASYNC: page-locked iq is only enqueued and must stay unmodified until
 *       uwspr_stream_wait_uploads returns; UWSPR_DEVICE: iq is device memory written by work on the
 *       context's stream.
 *   uwspr_stream_take_view(ctx, k, &frames, &stride, &pos)   the next k frames IN PLACE: frame j starts
 *       at frames + 2*stride*j floats (stride = hop; the frames overlap in memory as they do in the
 *       stream), valid until the next take; pos = stream index of the first one's first sample;
 *       consumes k*hop samples.  Use with uwspr_set_frame_stride(ctx, stride) and where = UWSPR_DEVICE
 *       or UWSPR_DEVICE_FRAMES; the kernels reading them must be enqueued before the next take.
 *   uwspr_stream_take(ctx, k, dev_dst, &frames, &pos)        the same frames COPIED out as contiguous
 *       [k][fl] pairs (dev_dst, or NULL = a buffer of the context, valid until the next take) for
 *       consumers that keep frames beyond the next take (the block mirror's FDR -> sync hand-over).
 *   uwspr_stream_reset(ctx, pos)                   drop what is buffered; the next sample pushed has index pos
 * Frame k of the stream is bit for bit the frame the reference's PDU k carries.
 *
 * uwspr_set_frame_stride(ctx, s): the calls that follow read frame b of their `frames` argument at
 * frames + 2*s*b floats instead of 2*fl*b (s = 0 restores fl).  With where = UWSPR_HOST the span
 * (B-1)*s + fl is uploaded once, so a host buffer holding a stretch of the stream (s = hop) is also
 * ingested without repeats.  Applies to every entry point that takes frames. */
int uwspr_set_frame_stride(uwspr_ctx *ctx, int stride_samples);
int uwspr_stream_open(uwspr_ctx *ctx, int hop, int max_frames);
int uwspr_stream_push(uwspr_ctx *ctx, const float *iq, int nsamples, int where, int *nready);
int uwspr_stream_wait_uploads(uwspr_ctx *ctx);
int uwspr_stream_take_view(uwspr_ctx *ctx, int nframes, const float **frames, int *stride, long long *first_pos);
int uwspr_stream_take(uwspr_ctx *ctx, int nframes, float *dev_dst, const float **frames, long long *first_pos);
int uwspr_stream_reset(uwspr_ctx *ctx, long long pos);
/* device memory for callers without a HIP runtime of their own (the block mirror's frame hand-over) */
int uwspr_device_alloc(size_t bytes, void **ptr);
void uwspr_device_free(void *ptr);
/* Page-locked host memory for frames handed over as host pointers: the upload is then one DMA at the
 * link's rate instead of the runtime's staged copy of pageable memory (any page-locked buffer is
 * recognised, whoever allocated it). */
int uwspr_host_alloc(size_t bytes, void **ptr);
void uwspr_host_free(void *ptr);

/* ---- coarse search: FDR_impl::transform, FDR_impl.cc:214-456 ------------ */
/* cands: [B][maxfreqs] records; npk: [B].  Same candidate order, fields and
 * selection rule (running-best, FDR_impl.cc:360,392) as the reference. */
int uwspr_fdr_batch(uwspr_ctx *ctx, const float *frames, int B, int where,
                    uwspr_candidate *cands, int32_t *npk);

/* Inspection of the intermediates of the LAST uwspr_fdr_batch call (host
 * pointers; any may be NULL):
 *   ps_band [B][n][band_w]   FDR_impl.cc:246-253 (columns band_lo..)
 *   psavg   [B][band_w]      cc:257-263
 *   smraw   [B][finpb]       cc:268-275
 *   smspec  [B][finpb]       cc:287-291
 *   noise   [B]              cc:285 */
int uwspr_fdr_read_spectrum(uwspr_ctx *ctx, int B, float *ps_band, float *psavg,
                            float *smraw, float *smspec, float *noise);
/* When enabled, the next uwspr_fdr_batch also keeps every hypothesis metric:
 * grid [B][ncand_cap][5][26][cell_hyps] (cc:357,390), read back with
 * uwspr_fdr_read_syncgrid.  Off by default: it is 65 KB per candidate, and the coarse search then has to
 * evaluate every hypothesis instead of only those that can still be accepted (same candidates either way). */
int uwspr_fdr_keep_syncgrid(uwspr_ctx *ctx, int ncand_cap);
int uwspr_fdr_read_syncgrid(uwspr_ctx *ctx, int B, float *grid);

/* ---- fine sweep: core of sync_and_demodulate, cc:126-256 ---------------- */
/* One metric per (freq, lag, drift) hypothesis, in any order (grouping them by
 * frame keeps a frame's samples in L2).  sync: [H]; symbols: [H][162] soft
 * symbols (cc:240-254) or NULL to skip them (modes 0/1). */
int uwspr_sync_sweep(uwspr_ctx *ctx, const float *frames, int B,
                     const uwspr_hyp *hyps, int H, int where, float *sync,
                     uint8_t *symbols);

/* The (freq, drift, lag) grid sweep of BASELINE configs[2].  Every frame b has a
 * centre (a candidate record: freq, shift, m_type and its drift / SLM parameters)
 * and the same offsets are applied around every centre:
 *     f0    = centre.freq  + df[i]            (binary32 add, as cc:164 forms f0)
 *     drift = centre.m_linear.drift + ddrift[j]   (binary32 add; linear centres only)
 *     lag   = centre.shift + dlag[k]
 * One grid point = one hypothesis of uwspr_sync_sweep (same arithmetic, bit-identical
 * results); the grid form lets the kernel keep each symbol window on chip for all
 * of a frame's points and share the tone phasors between points that differ only in
 * lag.  df/ddrift/dlag are host arrays (nf, ndrift <= 32); centres [B], sync
 * [B][nf][ndrift][nlag], symbols [B][nf][ndrift][nlag][162] (or NULL) per `where`. */
int uwspr_sync_grid(uwspr_ctx *ctx, const float *frames, int B, int where,
                    const uwspr_candidate *centres, int nf, const float *df, int ndrift,
                    const float *ddrift, int nlag, const int32_t *dlag, float *sync,
                    uint8_t *symbols);

/* Argument-for-argument batch form of sync_and_demodulate() (cc:126-131):
 * one call = one invocation of the reference function; results are what it
 * writes back through sync/shift1/f1 (and symbols in mode 2).  Host pointers
 * for the call records; frames per `where`. */
typedef struct uwspr_sync_call {
  int32_t frame;
  uwspr_candidate candidate;
  float f1;
  int32_t ifmin, ifmax;
  float fstep;
  int32_t shift1, lagmin, lagmax, lagstep;
  float drift1;
  int32_t symfac;
  int32_t mode;
} uwspr_sync_call;
typedef struct uwspr_sync_result {
  float sync;
  int32_t shift1;
  float f1;
  uint8_t symbols[UWSPR_NSYM];
  uint8_t _pad[2];
} uwspr_sync_result;
int uwspr_sync_and_demodulate_batch(uwspr_ctx *ctx, const float *frames, int B,
                                    int where, const uwspr_sync_call *calls,
                                    int ncalls, uwspr_sync_result *results);

/* ---- refinement schedule: sync_and_demodulate_impl::demodulate ---------- */
/* cands [B][cand_stride], npk [B] as produced by uwspr_fdr_batch; the first
 * min(npk[b], max_per_frame) candidates of each frame are refined.
 * out [B][max_per_frame]; entries past npk[b] are zeroed. */
int uwspr_demod_batch(uwspr_ctx *ctx, const float *frames, int B, int where,
                      const uwspr_candidate *cands, const int32_t *npk,
                      int cand_stride, int max_per_frame, uwspr_demod_out *out);

/* Lazy jiggered shifts.  The reference stops at the first of its up to 17 mode-2 tries that
 * decodes (sync_and_demodulate_impl.cc:457-490); by default this library produces all 17 soft-symbol
 * vectors before the host sees any.  uwspr_set_tries(ctx, k), k < 17: the schedule calls that follow
 * produce only tries idt < k (the other entries of uwspr_demod_out are zero, so uwspr_decode_candidate
 * skips them) and keep what is needed to produce the rest later.  Try 0 repeats the hypothesis that won
 * the last stage: with the fused schedule kernel k = 1 costs no correlation at all, the staged form
 * runs its last stage on the k wanted tries only.  Both forms are resumed by the fused kernel.
 * uwspr_demod_resume(ctx, frames, B, where, need, max_per_frame, out): for the slots b*max_per_frame+j
 * with need[...] != 0 of the LAST schedule call (same frames, B, max_per_frame) all 17 tries are
 * produced, byte-identical to what a k = 17 call gives; the other records are left alone.
 * where == UWSPR_HOST (or UWSPR_DEVICE_FRAMES): need is host memory and `out` receives all B*max_per_frame records;
 * UWSPR_DEVICE: need and out are device memory, out being the buffer the first call wrote. */
int uwspr_set_tries(uwspr_ctx *ctx, int ntries);

/* Options of a context.  The reference's blocks have no run-time switches (their constants are
 * hard-coded); these choose between forms of THIS implementation that give the same
 * bytes -- except "fast_search":
 *   "sched"          1 (default): the whole S0..S5 schedule of a candidate in one workgroup; 0: one launch
 *                    per stage (what uwspr_pipe_* uses from three lanes up)
 *   "stage_kernels"  staged form: 1 (default) packed / ring / pair kernels, 0 the flat kernel for
 *                    every stage -- an independent form the equivalence tests compare the others with
 *   "reuse"          1 (default): the hypothesis that repeats the previous stage's winner
 *                    (sync_and_demodulate_impl.cc:416-452 evaluate it again and get the same number) is skipped
 *   "phasor_tables"  1 (default): drift-free stages read the tone phasors of cc:186-199 from per-slot tables
 *   "fast_search"    0 (default).  1: stages S0..S4 with fused multiply-adds and wavefront shuffle-tree sums --
 *                    NOT the reference's arithmetic: sync metrics agree to ~1e-6 relative (1e-5 is BASELINE's
 *                    tolerance), integer results and soft symbols were identical on every frame tried
 *                    (tests/test_gpu_fast_search.py); stage 5 -- the soft symbols -- and every other entry
 *                    point stay exact.  Staged form only (it sets "sched" 0).
 * Further names ("k4_t", "k5_lanes", "sched_stamps", "sched_grid", "dist_force_comm", and the
 * creation-time "k1_rows", "k3_tile", "k3_pitch") are diagnostics: DESIGN.md section 9.  The one environment
 * variable the library reads, UWSPR_OPTIONS="name=value,...", presets options for every context the process
 * creates.  Unknown names: UWSPR_ERR_ARG. */
int uwspr_set_option(uwspr_ctx *ctx, const char *name, int value);
int uwspr_get_option(uwspr_ctx *ctx, const char *name, int *value);

int uwspr_demod_resume(uwspr_ctx *ctx, const float *frames, int B, int where, const uint8_t *need,
                       int max_per_frame, uwspr_demod_out *out);

/* FDR followed by the schedule with candidates kept in HBM in between (the
 * FDR -> sync_and_demodulate PDU hop of the flowgraph, examples/
 * WaveFilePlusNoiseDecode.grc).  Any output pointer may be NULL. */
int uwspr_pipeline_batch(uwspr_ctx *ctx, const float *frames, int B, int where,
                         int max_per_frame, uwspr_candidate *cands,
                         int32_t *npk, uwspr_demod_out *out);

/* Multi-GPU hand-off: packs the results of the LAST uwspr_pipeline_batch into
 * one fixed-size slab per frame, ready for a gather to the root rank:
 *   int32 npk, 12 pad bytes | candidate_t[K] (first K candidates, zero-filled past npk)
 *   | float f1, drift1, sync1; int32 shift1 of the frame's top candidate
 * = 32 + 48*K bytes per frame; slabs: [B][32+48K]. */
int uwspr_pack_slabs(uwspr_ctx *ctx, int B, int K, void *slabs, int where);
/* The same slabs without a launch of their own: the NEXT uwspr_pipeline_batch (one shot) also writes its frames' slabs
 * to slabs_device ([B][32+48K] bytes in device memory) -- from the schedule's last kernel where the schedule form has
 * one, else with the packing kernel behind it.  slabs_device = NULL cancels. */
int uwspr_pipeline_slabs(uwspr_ctx *ctx, int K, void *slabs_device);

/* ---- multi-GPU: the final gather over RCCL --------------------------------------- */
/* One process per GPU, frames sharded round-robin (global frame b on rank b mod G), no data-path
 * collective; the one exchange is the gather of the uwspr_pack_slabs output to the root rank:
 * point-to-point RCCL transfers (ncclSend / ncclRecv in one group), each peer -> root over its own xGMI
 * link.  librccl is loaded on first use; world == 1 needs no RCCL at all (the gather is a copy).
 *   uwspr_dist_unique_id(id)             rank 0: 128 bytes to hand to every rank out of band (ncclGetUniqueId)
 *   uwspr_dist_init(ctx, rank, world, id)  collective: every rank calls it with the same id
 *   uwspr_dist_gather(ctx, send, bytes, recv, root, UWSPR_DEVICE)
 *        every rank contributes `bytes` bytes of device memory (equal on all ranks: pad the shards);
 *        the root receives world * bytes in rank order (recv may be NULL elsewhere).  Asynchronous on
 *        the context's stream, after whatever produced `send` there.
 *   uwspr_dist_finalize(ctx)             (also done by uwspr_ctx_destroy) */
int uwspr_dist_unique_id(void *id128);
int uwspr_dist_init(uwspr_ctx *ctx, int rank, int world, const void *id128);
int uwspr_dist_gather(uwspr_ctx *ctx, const void *send, size_t bytes, void *recv, int root, int where);
int uwspr_dist_finalize(uwspr_ctx *ctx);

/* ---- pipelined end-to-end decoder ---------------------------------------------------------- */
/* The whole receive chain of examples/WaveFilePlusNoiseDecode.grc behind one object:
 *   sliding_window_stream_to_pdu (lib/sliding_window_stream_to_pdu_impl.cc:97-138)  -> frames in place
 *   FDR::transform               (lib/FDR_impl.cc:214-456)                          -> candidates
 *   sync_and_demodulate::demodulate (lib/sync_and_demodulate_impl.cc:315-534)       -> 7-byte messages
 * with the stages of CONSECUTIVE batches overlapped: page-locked samples go up on a copy stream while
 * the batches before are searched; the schedule produces only the first jiggered shift
 * (uwspr_set_tries(1): the reference stops at its first decoding try, cc:457-490); Fano runs on a
 * persistent pool of host threads under the kernels of the next batch; the candidates whose first
 * try did not decode are resumed on the GPU (uwspr_demod_resume) and tried again.  Results are
 * byte-for-byte those of the sequential calls (uwspr_pipeline_batch with all 17 tries +
 * uwspr_decode_batch), in frame order.  One producer thread drives a pipe. */
typedef struct uwspr_pipe uwspr_pipe;
typedef struct uwspr_pipe_opts {
  int32_t hop;            /* samples between frame starts (shift*fs = 3375); pushed streams only */
  int32_t batch_frames;   /* frames per GPU batch (256) */
  int32_t max_per_frame;  /* candidates refined per frame (cc:389 refines all npk; 1 = the strongest) */
  int32_t lanes;          /* most batches in flight, each with its own context (0: 9).  The first three have a HIP stream
                             each and take the batches in turn; the others share those streams and are spares, opened
                             only while every one of the three is busy and a host tail (Fano time-outs) has lasted > 2.5 ms */
  int32_t host_threads;   /* Fano threads (0: uwspr_host_threads() - 2, leaving the producer and the HIP runtime a core each) */
  int32_t eager;          /* 1: all 17 tries in the first pass, no resume (A/B against the lazy flow) */
  int32_t sched_form;     /* 0: staged launches from 3 lanes up, the fused kernel below (a "sched" entry of UWSPR_OPTIONS overrides), 1: fused, 2: staged */
  int32_t spare_after_us; /* a spare lane opens when every base lane is busy and a host tail has lasted this long (0: 2500) */
} uwspr_pipe_opts;
/* one refined candidate (j < min(npk, max_per_frame)) of one frame */
typedef struct uwspr_decode {
  int64_t frame;          /* running index of the frame in submission order */
  int64_t stream_pos;     /* index of its first sample in the pushed stream (-1: submitted frames) */
  int32_t cand, npk;      /* rank of this candidate, candidates found in the frame */
  uwspr_candidate coarse; /* the FDR record (candidate_t) */
  float f1, drift1, sync1;
  int32_t shift1, worth_a_try;
  int32_t decoded;        /* 1: message holds the 7 bytes sync_and_demodulate publishes (cc:528-530) */
  int32_t idt;            /* the jiggered try that decoded (-1: none) */
  int8_t message[7];
  uint8_t _pad[5];
} uwspr_decode;
typedef struct uwspr_pipe_stats {
  int64_t frames, batches, candidates, decoded, resumed;   /* resumed: records whose other tries were produced */
  int64_t fano_calls, fano_timeouts;                       /* tries that passed the gates (cc:470) / ran to the cycle limit */
  double gpu_wait_s, fano_s, resume_s;                     /* coordinator thread: where its time went */
} uwspr_pipe_stats;
int uwspr_pipe_open(const uwspr_params *p, int device, const uwspr_pipe_opts *o, uwspr_pipe **out);
void uwspr_pipe_close(uwspr_pipe *pipe);
const char *uwspr_pipe_last_error(const uwspr_pipe *pipe);
/* A page-locked buffer for the next nsamples (I,Q) pairs of the stream (at most batch_frames*hop): fill it,
 * then commit.  Blocks only while every staging buffer still has its upload in flight. */
int uwspr_pipe_acquire(uwspr_pipe *pipe, int nsamples, float **iq);
int uwspr_pipe_commit(uwspr_pipe *pipe, int nsamples);
/* acquire + memcpy + commit for samples that live elsewhere */
int uwspr_pipe_push(uwspr_pipe *pipe, const float *iq, int nsamples);
/* B frames already in device memory (frame b at frames + 2*stride*b floats, stride 0 = fl): searched as one
 * batch.  The memory must stay valid until the batch's results have been collected. */
int uwspr_pipe_submit_device(uwspr_pipe *pipe, const float *dev_frames, int B, int stride);
/* launch what is complete (a last, short batch of a pushed stream) and wait for everything in flight */
int uwspr_pipe_flush(uwspr_pipe *pipe);
/* up to cap finished records in frame order; wait != 0: block until at least one is there or nothing is
 * in flight.  returns the count (>= 0) or a negative status. */
int uwspr_pipe_collect(uwspr_pipe *pipe, uwspr_decode *out, int cap, int wait);
int uwspr_pipe_get_stats(uwspr_pipe *pipe, uwspr_pipe_stats *st);
/* uwspr_set_option on every lane's context (e.g. "fast_search").  Only while nothing is in flight (after
 * uwspr_pipe_flush / before the first batch): UWSPR_ERR_ARG otherwise, and for "sched", which the pipe chooses by its
 * lane count (uwspr_pipe_opts.sched_form). */
int uwspr_pipe_set_option(uwspr_pipe *pipe, const char *name, int value);
/* Error behaviour.  An argument error (too many samples, a bad B or stride) fails THAT call with UWSPR_ERR_ARG and
 * its message; the pipe goes on.  A runtime failure (HIP, a lane's context) is sticky: the batch it hit emits
 * nothing, records of the batches before it stay collectable, every later call -- and uwspr_pipe_collect once
 * those records are gone -- returns the status; uwspr_pipe_close always returns.
 * Test hook: the batch with running number `batch` fails at its launch (where = 0) or in its host tail (where = 1). */
int uwspr_pipe_inject_failure(uwspr_pipe *pipe, long long batch, int where);

/* ---- measurement -------------------------------------------------------- */
enum { UWSPR_K_SPECTROGRAM = 0, UWSPR_K_SPECTRUM = 1, UWSPR_K_COARSE = 2,
       UWSPR_K_TONECORR = 3, UWSPR_K_FOLD = 4, UWSPR_K_SCHED = 5, UWSPR_K_COUNT = 6 };
typedef struct uwspr_prof {
  double ms[UWSPR_K_COUNT];        /* summed HIP-event time per kernel family */
  int64_t launches[UWSPR_K_COUNT];
  int64_t units[UWSPR_K_COUNT];    /* frames (K0,K1), candidates (K2), hypotheses (K3,K4) */
} uwspr_prof;
/* HIP events are recorded around the kernel launches of the selected families on
 * the context's stream; uwspr_prof_read synchronises, sums and resets them.
 * mask: bit (1 << UWSPR_K_*) per family, UWSPR_PROF_ALL for all, 0 = off.
 * (Each recorded event is a marker packet on the queue: measuring only the
 * dominant kernel keeps the timed region close to the unobserved one.) */
#define UWSPR_PROF_ALL 0x3f
int uwspr_prof_enable(uwspr_ctx *ctx, int mask);
int uwspr_prof_read(uwspr_ctx *ctx, uwspr_prof *out);
/* Start/stop times (ms after `epoch_event`, a hipEvent_t recorded by the caller
 * on the same device) of up to `cap` recorded launches of one family, WITHOUT
 * resetting them; *n receives the count.  Lets a caller that rotates batches over
 * several contexts/streams account for launches that overlap in time. */
int uwspr_prof_intervals(uwspr_ctx *ctx, int kind, void *epoch_event, double *start_ms,
                         double *stop_ms, int cap, int *n);

/* ---- host-side tail of the path ------------------ */
/* sync_and_demodulate_impl.cc:265-282 */
void uwspr_deinterleave(uint8_t *symbols162);
/* lib/Fano.cc:110-252 with the block's metric table (Fano.cc:36-45).
 * returns 0 decoded, -1 timeout (same as the reference). data: 11 bytes. */
int uwspr_fano_decode(const uint8_t *symbols162, uint8_t *data11,
                      uint32_t *metric, uint32_t *cycles, uint32_t *maxnp,
                      int delta, uint32_t maxcycles);
/* lib/Fano.cc:81-100 */
int uwspr_fano_encode(uint8_t *symbols, const uint8_t *data, uint32_t nbytes);
/* Replays cc:457-490 for one candidate on a uwspr_demod_out: gates, deinterleave,
 * Fano in reference order.  returns 1 and fills message7 on decode, else 0. */
int uwspr_decode_candidate(const uwspr_demod_out *d, int8_t *message7,
                           int32_t *idt_used);
/* CPUs this process may keep busy: hardware threads capped by the affinity mask and the cgroup CPU
 * quota (what "one per hardware thread" means below). */
int uwspr_host_threads(void);
/* One process per GPU: the `ranks` processes of a job that share this host (the launcher's LOCAL_WORLD_SIZE) share its
 * CPUs too.  After uwspr_host_set_ranks(n), uwspr_host_threads() is the share above divided by n (at least 1), and that
 * is what the process-wide host pool -- uwspr_decode_batch, the Fano stage of uwspr_pipe_* -- is sized by: eight ranks
 * decode on eight eighths of the host, not on eight full pools (BASELINE configs[4] is host-bound: Fano time-outs at low
 * SNR).  Call it before the first uwspr_decode_batch / uwspr_pipe_open of the process: once the pool exists its size is
 * fixed and the call returns UWSPR_ERR_UNSUPPORTED (nothing changes).  ranks < 1: UWSPR_ERR_ARG. */
int uwspr_host_set_ranks(int ranks);
/* The same for n records on `nthreads` host threads (<= 0: one per hardware
 * thread): the per-candidate loop of cc:389 with its Fano calls spread over the
 * host cores.  messages [n][7] (zero when not decoded), idt_used [n] or NULL,
 * decoded [n] = 0/1.  returns the number decoded (or UWSPR_ERR_ARG). */
int uwspr_decode_batch(const uwspr_demod_out *d, int n, int nthreads,
                       int8_t *messages, int32_t *idt_used, uint8_t *decoded);
/* lib/helpers.cc unpk_ (called at WSPR_unpacker_impl.cc:129), without the
 * on-disk hash table: "CALL GRID dBm" into out (>= 23 bytes). returns 0 ok. */
int uwspr_unpack_message(const int8_t *message7, char *out, size_t out_len);
/* .c2 reader, lib/c2file_source_impl.cc:80-96 (Q negated on load).
 * iq: 2*45000 floats. */
int uwspr_c2_read(const char *path, float *iq, double *dial_freq, int32_t *type);

#ifdef __cplusplus
}
#endif
#endif /* UWSPR_HIP_H */
