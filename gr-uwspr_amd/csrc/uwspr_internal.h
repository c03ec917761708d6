// uwspr_internal.h -- shared declarations of the HIP implementation
// (context, device-side records, kernel launchers).  Not part of the ABI.
#pragma once

#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <stdint.h>
#include <vector>

#include "../../include/uwspr_hip.h"
#include "stream_ring.h"

// Options of a context, by index (names and defaults: uwspr_api.hip: kOptions; ABI: uwspr_set_option).
enum {
  UWSPR_OPT_SCHED = 0,          // 1 fused kernel (default), 0 staged launches
  UWSPR_OPT_STAGE_KERNELS,      // staged form: 1 packed / ring kernels (default), 0 flat kernel everywhere
  UWSPR_OPT_REUSE,              // 1 (default): the hypothesis that repeats the previous stage's winner is not correlated again
  UWSPR_OPT_PHASOR_TABLES,      // 1 (default)
  UWSPR_OPT_FAST_SEARCH,        // 0 (default)
  UWSPR_OPT_K4_T,               // flat kernel: tones per lane (0 = by size)
  UWSPR_OPT_K5_LANES,           // fold form: -1 by size (default), 0 wave form, 1 lanes form
  UWSPR_OPT_K1_ROWS,            // spectrogram rows per wavefront walk (0 = default)
  UWSPR_OPT_K3_TILE,            // coarse-search tile form (-1 = by size)
  UWSPR_OPT_K3_PITCH,           // coarse-search tile row pitch (0 = default)
  UWSPR_OPT_SCHED_STAMPS,       // diagnostics: phase times of the fused kernel
  UWSPR_OPT_SCHED_GRID,         // fused kernel: workgroups (0 = one per CU)
  UWSPR_OPT_DIST_FORCE_COMM,    // tests: a one-rank communicator is really created
  UWSPR_OPT_FRONTEND,           // K0 taps: 0 the flowgraph's three-stage GNU Radio chain (default), 1 the compact single stage
  UWSPR_OPT_K4_FORMS,           // staged form, bit mask of kernel forms (default: all that won their A/B): bit 0 = S5 register ring
  UWSPR_NOPT
};

namespace uwspr {

// 162 WSPR sync bits (lib/pr3.h:5-13), LSB-first packed; a protocol constant.
#define UWSPR_PR3_WORDS                                                        \
  { 0x07a47103u, 0x58b340a4u, 0x56349558u, 0xe2cdc904u, 0x63580ca0u, 0x00000000u }
__host__ __device__ constexpr uint32_t pr3_word(int w) {
  constexpr uint32_t t[6] = UWSPR_PR3_WORDS;
  return t[w];
}
__host__ __device__ constexpr int pr3_bit(int k) { return (pr3_word(k >> 5) >> (k & 31)) & 1; }

// IEEE-754 correctly rounded binary32 sqrt and divide.  NOTE: HIP's __fsqrt_rn /
// __fdiv_rn are NOT that (they lower to the 1-ulp native forms unless
// OCML_BASIC_ROUNDED_OPERATIONS is defined); plain sqrtf() and `/` are, under
// -fhip-fp32-correctly-rounded-divide-sqrt (set explicitly in the build flags).
__device__ __forceinline__ float ieee_sqrtf(float x) { return __builtin_sqrtf(x); }
__device__ __forceinline__ float ieee_divf(float x, float y) { return x / y; }

// Workgroups are dealt to the 8 XCDs round-robin by blockIdx, each XCD with a private
// 4 MB L2.  This bijective remap hands every XCD a CONTIGUOUS range of logical blocks,
// so the blocks that re-read the same symbol windows (the hypotheses of one candidate)
// share one L2 instead of each missing in its own.  Pure placement: results unchanged.
__device__ __forceinline__ unsigned xcd_swizzle(unsigned bid, unsigned nwg) {
#ifdef UWSPR_NO_XCD_SWIZZLE
  return bid;
#else
  const unsigned q = nwg >> 3, r = nwg & 7u, x = bid & 7u;
  return (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + (bid >> 3);
#endif
}

// Instruction-issue priority of the short kernels' wavefronts (s_setprio, 0..3).  Under several streams the kernels
// of different lanes share the SIMDs and the arbiter serves higher priority first, then age: a fold or a schedule
// transition that takes 6 us alone took 40-65 us beside the other lanes' correlation launches, and every lane's chain
// waits on five of them per step.  With priority 3 in the K5 family (folds, schedule init / transitions / finish,
// slab packing) `value` rises 2 % (profiles/r03_ab_experiments.txt); raising K1..K3 or the S0..S4 correlation
// launches above the jiggered-shift ring as well adds nothing.  -DUWSPR_SMALL_PRIO=0 restores the default priority.
#ifndef UWSPR_SMALL_PRIO
#define UWSPR_SMALL_PRIO 3
#endif
#define UWSPR_SET_PRIO(level) do { if ((level) > 0) __builtin_amdgcn_s_setprio(level); } while (0)

// fine-grid hypothesis as the kernels consume it (24 B)
struct dev_hyp {
  int32_t frame;   // <0: skip
  int32_t lag;
  float f0;
  float drift;     // linear: *drift1
  float slmc;      // nonlinear: slmFrequencyDrift(m_nl, cf, t=0)
  int32_t m_type;
};

// A lag group: up to 8 hypotheses that differ only in their time lag (same
// frame, frequency, drift model).  They share the per-symbol tone phasors, so
// K4 advances the phasor recurrence once and correlates all lags against it.
struct dev_grp {
  int32_t frame;     // <0: skip
  int32_t m_type;
  float f0, drift, slmc;
  int32_t nvalid;    // lags in use (the kernel is instantiated for NL >= nvalid)
  int32_t hyp_base;  // lag slot l is hypothesis hyp_base + ((hmap >> 4l) & 15)
  int32_t lag[8];
  uint32_t hmap;     // 0x76543210 when the slots are in hypothesis order
};
static_assert(sizeof(dev_grp) == 64, "dev_grp is one 64-byte record");

// per-candidate refinement state kept in HBM between schedule stages
struct cand_state {
  int32_t frame;       // <0: empty slot
  int32_t m_type;
  float slmc;
  float f1, drift1, sync1;
  int32_t shift1;
  int32_t worth;
  float driftp, driftm;
  // The middle hypothesis of S1, S3 and S4 is the hypothesis that won the stage before it
  // (same f0, lag, drift): its metric is already known and K4 / the fold skip it.
  float csync;         // metric of (f1, shift1, drift1) when cknown
  int32_t cknown;
  // staged form: phasor tables of the lag stages (k5: ptab_build): the centre frequency each set was built
  // around and whether it was built (the frequency did not depend on the symbol at that point)
  float tabA_f, tabB_f;
  int32_t tabA_ok, tabB_ok;
};
// phasor tables of the staged schedule, per slot: tables 0..4 = set A (candidate frequency + {-2..2} 0.25 Hz: S0 uses
// the middle one, S1 all five), tables 5..9 = set B (f1 + {-2..2} 0.05 Hz after S2: S3 the middle one, S4 all five,
// S5 the stage-4 winner's); each [4 tones][256 steps] (c, s) -- the sequence of cc:186-199
constexpr int kPtabPerSlot = 10;
constexpr int kPtabSetB = 5;           // first table of set B
constexpr int kPtabFloat2 = 4 * 256;   // float2 per table

struct fdr_consts {
  int fl, n, size, m, hpbm, finpb, noiseidx, maxfreqs, maxdrift;
  int band_lo, band_w;
  int cell_hyps, nlin, ntot;      // hyps per (ifr,k0) cell; linear ones; 130*cell_hyps
  int off_min, off_max, nc;       // ifd-ifr range; tile centres per row = 5 + off_max-off_min
  int tp;                         // K3 tile row pitch, in float4 (tile form 0) or floats (forms 1, 2)
  int k3_mode;                    // K3 tile form (k3_coarse.hip: K3_TILE_*)
  int ifr_lo, n_ifr;              // rows of the offset table
  int umax;                       // distinct offset sequences per cell (max over rows)
  int cand_slots;                 // max candidates a frame can yield
  float df, min_snr, min_snr_floor, threshold;
};

struct ev_pair { hipEvent_t a, b; int kind; int64_t units; };

}  // namespace uwspr

struct uwspr_ctx {
  uwspr_params p;
  uwspr::fdr_consts fc;
  int device;
  int num_cus;
  char device_name[64];
  hipStream_t own_stream, stream;
  char err[512];

  // constant tables in HBM
  float *d_window;     // [512]
  float *d_twiddle;    // [256][2]
  float *d_k3_tile;    // [num_cus][n][tp] sqrt rows when the coarse tile does not fit LDS (K3_TILE_F1_HBM), else null
  uint32_t *d_off;     // [n_ifr][umax][84]: distinct offset sequences, 2 x u16 tile byte offsets per word (k3_coarse.hip)
  uint16_t *d_umap;    // [n_ifr][cell_hyps]: hypothesis -> distinct sequence
  float *d_fe_taps;    // [32][fe_J][2] complex front-end taps by phase (K0), built on first use for mode fe_mode
  int fe_mode, fe_J, fe_dcols;
  size_t cap_audio; float *d_audio;   // staging when the audio is host memory

  // batch scratch (grown on demand, never shrunk)
  size_t cap_frames_bytes; float *d_frames;       // staging when frames are host memory
  int cap_B;
  float *d_ps;          // [B][n][band_w]
  float *d_psavg;       // [B][band_w]
  float *d_smraw;       // [B][finpb]
  float *d_smspec;      // [B][finpb]
  float *d_noise;       // [B]
  uwspr_candidate *d_cands;  // [B][maxfreqs]
  int32_t *d_npk;       // [B]
  int32_t *d_work;      // [1 + B*cand_slots]: count, then frame*cand_slots + j items
  int last_B;
  int grid_cap; size_t cap_grid_bytes; float *d_syncgrid;  // [B][grid_cap][ntot]

  size_t cap_hyps; uwspr::dev_hyp *d_hyps;
  size_t cap_grps; uwspr::dev_grp *d_grps;
  size_t cap_cent; uwspr_candidate *d_cent; int32_t *d_cent_frame;   // per-slot grid centres (S1/S2/S4)
  size_t cap_abi_hyps; uwspr_hyp *d_abi_hyps;
  size_t cap_p; float4 *d_p;                      // [H][162] tone magnitudes
  size_t cap_sync; float *d_sync;                 // [H]
  size_t cap_sym; uint8_t *d_sym;                 // [H][162]
  size_t cap_state; uwspr::cand_state *d_state;
  size_t cap_dout; uwspr_demod_out *d_dout;
  // buffers the current call writes (the context's own, or the caller's device memory)
  uwspr_candidate *cur_cands; int32_t *cur_npk; uwspr_demod_out *cur_dout;
  int last_per_frame;
  // Options (uwspr_set_option; defaults in uwspr_api.hip: kOptions).  The ones the launch sequences branch on:
  int opt[UWSPR_NOPT];
  bool opt_set[UWSPR_NOPT];   // set by the caller (environment or uwspr_set_option), not the default
  bool use_fused;        // "sched" = 1: one workgroup per candidate runs S0..S5 (k6_sched); 0: staged launches
  bool use_stage_kernels;   // "stage_kernels" >= 1: the staged form's packed / ring / rows kernels; 0: the flat kernel for every stage
  bool reuse_centre;     // "reuse": skip the stage-winner hypothesis in S1/S3/S4 and try 0 of S5 (0: recompute it)
  bool use_ptab;         // "phasor_tables": the lag stages read their phasors from per-slot tables (0: every lane runs the recurrence)
  // "fast_search" = 1: stages S0..S4 of the schedule with fused multiply-adds and shuffle-tree sums (not
  // the reference's arithmetic; S5 and every other entry point stay exact).  fast_now: set around those launches.
  bool fast_search, fast_now;
  bool cands_from_fdr;   // the schedule call's candidates are this context's own FDR output (drift within +-maxdrift)
  uint8_t *next_slab = nullptr; int next_slab_K = 0; bool next_slab_done = false;   // uwspr_pipeline_slabs (one shot)
  int sched_per_frame = 1;   // candidate slots per frame of the schedule being launched
  // fused schedule (k6_sched)
  int sched_grid;
  size_t cap_tabs; float *d_tabs;     // [sched_grid][2][5][4][256](c, s) phasor tables
  int *d_counter;                     // candidate queue head of the running launch
  size_t cap_tmpc; uwspr_candidate *d_tmpc; size_t cap_tmpn; int32_t *d_tmpn;   // uwspr_demod_batch: host records staged
  // pinned staging for host -> device copies (two halves, ping-pong)
  void *h_pin; hipEvent_t pin_ev[2]; bool pin_busy[2];
  // overlap-aware stream ingest (uwspr_stream_*): the stream tail lives on the device (stream_ring.h)
  uwspr::stream_ring ring; float *d_stream_frames; size_t cap_stream_frames;
  hipEvent_t ring_ev;                 // recorded on `stream` at every take: the readers of earlier views are behind it
  // Frame pitch of the current calls (uwspr_set_frame_stride; default fl = contiguous frames) and the
  // sample bound of the fine search: npoints = 45000 whatever fl is (sync_and_demodulate_impl.cc:92,
  // passed at cc:413-465), capped at fl where the reference would read past its arrays.
  int fstride, np;
  int ntries;                         // mode-2 tries per candidate a schedule call produces (uwspr_set_tries)
  size_t cap_ptab; float2 *d_ptab;    // [nslots][kPtabPerSlot][4][256] phasor tables of the lag stages (staged form)
  size_t cap_pwin; float *d_pwin;     // [nslots][162][4] magnitudes of the stage winner (f1, shift1, drift1): try 0 of stage 5, and what uwspr_demod_resume starts from
  size_t cap_need; uint8_t *d_need;   // staging of the resume mask
  int last_slots, last_sched_B, last_sched_per_frame;
  bool last_sched_lazy; uwspr_demod_out *last_sched_out;   // what uwspr_demod_resume may continue (run_schedule)
  unsigned long long *d_sched_stamps; size_t cap_sched_stamps;   // option "sched_stamps": phase times of the last launch
  size_t cap_slab; uint8_t *d_slab;
  // multi-GPU gather (dist.hip): RCCL communicator of this rank, or null (single rank / not initialised)
  void *dist_comm; int dist_rank, dist_world;

  int prof_mask;
  std::vector<uwspr::ev_pair> prof_events;
  std::vector<hipEvent_t> ev_pool;
};

namespace uwspr {

// ---- launchers (each enqueues on ctx->stream) ------------------------------
int frontend_design(int mode, int stage, std::vector<double> &out, int *delay);
int frontend_tap_image(int mode, std::vector<float> &img, int *J, int *dcols);
int frontend_prepare();
void launch_frontend(uwspr_ctx *c, const float *audio, int B, int nin, float2 *out, int nout);
void launch_spectrogram(uwspr_ctx *c, const float *frames, int B);
void launch_spectrum(uwspr_ctx *c, int B);
void launch_coarse(uwspr_ctx *c, int B);
void launch_prep_hyps(uwspr_ctx *c, const uwspr_hyp *abi, dev_hyp *out, int H);
// flat form; taken != null: hypotheses of groups that have their phasor table are left alone (see k4_tonecorr);
// skip_pairs: stage S2 -- the mirrored drift tries launch_tonecorr_dpair computes are left alone
void launch_tonecorr(uwspr_ctx *c, const float *frames, int B, const dev_hyp *hyps, int H,
                     float4 *p, const dev_grp *taken = nullptr, int hyps_per_grp = 1, bool skip_pairs = false);
// stage S2 (hyps 2 s, 2 s + 1 = the two drift tries of slot s): the slots whose tries mirror each other (k4_pair.hip)
void launch_tonecorr_dpair(uwspr_ctx *c, const float *frames, int B, const dev_hyp *hyps, int nslots, float4 *p);
void launch_tonecorr_lag0(uwspr_ctx *c, const float *frames, int B, const dev_grp *grps, int nslots,
                          int64_t nhyps, float4 *p);
void launch_tonecorr_fstage(uwspr_ctx *c, const float *frames, int B, const dev_hyp *hyps, int nslots,
                            int64_t nhyps, float4 *p);
void launch_tonecorr_ring(uwspr_ctx *c, const float *frames, int B, const dev_grp *grps, int G,
                          int NL, int step, int64_t nhyps, float4 *p, int groups_per_slot = 1);
// S5's jiggered shifts with the sample pairs in a register ring (k4_jig.hip)
void launch_tonecorr_jig(uwspr_ctx *c, const float *frames, int B, const dev_grp *grps, int G, int64_t nhyps, float4 *p,
                         int groups_per_slot);
// grid form (one centre per frame, shared sample windows); false = does not fit, use the flat path
bool launch_tonecorr_grid(uwspr_ctx *c, const float *frames, int B, const uwspr_candidate *centres,
                          int nf, const float *df, int ndrift, const float *ddrift, int nlag,
                          const int *dlag_host, const int *dlag_dev, dev_hyp *hyps, float4 *p);
void launch_fold(uwspr_ctx *c, const dev_hyp *hyps, const float4 *p, int H, float *sync,
                 uint8_t *symbols, const float4 *pwin = nullptr, int per_slot = 1);
void launch_keep_try0(uwspr_ctx *c, int nslots, int njig);
// schedule stages; see k5_schedule.hip
void launch_sched_init(uwspr_ctx *c, const uwspr_candidate *cands, const int32_t *npk,
                       int cand_stride, int B, int per_frame);
void launch_fold_step(uwspr_ctx *c, int stage, int ncand, int njig = UWSPR_NJIG);
void launch_pack_slabs(uwspr_ctx *c, const uwspr_candidate *cands, const int32_t *npk,
                       const uwspr_demod_out *dout, int per_frame, int K, uint8_t *slab, int B);
void launch_sched_finish(uwspr_ctx *c, int ncand, int njig = UWSPR_NJIG);
// the whole schedule in one launch (k6_sched.hip); njig = mode-2 tries to produce (17 = all)
void launch_sched_fused(uwspr_ctx *c, const float *frames, int B, const uwspr_candidate *cands,
                        const int32_t *npk, int cand_stride, int per_frame, uwspr_demod_out *out,
                        int njig, const uint8_t *resume = nullptr);
constexpr int kSchedTabFloats = 2 * 5 * 4 * 512;   // per resident workgroup

// profiling brackets
struct prof_scope {
  uwspr_ctx *c; int idx; bool ext;
  // ext = false: events recorded on the stream before / after the bracketed launches (one
  // marker packet each).  ext = true: the ONE launch inside the scope goes through
  // launch_timed(), which hands the pair to hipExtLaunchKernelGGL -- the kernel's own dispatch
  // packet carries the time stamps and no extra packet (and no ~5 us bubble) enters the queue.
  prof_scope(uwspr_ctx *c, int kind, int64_t units, bool ext = false);
  ~prof_scope();
};

template <typename K, typename... A>
inline void launch_timed(uwspr_ctx *c, prof_scope &ps, K kernel, dim3 grid, dim3 block, size_t shmem,
                         A... args) {
  if (ps.idx >= 0 && ps.ext)
    hipExtLaunchKernelGGL(kernel, grid, block, (uint32_t)shmem, c->stream, c->prof_events[ps.idx].a,
                          c->prof_events[ps.idx].b, 0, args...);
  else
    hipLaunchKernelGGL(kernel, grid, block, shmem, c->stream, args...);
}

}  // namespace uwspr
