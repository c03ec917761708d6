// K3 -- coarse (freq, start offset, drift) search and candidate selection.
//
// Reference: FDR_impl::transform hot loop 2, lib/FDR_impl.cc:339-409 with the
// powersum kernel cc:188-210 and the SLM trajectory generator lib/slm.cc:36-121.
// Per candidate: 5 tuned bins x 26 half-symbol offsets x ((2*maxdrift+1) linear
// drifts + 125 straight-line trajectories) = 16380 hypotheses at the defaults,
// each a 162-term sync-vector correlation over the spectrogram.
//
// What makes it cheap without changing a single result:
//  * A hypothesis is fully described by its sequence of bin offsets ifd-ifr over
//    the 162 symbols.  Those sequences are built once per context with the
//    reference's exact expressions (cc:353 binary64 / cc:382-385 binary32) and
//    DEDUPLICATED: at the defaults only 38 of the 126 per-cell sequences are
//    distinct (the SLM reach is a few bins, so many trajectories quantise to the
//    same path).  Identical sequences give bit-identical metrics, so each distinct
//    one is evaluated once (130 x 38 = 4940 evaluations per candidate, not 16380)
//    and the selection below reads it back through a hypothesis -> sequence map.
//    The sequences are stored as 16-bit byte offsets into the LDS tile (k3_seq_entry) and stay in
//    HBM/L2 (82 KB per ifr window), read 16 B = 8 symbols at a time two loads ahead.
//  * The spectrogram window the candidate can touch is staged once into LDS as
//    float4 {sqrt ps[row][c-3], [c-1], [c+1], [c+3]} per (row, centre column c):
//    one ds_read_b128 gather per symbol instead of four gathers + four sqrt.
//  * Most nonlinear hypotheses cannot be accepted and are never evaluated (exactly):  a metric is
//    ss/pow with |ss| <= pow up to rounding, so |sync| <= 1.0001, and the nonlinear rule accepts on
//    sync/best > threshold (cc:392) -- impossible once the running best is >= 1.001/threshold
//    (0.1001 at the flowgraph's threshold = 10).  While best > 0 it only grows, and it is at least
//    the largest linear metric seen so far.  So: the linear metrics of all 130 cells first; c* = the
//    first cell whose linear metric reaches the bound; every sequence is evaluated only for the cells
//    before c* (where a nonlinear hypothesis may still win), and the rule is replayed over
//    {all hypotheses of cells < c*} then {linear hypotheses of cells >= c*}.  On a real signal c* is
//    the cell of the signal's own bin and timing (53 of 130 on the benchmark frames): 2 100 sequence
//    evaluations instead of 4 940.  threshold <= 0, a requested metric grid, or a candidate whose
//    linear metrics never reach the bound: everything is evaluated.
//  * One lane = one (cell, sequence), 162 sequential steps, so ss and pow
//    accumulate in the reference's order (cc:207-209): bit-identical metrics.
//  * Wave 0 then replays the reference's ORDER-DEPENDENT running-best rule over
//    the hypotheses that can still be accepted, in reference order (strict > for
//    linear cc:360, ratio against the running best for nonlinear cc:392) with
//    ballots: 64 hypotheses per step, serialising only on acceptances.
// One 512-thread workgroup per candidate; everything between the spectrogram
// tile read and the 48-byte candidate record stays in LDS.
// Roofline: LDS-gather / VALU bound; HBM traffic is the tile once (~60 KB).
#include <algorithm>
#include <cstdlib>

#include "uwspr_internal.h"

#pragma clang fp contract(off)

namespace uwspr {

// 8 wavefronts per candidate: alone the kernel is as fast as with 16 (52 vs 50 us: its rounds are latency-bound),
// and it leaves half of a CU's register file to the other lanes' kernels while more than half of its own
// wavefront cycles are waits (+2.4 % frames/s under three streams; 4 wavefronts: 74 us alone, slower overall)
#ifndef K3_THREADS_N
#define K3_THREADS_N 512
#endif
constexpr int K3_THREADS = K3_THREADS_N;
// Offset sequences: per (ifr row, distinct sequence) K3_SEQ_WORDS words = 168 x u16 (162 used; 16-B loads), entry k =
// byte offset of the symbol's float4 in the tile relative to the float4 of (row k0 + 2*32*(k/32), the
// cell's own centre column): ((2*(k mod 32))*tp + (ifd-ifr)[k] - off_min) * 16, tp = tile row pitch in
// float4.  Segments of 32 symbols keep that below 65536 for every pitch the 160 KB tile allows (<= 29).
// Pitch: consecutive lanes are consecutive sequences of one cell (same row: offsets differ by < 16
// float4, no conflict inside a ds_read_b128 lane group), but a group that straddles two cells reads
// rows one apart, i.e. float4 indices tp + (difference of offsets) apart: with tp = nc = 13 every
// pair three columns apart collides (27 % of the kernel's LDS cycles); tp = 8 (mod 16) leaves only the
// pairs eight columns apart (option "k3_pitch"=24; see coarse_tile_pitch() for why it is not the default).
//
// Three tile forms (fdr_consts::k3_mode, chosen at context creation by what fits the 160 KB of LDS):
//   K3_TILE_F4     float4 {sqrt ps[c-3], [c-1], [c+1], [c+3]} per (row, centre c) in LDS: one
//                  ds_read_b128 per symbol; every flowgraph-default geometry up to cf ~ 4500
//   K3_TILE_F1     the plain sqrt row (nc + 6 floats per row) in LDS, four 4-byte gathers per symbol:
//                  a quarter of the bytes, for search reaches up to ~ 100 columns (cf ~ 15000)
//   K3_TILE_F1_HBM the same row image in a per-workgroup HBM/L2 scratch: any reach, slowly
// F1 entries are ((2*(k mod 8))*tp + off - off_min) * 4 with tp = the row pitch in floats.
constexpr int K3_SEQ_WORDS = 84;
enum { K3_TILE_F4 = 0, K3_TILE_F1 = 1, K3_TILE_F1_HBM = 2 };
__host__ __device__ constexpr int k3_seg_syms(int mode) { return mode == K3_TILE_F4 ? 32 : 8; }
__host__ __device__ constexpr int k3_unit(int mode) { return mode == K3_TILE_F4 ? 16 : 4; }
__host__ __device__ inline uint32_t k3_seq_entry(int mode, int k, int off, int off_min, int tp) {
  return (uint32_t)((2 * (k % k3_seg_syms(mode))) * tp + off - off_min) * (uint32_t)k3_unit(mode);
}
typedef uint32_t k3_u4 __attribute__((ext_vector_type(4)));   // HIP's uint4 class cannot live behind an address-space pointer
constexpr int K3_SLICE_PITCH = 68;   // floats per 64-hypothesis slice of the expanded metrics (16-B pad)

// lane ^ 1 and lane ^ 2 inside a quad: DPP quad_perm, no LDS round trip
__device__ __forceinline__ float quad_swap1(float v) {
  return __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(v), 0xB1, 0xF, 0xF, true));
}
__device__ __forceinline__ float quad_swap2(float v) {
  return __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(v), 0x4E, 0xF, 0xF, true));
}

struct k3_lds_layout { size_t tile, sync, umap, misc, total; };
__host__ __device__ inline k3_lds_layout k3_layout(const fdr_consts &f) {
  k3_lds_layout l;
  l.tile = 0;
  l.sync = l.tile + (f.k3_mode == K3_TILE_F1_HBM ? 0 : (size_t)f.n * f.tp * k3_unit(f.k3_mode));
  l.umap = l.sync + (size_t)UWSPR_NIFR * UWSPR_NK0 * f.umax * 4;
  l.misc = l.umap + (((size_t)UWSPR_NIFR * f.cell_hyps * 2 + 15) & ~(size_t)15);   // c* (all LDS is dynamic: the
  l.total = l.misc + 16;                                                           //  kernel may use the full 160 KB)
  // after the evaluation the tile + offset-table range is reused for the expanded
  // metrics [64-value slices at K3_SLICE_PITCH] + 3 floats per slice: grow it if needed
  const size_t reuse = ((size_t)K3_SLICE_PITCH + 3) * (size_t)((f.ntot + 63) >> 6) * 4;
  if (reuse > l.sync) {
    const size_t extra = (reuse - l.sync + 15) & ~(size_t)15;
    l.sync += extra; l.umap += extra; l.misc += extra; l.total += extra;
  }
  return l;
}

template <int MODE>
__global__ __launch_bounds__(K3_THREADS, 1024 / K3_THREADS) void k3_coarse(
    const float *__restrict__ ps, fdr_consts f, const uint32_t *__restrict__ uoff_tab,
    const uint16_t *__restrict__ umap_tab, uwspr_candidate *__restrict__ cands,
    const int32_t *__restrict__ work, float *__restrict__ syncgrid, int grid_cap,
    float *__restrict__ tile_scratch) {
  extern __shared__ __align__(16) unsigned char smem[];
  const k3_lds_layout lay = k3_layout(f);
  constexpr int K3_SEG_SYMS = k3_seg_syms(MODE), UNIT = k3_unit(MODE);
  // the tile: [n][tp] float4 (nc used per row) or [n][tp] floats (nc + 6 used), in LDS or in HBM
  typedef typename std::conditional<MODE == K3_TILE_F1_HBM, const char __attribute__((address_space(1))) *,
                                    const char *>::type tile_bytes;
  float4 *tile = reinterpret_cast<float4 *>(smem + lay.tile);
  float *tile1 = MODE == K3_TILE_F1_HBM ? tile_scratch + (size_t)blockIdx.x * f.n * f.tp
                                        : reinterpret_cast<float *>(smem + lay.tile);
  float *syncbuf = reinterpret_cast<float *>(smem + lay.sync);       // [130][umax]
  uint16_t *umap = reinterpret_cast<uint16_t *>(smem + lay.umap);    // [5][cell_hyps]

  int &s_cstar = *reinterpret_cast<int *>(smem + lay.misc);
  const int tid = threadIdx.x;
  const int nwork = work[0];
#ifdef K3_STAMPS
  long long t_a = clock64(), t_b = 0, t_c = 0, t_d = 0;
#endif
  // persistent workgroups: K2 compacted the (frame, candidate) pairs into a work
  // list, so no workgroup is launched only to find it has no candidate
  for (int wi = blockIdx.x; wi < nwork; wi += gridDim.x) {
  const int item = work[1 + wi];
  const int b = item / f.cand_slots, j = item - b * f.cand_slots;
  uwspr_candidate *cand = cands + (size_t)b * f.maxfreqs + j;
  const float freq0 = cand->freq;
  // cc:341: if0 = freq/df + m (binary32), truncated
  const int if0 = (int)(ieee_divf(freq0, f.df) + (float)f.m);
  const int r0 = if0 - 2 - f.ifr_lo;  // first of the 5 rows of the offset tables

  // ---- stage the sqrt tile and this candidate's offset rows ----------------
  const float *psb = ps + (size_t)b * f.n * f.band_w;
  const int c0 = if0 - 2 + f.off_min - f.band_lo;  // band column of centre index 0
  if (MODE == K3_TILE_F4) {
    for (int idx = tid; idx < f.n * f.nc; idx += K3_THREADS) {
      int row = idx / f.nc, ci = idx - row * f.nc;
      const float *pr = psb + (size_t)row * f.band_w + c0 + ci;
      tile[row * f.tp + ci] = make_float4(ieee_sqrtf(pr[-3]), ieee_sqrtf(pr[-1]), ieee_sqrtf(pr[1]),
                                          ieee_sqrtf(pr[3]));
    }
  } else {   // column q of a row = band column c0 - 3 + q: centre ci reads q = ci, ci + 2, ci + 4, ci + 6
    const int ncw = f.nc + 6;
    for (int idx = tid; idx < f.n * ncw; idx += K3_THREADS) {
      int row = idx / ncw, q = idx - row * ncw;
      tile1[row * f.tp + q] = ieee_sqrtf(psb[(size_t)row * f.band_w + c0 - 3 + q]);
    }
  }
  // the offset sequences stay in HBM/L2 (82 KB would not leave room for the tile) and are read
  // through a GLOBAL-address-space pointer: a generic one makes them flat loads, which count on
  // lgkmcnt as well and turn every LDS wait of the gather loop into a wait for L2
  typedef const k3_u4 __attribute__((address_space(1))) *seq_ptr;
  const seq_ptr uo_base = (seq_ptr)(uintptr_t)(uoff_tab + (size_t)r0 * f.umax * K3_SEQ_WORDS);
  for (int idx = tid; idx < UWSPR_NIFR * f.cell_hyps; idx += K3_THREADS)
    umap[idx] = umap_tab[(size_t)r0 * f.cell_hyps + idx];
  __syncthreads();
#ifdef K3_STAMPS
  t_b = clock64();
#endif

  // ---- one lane per (cell, offset sequence) ------------------------------------------------
  // Work items: first the linear hypotheses' sequences of all cells (NCELL * nlin items, reference
  // order), then every sequence of cell 0, 1, 2, ...  The first round(s) cover the linear items (one
  // round of K3_THREADS at the defaults, filled up with full-set items); then c* is known and the item
  // list ends after cell c* - 1.
  const int segstep = 2 * K3_SEG_SYMS * f.tp * UNIT;
  const int hc = f.cell_hyps;
  constexpr int NCELL = UWSPR_NIFR * UWSPR_NK0;
  const int nlin_items = NCELL * f.nlin;
  const bool want_grid = syncgrid != nullptr && j < grid_cap;
  // |sync| <= 1.0001 (648 roundings of 2^-24 each): no nonlinear acceptance once best >= bound
  const float bound = (f.threshold > 0.0f && !want_grid) ? 1.001f / f.threshold : __builtin_inff();
  // the rounds that cover the linear items (one at the defaults), filled up with full-set items
  int nitems = min(nlin_items + NCELL * f.umax, (nlin_items + K3_THREADS - 1) / K3_THREADS * K3_THREADS);
  int cstar = NCELL;
  bool have_cstar = false;
  // (splitting the items left after round 0 evenly over the remaining rounds is slower: a round is
  // latency-bound below ~12 wavefronts, so one full round + a short one beats two medium ones)
  for (int base = 0; base < nitems; base += K3_THREADS) {
    const int g = base + tid;
    if (g < nitems) {
    int cell, u;
    if (g < nlin_items) {
      cell = g / f.nlin;
      u = umap[(cell / UWSPR_NK0) * hc + (g - cell * f.nlin)];
    } else {
      const int q = g - nlin_items;
      cell = q / f.umax; u = q - cell * f.umax;
    }
    const int ifr_i = cell / UWSPR_NK0, k0 = cell - ifr_i * UWSPR_NK0;
    const seq_ptr ot = uo_base + (ifr_i * f.umax + u) * (K3_SEQ_WORDS / 4);
    tile_bytes tb = (MODE == K3_TILE_F4 ? (tile_bytes)reinterpret_cast<const char *>(tile)
                                        : (tile_bytes)(uintptr_t)tile1) + (k0 * f.tp + ifr_i) * UNIT;
    float ss = 0.0f, pw = 0.0f;
    // Software pipeline over groups of 4 symbols: the 4 gathers of the next group are in flight
    // while this one is accumulated, and the offset words (16 B = 8 symbols per load) are
    // requested two loads ahead of their use: neither the LDS nor the L2 latency is exposed.
    float4 Pa[4], Pb[4];
    k3_u4 wa = ot[0], wb = ot[1], wc = ot[2];
    auto gather1 = [&](uint32_t e) -> float4 {
      if (MODE == K3_TILE_F4) return *reinterpret_cast<const float4 *>((const char *)tb + e);
      typedef typename std::conditional<MODE == K3_TILE_F1_HBM, const float __attribute__((address_space(1))) *,
                                        const float *>::type fp;
      const fp q = (fp)(tb + e);
      return make_float4(q[0], q[2], q[4], q[6]);
    };
    auto gather2 = [&](float4 *P, uint32_t w) {
      P[0] = gather1(w & 0xffffu);
      P[1] = gather1(w >> 16);
    };
    auto accumulate = [&](const float4 *P, int k, int count) {
#pragma unroll
      for (int kk = 0; kk < 4; kk++) {
        if (kk < count) {
          const float4 q = P[kk];
          const float cm = (q.y + q.w) - (q.x + q.z);
          ss = pr3_bit(k + kk) ? ss + cm : ss - cm;  // (2*pr3[k]-1)*cm, cc:207
          pw = pw + q.x; pw = pw + q.y; pw = pw + q.z; pw = pw + q.w;  // cc:209
        }
      }
    };
    gather2(Pa, wa.x); gather2(Pa + 2, wa.y);
#pragma unroll
    for (int k8 = 0; k8 < K3_SEQ_WORDS / 4; k8++) {
      const int k = 8 * k8;   // symbols k..k+3 are in Pa
      if (k + 4 < UWSPR_NSYM) { gather2(Pb, wa.z); gather2(Pb + 2, wa.w); }
      accumulate(Pa, k, UWSPR_NSYM - k);
      if (k + 8 < UWSPR_NSYM) {
        if ((k + 8) % K3_SEG_SYMS == 0) tb += segstep;   // next segment's base
        gather2(Pa, wb.x);
        if (k + 10 < UWSPR_NSYM) gather2(Pa + 2, wb.y);
      }
      if (k + 4 < UWSPR_NSYM) accumulate(Pb, k + 4, UWSPR_NSYM - k - 4);
      wa = wb; wb = wc;
      if (k8 + 3 < K3_SEQ_WORDS / 4) wc = ot[k8 + 3];
    }
    syncbuf[cell * f.umax + u] = ieee_divf(ss, pw);  // cc:357,390 (a sequence evaluated twice writes the same value twice)
    }
    if (!have_cstar && base + K3_THREADS >= nlin_items) {   // every linear metric is in syncbuf now
      have_cstar = true;
      // c*: first cell (reference order) with a linear metric >= bound; NaN never compares true
      __syncthreads();
      if (tid < 64) {
        int first = NCELL * f.nlin;
        for (int blk = 0; blk < nlin_items && first == NCELL * f.nlin; blk += 64) {
          const int q = blk + tid;
          bool hit = false;
          if (q < nlin_items) {
            const int c = q / f.nlin;
            hit = syncbuf[c * f.umax + umap[(c / UWSPR_NK0) * hc + (q - c * f.nlin)]] >= bound;
          }
          const unsigned long long m = __ballot(hit);
          if (m != 0ull) first = blk + __ffsll((long long)m) - 1;
        }
        if (tid == 0) s_cstar = first / f.nlin;   // NCELL when no linear metric reaches the bound
      }
      __syncthreads();
      cstar = s_cstar;
      nitems = nlin_items + cstar * f.umax;
    }
  }
  __syncthreads();
#ifdef K3_STAMPS
  t_c = clock64();
#endif

  if (want_grid) {   // (cstar == NCELL: every sequence was evaluated)
    float *gout = syncgrid + ((size_t)b * grid_cap + j) * f.ntot;
    for (int g = tid; g < f.ntot; g += K3_THREADS) {
      const int cell = g / hc, h = g - cell * hc;
      gout[g] = syncbuf[cell * f.umax + umap[(cell / UWSPR_NK0) * hc + h]];
    }
  }

  // ---- replay the running-best selection in reference order ---------------
  // The rule (cc:360 strict > for linear, cc:392 ratio against the RUNNING best
  // for nonlinear) is an order-dependent fold over all ntot hypotheses.  It is
  // replayed exactly, but cheaply:
  //  (1) all threads expand the metrics into reference order (`full`, reusing the
  //      tile / offset-table LDS, no longer needed) and reduce every 64-value
  //      slice to {max over linear, max and min over nonlinear} hypotheses;
  //  (2) wave 0 scans: 64 slice summaries at a time are tested against the
  //      current best -- the predicates are monotone in v (v > best; v/best > thr
  //      rises with v for best > 0 and falls for best < 0), so a slice whose
  //      extreme value fails cannot contain an acceptance and is skipped -- and
  //      only slices that may accept are scanned value by value with ballots.
  // The list that is replayed, in reference order: entries e < nfull = c* x hc are all hypotheses of
  // the cells before c*; the rest are the nlin linear hypotheses of each cell from c* on.
  const int nfull = cstar * hc;
  const int ne = nfull + (NCELL - cstar) * f.nlin;
  auto cell_h = [&](int e, int &cell, int &h) {
    if (e < nfull) { cell = e / hc; h = e - cell * hc; }
    else { const int r = e - nfull; cell = cstar + r / f.nlin; h = r - (cell - cstar) * f.nlin; }
  };
  float *full = reinterpret_cast<float *>(smem);                       // [nslice][K3_SLICE_PITCH]
  const int nslice = (ne + 63) >> 6;
  float *ssum = full + (size_t)nslice * K3_SLICE_PITCH;                // [nslice][3]
  // (1a) expansion, consecutive lanes = consecutive entries (conflict-free LDS traffic)
  for (int e = tid; e < ne; e += K3_THREADS) {
    int cell, h;
    cell_h(e, cell, h);
    full[e + (e >> 6) * (K3_SLICE_PITCH - 64)] = syncbuf[cell * f.umax + umap[(cell / UWSPR_NK0) * hc + h]];
  }
  __syncthreads();
  {
    // (1b) slice summaries: four lanes per slice, 16 consecutive values each (four ds_read_b128;
    // the 4-float pad per slice keeps the 16 lanes of a read group on distinct banks), reduced in
    // registers and then over the quad with two DPP steps -- not 18 ds_bpermute per 64 values
    const float ninf = -__builtin_inff(), pinf = __builtin_inff();
    for (int sl = tid >> 2; sl < nslice; sl += K3_THREADS / 4) {
      const int qd = tid & 3;
      const int g0 = sl * 64 + qd * 16;
      const float4 *src = reinterpret_cast<const float4 *>(full + (size_t)sl * K3_SLICE_PITCH + qd * 16);
      float vv[16];
#pragma unroll
      for (int i = 0; i < 4; i++) {
        const float4 q4 = src[i];
        vv[4 * i] = q4.x; vv[4 * i + 1] = q4.y; vv[4 * i + 2] = q4.z; vv[4 * i + 3] = q4.w;
      }
      int h = g0 % hc;
      float mlin = ninf, mxnl = ninf, mnnl = pinf;
#pragma unroll
      for (int i = 0; i < 16; i++) {
        const float v = vv[i];
        // NaN never satisfies a predicate: keep it out of the extremes
        const bool use = (g0 + i < ne) && (v == v);
        const bool lin = (g0 + i >= nfull) || h < f.nlin;
        mlin = fmaxf(mlin, (use && lin) ? v : ninf);
        mxnl = fmaxf(mxnl, (use && !lin) ? v : ninf);
        mnnl = fminf(mnnl, (use && !lin) ? v : pinf);
        h = (h + 1 == hc) ? 0 : h + 1;
      }
      mlin = fmaxf(mlin, quad_swap1(mlin)); mlin = fmaxf(mlin, quad_swap2(mlin));
      mxnl = fmaxf(mxnl, quad_swap1(mxnl)); mxnl = fmaxf(mxnl, quad_swap2(mxnl));
      mnnl = fminf(mnnl, quad_swap1(mnnl)); mnnl = fminf(mnnl, quad_swap2(mnnl));
      if (qd == 0) { ssum[3 * sl] = mlin; ssum[3 * sl + 1] = mxnl; ssum[3 * sl + 2] = mnnl; }
    }
  }
  __syncthreads();
#ifdef K3_STAMPS
  long long t_c2 = clock64();
#endif
  if (tid < 64) {
    float best = -1e30f;
    int gbest = -1;
    int pos = 0;  // first slice not yet decided
    while (pos < nslice) {
      // find the first slice >= pos that may contain an acceptance
      int hit = -1;
      for (int blk = pos & ~63; blk < nslice && hit < 0; blk += 64) {
        const int sl = blk + tid;
        bool may = false;
        if (sl >= pos && sl < nslice) {
          const float mlin = ssum[3 * sl], mxnl = ssum[3 * sl + 1], mnnl = ssum[3 * sl + 2];
          may = mlin > best;
          if (best > 0.0f) may = may || (ieee_divf(mxnl, best) > f.threshold);
          else if (best < 0.0f) may = may || (ieee_divf(mnnl, best) > f.threshold);
          else may = true;  // best == +-0: scan exactly
        }
        const unsigned long long m = __ballot(may);
        if (m != 0ull) hit = blk + __ffsll((long long)m) - 1;
      }
      if (hit < 0) break;
      // exact scan of slice `hit`
      const int g = hit * 64 + tid;
      const bool in = g < ne;
      const float v = in ? full[hit * K3_SLICE_PITCH + tid] : 0.0f;
      const bool lin = in && (g >= nfull || (g % hc) < f.nlin);
      int start = 0;
      for (;;) {
        const bool pred = in && tid >= start &&
                          (lin ? (v > best) : (ieee_divf(v, best) > f.threshold));
        const unsigned long long mask = __ballot(pred);
        if (mask == 0ull) break;
        const int first = __ffsll((long long)mask) - 1;
        best = __shfl(v, first);
        gbest = hit * 64 + first;
        start = first + 1;
      }
      pos = hit + 1;
    }
    if (tid == 0) {
      cand->sync = best;
      if (gbest >= 0) {
        int cell, h;
        cell_h(gbest, cell, h);
        const int ifr_i = cell / UWSPR_NK0, k0 = cell - ifr_i * UWSPR_NK0;
        cand->shift = 128 * k0;                                // cc:361,397
        cand->freq = (float)(if0 - 2 + ifr_i - f.m) * f.df;    // cc:362,398
        if (h < f.nlin) {
          cand->m_type = UWSPR_LINEAR;
          cand->m_nonlinear.V1 = 0.0; cand->m_nonlinear.V2 = 0.0;
          cand->m_nonlinear.p1 = 0; cand->m_nonlinear.p2 = 0;
          cand->m_linear.drift = (float)(h - f.maxdrift);      // cc:366
        } else {
          const int s = h - f.nlin;  // slm.cc:76-116: p2 fastest, then V1, then V2
          cand->m_type = UWSPR_NONLINEAR;
          cand->m_nonlinear.V1 = (double)((s / 5) % 5) - 2.0;
          cand->m_nonlinear.V2 = (double)(s / 25) - 2.0;
          cand->m_nonlinear.p1 = 0;
          cand->m_nonlinear.p2 = 50 + 200 * (s % 5);
        }
      }
    }
  }
  __syncthreads();  // LDS is reused by the next work item
#ifdef K3_STAMPS
  t_d = clock64();
  if (tid == 0 && blockIdx.x == 7) printf("K3 stamps (cycles): stage %lld eval %lld expand %lld fold %lld\n", t_b - t_a, t_c - t_b, t_c2 - t_c, t_d - t_c2);
#endif
  }
}

size_t coarse_lds_bytes(const fdr_consts &f) { return k3_layout(f).total; }
int coarse_seq_words() { return K3_SEQ_WORDS; }
// Picks the tile form and its row pitch (f.k3_mode, f.tp); returns the floats of HBM scratch one
// workgroup needs (0 unless the tile lives in HBM).
size_t coarse_plan(fdr_consts &f, int pitch_opt, int tile_opt) {   // options "k3_pitch", "k3_tile"
  // measured: pitch 24 instead of 13 made the kernel 3 % faster for 61 KB more LDS (the conflicts
  // are not what bounds the gather loop) -- the compact tile stays the default
  int want = f.nc;
  if (pitch_opt >= f.nc) want = pitch_opt;
  const int force = tile_opt;   // tests: 0 / 1 / 2
  f.k3_mode = K3_TILE_F4;
  if (force <= K3_TILE_F4) {
    f.tp = want;
    if (k3_layout(f).total <= 160 * 1024) return 0;
    f.tp = f.nc;
    if (k3_layout(f).total <= 160 * 1024) return 0;
  }
  f.tp = f.nc + 6;
  f.k3_mode = K3_TILE_F1;
  if (force <= K3_TILE_F1 && k3_layout(f).total <= 160 * 1024) return 0;
  f.k3_mode = K3_TILE_F1_HBM;
  return (size_t)f.n * f.tp;
}
uint32_t coarse_seq_entry(const fdr_consts &f, int k, int off) {
  return k3_seq_entry(f.k3_mode, k, off, f.off_min, f.tp);
}

void launch_coarse(uwspr_ctx *c, int B) {
  const fdr_consts &f = c->fc;
  prof_scope ps(c, UWSPR_K_COARSE, (int64_t)B);
  const long long items_max = (long long)f.cand_slots * B;
  const int grid = (int)std::min<long long>(items_max, c->num_cus);
  auto go = [&](auto kern) {
    hipLaunchKernelGGL(kern, dim3(grid), dim3(K3_THREADS), coarse_lds_bytes(f), c->stream,
                       c->d_ps, f, c->d_off, c->d_umap, c->cur_cands, c->d_work, c->d_syncgrid,
                       c->d_syncgrid ? c->grid_cap : 0, c->d_k3_tile);
  };
  if (f.k3_mode == K3_TILE_F4) go(k3_coarse<K3_TILE_F4>);
  else if (f.k3_mode == K3_TILE_F1) go(k3_coarse<K3_TILE_F1>);
  else go(k3_coarse<K3_TILE_F1_HBM>);
}

int coarse_configure(const fdr_consts &f) {
  size_t need = coarse_lds_bytes(f);
  if (need > 160 * 1024) return -1;
  // the attribute belongs to the function, not to a context: always the device maximum, so that a
  // second context with a smaller tile cannot lower the limit under an earlier one
  const void *fn[3] = {reinterpret_cast<const void *>(k3_coarse<K3_TILE_F4>),
                       reinterpret_cast<const void *>(k3_coarse<K3_TILE_F1>),
                       reinterpret_cast<const void *>(k3_coarse<K3_TILE_F1_HBM>)};
  for (const void *p : fn)
    if (hipFuncSetAttribute(p, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess) return -2;
  return 0;
}

}  // namespace uwspr
