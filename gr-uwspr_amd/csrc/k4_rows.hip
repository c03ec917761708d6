// K4, rows form -- the tone correlations of the refinement schedule's stages S0, S1, S3, S4 and S5
// (sync_and_demodulate_impl.cc:409-419, 444-452, 457-468 calling cc:126-256), one launch per stage.
//
// Reference: the hot loop of sync_and_demodulate_impl::sync_and_demodulate, cc:167-212 (per-symbol
// frequency cc:170-183, phasor recurrence cc:186-199, correlation cc:200-211).
//
// An OPTION ("stage_kernels" = 2), not the default.  It is the balanced shape of the staged form: a launch is 768
// workgroups of EQUAL work -- one per (candidate slot, third of its 162 symbols) -- of 4 HS wavefronts, exactly 3 per
// CU, all resident from the first cycle to the last (the packed kernels of k4_tonecorr.hip put two workgroups on 120
// CUs and three on 136).  On one HIP stream it is the faster form (K4 0.317 ms per 256-frame step against 0.332); under
// three streams it is the slower one (653 k frames/s against 712 k): it fills every CU by itself, and 54 of 64 lanes
// carry a symbol.  What was measured on the way (profiles/r04_*): a wavefront issues a binary32
// VALU instruction at best every 4.5 cycles whatever the dependences between its instructions; equal priorities are
// served oldest first, so of three equal workgroups on a CU the last dispatched finishes 15 us after the first -- hence
// the priority rotation in chunk_turn(); with the phasors fetched by scalar loads a wavefront is bound by their latency
// (590 cycles per 64-instruction block, one block ahead is not enough) and a CU by the scalar path's one load per ~40
// cycles; through LDS the kernel sits where every K4 form sits, at ~0.6 of the VALU issue rate with the LDS array ~70 %
// busy; removing its arithmetic leaves 25 of S0's 44 us (loader + barriers), cache-resident samples change nothing.
//
// Mapping.  Workgroup = (slot, third): symbols 54 t .. 54 t + 53, one per LANE (lane 54 of the last third carries S0's
// virtual 163rd row).  Wavefront w = tone (w & 3) x hypothesis subset (w >> 2): a lane accumulates inp / quad of its
// symbol window against ONE tone for the subset's hypotheses, every accumulator seeing exactly the reference's sequence
// of binary32 operations (cc:206-207: no FMA, no tree).
//  * Samples, sample-major: the workgroup streams its rows [L0 + 256 i, L0 + 256 i + 256 + span) through a
//    double-buffered LDS image, 32 samples per row and chunk (coalesced 8-byte loads; cc:205's n > 0 && n < np test is
//    applied by the loader: a skipped sample is a zero, which leaves inp / quad unchanged), one barrier per chunk.  A
//    hypothesis whose lag is L0 + D sees stream position a as its sample k = a - D: the lag sweeps (S0: 4 lags 64
//    apart, S3: 5 lags 16 apart, S5: 17 lags 8 apart) are ONE pass over the rows, nothing is loaded per lag.  Which
//    hypotheses are inside their windows at a stream position is compile time (phases: rows_make_phases).
//  * Phasors.  When the per-symbol frequency does not depend on the symbol (drift 0 or the straight-line model with
//    t = 0: the reference's `fplast` cache hits for the same reason, cc:185) the sequence c[k], s[k] of cc:186-199 is a
//    table per (frequency, tone), built once per slot by the schedule kernels (k5_fold_schedule.hip: ptab_build; set A
//    around the candidate frequency for S0 / S1, set B around the S2 result for S3 / S4 / S5).  The workgroup copies
//    the chunk's slices (frequency stages) or the whole table (lag stages) into an LDS image with vector loads; a lane
//    reads two steps per ds_read_b128 (the same address in every lane: broadcast).  Stage 5 (17 lags) keeps the
//    generic walk with scalar loads (its phased code would be 50 KB).  With a per-symbol frequency (a drifting linear
//    model) every lane runs its own recurrences (cc:193-195), 14 instructions per sample and hypothesis -- same
//    kernel, workgroup-uniform branch.
//  * S0's fifth lag is its first one symbol later: (lag 4, symbol i) is (lag 0, symbol i + 1) whenever the frequency
//    does not depend on the symbol; lane 54 of the last third walks a virtual symbol 162 for (lag 4, symbol 161).
#include <type_traits>

#include "k4_common.h"

#pragma clang fp contract(off)

namespace uwspr {

constexpr int KR_TROWS = 54;                 // symbols per workgroup: 162 = 3 x 54
constexpr int KR_NROWS = KR_TROWS + 1;       // + the virtual row of the S0 wrap
constexpr int KR_CH = 32;                    // samples per row and staged chunk
constexpr int KR_ROWDW = 2 * KR_CH + 4;      // dwords per staged row: 32 samples x 8 B + 16 B pad (conflict-free b128 column reads)

typedef float kr_f16 __attribute__((ext_vector_type(16)));
#define KR_CONST __attribute__((address_space(4)))

// The stages as compile-time geometry.  Hypothesis h of a stage has lag L0 + 8 dk8(h); NH hypotheses per slot.
//   S0 (cc:409-415): h = lag index, shift1 - 128 + 64 h            -> dk8 = 8 h
//   S1 / S4 (cc:416-419, 449-452): h = frequency index, one lag     -> dk8 = 0, table h of the set
//   S3 (cc:444-447): shift1 - 32 + 16 h                             -> dk8 = 2 h
//   S5 (cc:457-468): h = m, the jiggered shifts in ASCENDING order, shift1 - 64 + 8 m; try idt = 2|m-8| - (m < 8)
template <int KIND> struct rows_geom {
  static constexpr int NH = KIND == UWSPR_ROWS_S5 ? UWSPR_NJIG : 5;
  static constexpr bool LAGS = KIND == UWSPR_ROWS_S0 || KIND == UWSPR_ROWS_S3 || KIND == UWSPR_ROWS_S5;
  __host__ __device__ static constexpr int dk8(int h) {
    return KIND == UWSPR_ROWS_S0 ? 8 * h : KIND == UWSPR_ROWS_S3 ? 2 * h : KIND == UWSPR_ROWS_S5 ? h : 0;
  }
  // stream length in 32-sample chunks: 256 samples + the largest lag offset
  static constexpr int NCHUNK = (256 + 8 * dk8(NH - 1) + KR_CH - 1) / KR_CH;
  // output row of hypothesis h within the slot's NH (stage 5: the try number idt)
  __host__ __device__ static constexpr int out_index(int h) {
    return KIND == UWSPR_ROWS_S5 ? (h == 8 ? 0 : (h < 8 ? 2 * (8 - h) - 1 : 2 * (h - 8))) : h;
  }
  // hypotheses of subset `sub` of HS (bit h).  Lag stages: dealt so that every subset has early and late lags (a
  // lag is walked only while the stream position is inside its window)
  __host__ __device__ static constexpr uint32_t subset(int HS, int sub) {
    if (HS == 1) return (1u << NH) - 1u;
    uint32_t m = 0;
    for (int h = 0; h < NH; h++) {
      int s;
      if (KIND == UWSPR_ROWS_S5) s = h % HS;
      else if (KIND == UWSPR_ROWS_S0) s = (h == 0 || h == 3) ? 0 : 1;       // {0, 3}, {1, 2, 4}
      else if (KIND == UWSPR_ROWS_S3) s = (h == 0 || h == 4 || h == 2) ? 0 : 1;
      else s = h < 3 ? 0 : 1;                                               // {0, 1, 2}, {3, 4}
      if (s == sub) m |= 1u << h;
    }
    return m;
  }
};

// Phases of a wavefront's walk: runs of 8-sample units over which the set of its hypotheses that are inside their
// windows does not change (compile time: hypothesis h is walked for units dk8(h) .. dk8(h) + 31).
struct rows_phase { uint32_t am; int u0, n; };
struct rows_phases { rows_phase p[24]; int n; };
template <int KIND>
__host__ __device__ constexpr rows_phases rows_make_phases(uint32_t hm, int nunits) {
  using G = rows_geom<KIND>;
  rows_phases r{};
  r.n = 0;
  uint32_t cur = 0xffffffffu;
  for (int u = 0; u < nunits; u += 2) {   // (two units at a time: every lag offset of the phased stages is even)
    uint32_t am = 0;
    for (int h = 0; h < G::NH; h++)
      if (((hm >> h) & 1u) && u - G::dk8(h) >= 0 && u - G::dk8(h) < 32) am |= 1u << h;
    if (am != cur) { r.p[r.n].am = am; r.p[r.n].u0 = u; r.p[r.n].n = 0; r.n++; cur = am; }
    r.p[r.n - 1].n += 2;
  }
  return r;
}
__host__ __device__ constexpr int rows_first_bit(uint32_t m) { int b = 0; while (b < 31 && !((m >> b) & 1u)) b++; return b; }
__host__ __device__ constexpr int rows_popc(uint32_t m) { int c = 0; for (int b = 0; b < 32; b++) c += (m >> b) & 1u; return c; }
__host__ __device__ constexpr int rows_nth_bit(uint32_t m, int n) {
  int c = 0;
  for (int b = 0; b < 32; b++) if ((m >> b) & 1u) { if (c == n) return b; c++; }
  return 0;
}

#ifdef KR_STAMPS   // diagnostic build only (tools/kr_stamps.py): where a k4_rows wavefront's cycles go
constexpr int KR_STAMP_WAVES = 8192;
__device__ unsigned long long g_kr_stamps[KR_STAMP_WAVES * 8];
#ifndef KR_STAMP_KIND
#define KR_STAMP_KIND UWSPR_ROWS_S1
#endif
#endif

template <int KIND, int HS, bool FAST>
__global__ __launch_bounds__(256 * HS) __attribute__((amdgpu_waves_per_eu(3 * HS, 3 * HS))) void k4_rows(   // three workgroups per CU: the whole launch resident at once

    const float2 *__restrict__ frames, int fstride, int np, int nframes, const dev_row *__restrict__ rows,
    const dev_hyp *__restrict__ hyps, int nslots, const float2 *__restrict__ ptab, float *__restrict__ p_out) {
  using G = rows_geom<KIND>;
  constexpr int NH = G::NH;
  constexpr int NT = 256 * HS;
  __shared__ __align__(16) float stage[2][KR_NROWS * KR_ROWDW];
  // phasors: frequency stages [buffer][tone][step pair of the chunk][hypothesis], lag stages the whole table [tone][step pair]
  __shared__ __align__(16) float4 ptl[2 * 4 * (KR_CH / 2) * 5];
  static_assert(2 * 4 * (KR_CH / 2) * 5 >= 4 * 128, "a whole table fits the phasor image");

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int tone = wv & 3, hsub = wv >> 2;
#ifdef KR_STAMPS
  const unsigned long long st_c0 = __builtin_amdgcn_s_memtime(), st_r0 = __builtin_amdgcn_s_memrealtime();
  unsigned long long st_pro = 0, st_bar = 0, st_end = 0;
#endif
  const unsigned wg = xcd_swizzle(blockIdx.x, gridDim.x);
  const int slot = (int)(wg / 3u), third = (int)(wg % 3u);
  if (slot >= nslots) return;   // workgroup-uniform
  const unsigned prio_class = blockIdx.x / ((gridDim.x + 2u) / 3u);

  const dev_row R = rows[slot];
  const bool live = R.frame >= 0 && R.frame < nframes;
  const bool tabled = R.tab >= 0 && ptab != nullptr;
  // S0 wrap: with a symbol-independent frequency lag 4 is lag 0 one symbol later
  const bool wrap = KIND == UWSPR_ROWS_S0 && tabled;
  const uint32_t mask = wrap ? (R.mask & 0xfu) : R.mask;
  const bool virt = wrap && third == 2 && lane == KR_TROWS;     // the virtual symbol 162
  const int sym = KR_TROWS * third + lane;                      // this lane's symbol (lanes < 54; 162 on the virtual lane)
  const bool mine = lane < KR_TROWS;

  if (!live) {   // a dead slot's hypotheses read as zeros (workgroup-uniform, before any barrier)
    if (mine && hsub == 0)
#pragma unroll
      for (int h = 0; h < NH; h++)
        if ((R.mask >> h) & 1u) p_out[(((size_t)slot * NH + G::out_index(h)) * UWSPR_NSYM + sym) * 4 + tone] = 0.0f;
    return;
  }

  // The masks the schedule emits: everything, or everything but the hypothesis that repeats the previous winner
  // (the middle one; try 0 = m 8 of stage 5); S0 with the wrap: lags 0..3.  The table walk is compiled for these.
  constexpr uint32_t PAT_FULL = KIND == UWSPR_ROWS_S0 ? 0xfu : (1u << NH) - 1u;
  constexpr uint32_t PAT_KNOWN = KIND == UWSPR_ROWS_S0 ? 0xfu : PAT_FULL & ~(1u << (KIND == UWSPR_ROWS_S5 ? 8 : 2));
  const bool known = KIND != UWSPR_ROWS_S0 && mask == PAT_KNOWN;   // (any other mask: everything is computed, `mask` is stored)

  // ---- loader: round r of a chunk = row (NT / 32) r + tid / 32, sample tid % 32 -----------------------------------
  const int nrows = KR_TROWS + ((wrap && third == 2) ? 1 : 0);
  constexpr int RPR = NT / KR_CH;                       // rows per loader round
  constexpr int NR = (KR_NROWS + RPR - 1) / RPR;
  const int lk = tid % KR_CH, lr = tid / KR_CH;
#ifdef KR_SAME_FRAME   // (timing experiment: every workgroup reads frame 0 -- cache-resident samples)
  const float2 *fb = frames + (long long)(R.frame & 1) * fstride;
#else
  const float2 *fb = frames + (long long)R.frame * fstride;
#endif
  const int n00 = R.L0 + 256 * KR_TROWS * third;        // first sample of the workgroup's first row
  // chunks the stream needs: 256 samples + the offset of the latest lag that is computed (uniform)
  const uint32_t cmask = tabled ? (known ? PAT_KNOWN : PAT_FULL) : mask;
  const int nch = (256 + 8 * G::dk8(31 - __builtin_clz(cmask | 1u)) + KR_CH - 1) / KR_CH;
  const bool interior = (n00 > 0) && (n00 + 256 * (nrows - 1) + KR_CH * nch < np);   // workgroup-uniform
  float2 greg[NR];
  auto gload = [&](int c) {
    if (interior) {
#pragma unroll
      for (int r = 0; r < NR; r++) greg[r] = fb[n00 + 256 * min(lr + RPR * r, nrows - 1) + lk + KR_CH * c];
    } else {
#pragma unroll
      for (int r = 0; r < NR; r++) {
        const int n = n00 + 256 * min(lr + RPR * r, nrows - 1) + lk + KR_CH * c;
        const bool inr = (n > 0) && (n < np);           // cc:205, sample 0 excluded
        const float2 v = fb[min(max(n, 0), np - 1)];
        greg[r] = inr ? v : make_float2(0.0f, 0.0f);
      }
    }
  };
  auto gstore = [&](int buf) {
#pragma unroll
    for (int r = 0; r < NR; r++)
      if (lr + RPR * r < nrows) *reinterpret_cast<float2 *>(&stage[buf][(lr + RPR * r) * KR_ROWDW + 2 * lk]) = greg[r];
  };
  // phasor-table slices of a chunk (frequency stages): element e = (hypothesis e / 64, tone (e / 16) % 4, step pair e % 16)
  constexpr int TR = G::LAGS ? 1 : (5 * 64 + NT - 1) / NT;
  const float4 *tab4 = reinterpret_cast<const float4 *>(ptab + ((size_t)slot * kPtabPerSlot + (tabled ? R.tab : 0)) * kPtabFloat2);
  float4 treg[TR];
  auto tload = [&](int c) {
    if (G::LAGS || !tabled) return;
#pragma unroll
    for (int r = 0; r < TR; r++) {
      const int e = min(tid + NT * r, 5 * 64 - 1);
      treg[r] = tab4[(e >> 6) * (kPtabFloat2 / 2) + ((e >> 4) & 3) * 128 + (KR_CH / 2) * c + (e & 15)];
    }
  };
  auto tstore = [&](int buf) {
    if (G::LAGS || !tabled) return;
#pragma unroll
    for (int r = 0; r < TR; r++) {
      const int e = tid + NT * r;
      if (e < 5 * 64) ptl[((buf * 4 + ((e >> 4) & 3)) * (KR_CH / 2) + (e & 15)) * 5 + (e >> 6)] = treg[r];
    }
  };
  // a unit that opens a chunk: the chunk before it is done with -- publish the next one (loaded during it) and start
  // loading the one after.  One barrier per chunk, executed by every wavefront whatever its hypotheses.
  auto chunk_turn = [&](int u) {
#ifdef KR_STAMPS
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
#endif
    if (u) { gstore((u >> 2) & 1); tstore((u >> 2) & 1); __syncthreads(); }
    gload(min((u >> 2) + 1, nch - 1)); tload(min((u >> 2) + 1, nch - 1));
    // Issue priority, rotated chunk by chunk over the three workgroups that share a CU (workgroups b, b + G/3,
    // b + 2G/3 of a grid of G land on the same CU: tools/kr_wgmap.py).  The arbiter serves equal priorities oldest
    // first, so without this the first of the three runs ahead and the last one finishes 15 us after it -- alone on
    // its SIMDs, at two wavefronts each.  Taking turns keeps the three in step and the SIMDs full to the end.
#ifndef KR_NO_PRIO_ROTATION
    switch (((unsigned)(u >> 2) + prio_class) % 3u) {   // (the instruction takes an immediate; uniform branch)
      case 0: __builtin_amdgcn_s_setprio(0); break;
      case 1: __builtin_amdgcn_s_setprio(1); break;
      default: __builtin_amdgcn_s_setprio(2); break;
    }
#endif
#ifdef KR_STAMPS
    st_bar += __builtin_amdgcn_s_memtime() - t0;
#endif
  };

  // idle lanes shadow row 0 (their results are discarded)
  const int myrow = (mine || virt) ? lane : 0;
  float inp[NH], quad[NH];
#pragma unroll
  for (int h = 0; h < NH; h++) { inp[h] = 0.0f; quad[h] = 0.0f; }

  // ---- table walk: one block = (hypothesis, 8 samples) = 64 multiply-add instructions, phasors read from the LDS
  // image (the same address in every lane: broadcast), two steps per ds_read_b128.  A scheduling barrier closes every
  // unit: left alone the scheduler hoists a whole chunk's LDS reads above its arithmetic (170 VGPRs and spills).
  auto walk_tab = [&](auto sub_tag, auto known_tag) {
    constexpr uint32_t HM = G::subset(HS, decltype(sub_tag)::value) & (decltype(known_tag)::value ? PAT_KNOWN : PAT_FULL);
    constexpr uint32_t PAT = decltype(known_tag)::value ? PAT_KNOWN : PAT_FULL;
    constexpr int NU = 4 * ((256 + 8 * G::dk8(31 - __builtin_clz(PAT)) + KR_CH - 1) / KR_CH);
    constexpr rows_phases PL = rows_make_phases<KIND>(HM, NU);
    auto unit = [&](auto am_tag, int u) {
      constexpr uint32_t AM = decltype(am_tag)::value;
      const float *rowp = &stage[(u >> 2) & 1][myrow * KR_ROWDW + 16 * (u & 3)];
      float4 xv[4];
#pragma unroll
      for (int q = 0; q < 4; q++) xv[q] = *reinterpret_cast<const float4 *>(rowp + 4 * q);
#pragma unroll
      for (int h = 0; h < NH; h++) {
        if (!((AM >> h) & 1u)) continue;
        const float4 *pp = G::LAGS ? &ptl[tone * 128 + 4 * (u - G::dk8(h))]
                                   : &ptl[((((u >> 2) & 1) * 4 + tone) * (KR_CH / 2) + 4 * (u & 3)) * 5 + h];
#pragma unroll
        for (int q = 0; q < 4; q++) {
          const float4 ph = pp[G::LAGS ? q : 5 * q];   // (c, s) of two steps
          k4_mac<FAST>(inp[h], quad[h], xv[q].x, xv[q].y, ph.x, ph.y);       // cc:206-207
          k4_mac<FAST>(inp[h], quad[h], xv[q].z, xv[q].w, ph.z, ph.w);
        }
      }
      __builtin_amdgcn_sched_barrier(0);
    };
    auto phase = [&](auto p_tag) {
      constexpr int P = decltype(p_tag)::value;
      constexpr uint32_t AM = PL.p[P].am;
      constexpr int U0 = PL.p[P].u0, N = PL.p[P].n;
#pragma unroll 1
      for (int u = U0; u < U0 + N; u += 2) {
        if ((u & 3) == 0) chunk_turn(u);
#ifndef KR_NO_ARITH   // (timing experiment: the loader, the barriers and the stores alone)
        if constexpr (AM != 0) {
          unit(std::integral_constant<uint32_t, AM>{}, u);
          unit(std::integral_constant<uint32_t, AM>{}, u + 1);
        }
#endif
      }
    };
    [&]<int... P>(std::integer_sequence<int, P...>) { (phase(std::integral_constant<int, P>{}), ...); }(std::make_integer_sequence<int, PL.n>{});
  };

  // ---- per-lane recurrences (a drifting linear model: the frequency depends on the symbol), and stage 5 ----------
  auto walk_lane = [&](auto sub_tag, auto tab_tag) {
    constexpr bool TAB = decltype(tab_tag)::value;
    constexpr uint32_t SM = G::subset(HS, decltype(sub_tag)::value);
    const uint32_t wmask = mask & SM;                   // uniform
    // phasor steps from the row's symbol frequency (cc:170-189); lag stages share the step
    constexpr int NST = TAB ? 1 : (G::LAGS ? 1 : NH);
    constexpr int NPH = TAB ? 1 : NH;
    float pc[NPH], psn[NPH], cd[NST], sd[NST];
    if (!TAB) {
      const int own_i = min(sym, UWSPR_NSYM - 1);
#pragma unroll
      for (int h = 0; h < NST; h++) {
        if (!G::LAGS && !((SM >> h) & 1u)) { cd[h] = 1.0f; sd[h] = 0.0f; continue; }
        const dev_hyp hy = hyps[(size_t)slot * NH + G::out_index(G::LAGS ? 0 : h)];
        k4_tone_step(k4_symbol_freq(hy.m_type, hy.f0, hy.drift, hy.slmc, own_i), tone, cd[h], sd[h]);
      }
#pragma unroll
      for (int h = 0; h < NPH; h++) { pc[h] = 1.0f; psn[h] = 0.0f; }
    }
    const KR_CONST float *tabw = (const KR_CONST float *)(ptab + ((size_t)slot * kPtabPerSlot + (TAB ? R.tab : 0)) * kPtabFloat2 + tone * 256);
    for (int u = 0; u < 4 * nch; u++) {
      if ((u & 3) == 0) chunk_turn(u);
      const float *rowp = &stage[(u >> 2) & 1][myrow * KR_ROWDW + 16 * (u & 3)];
      float4 xv[4];
#pragma unroll
      for (int q = 0; q < 4; q++) xv[q] = *reinterpret_cast<const float4 *>(rowp + 4 * q);
#pragma unroll
      for (int h = 0; h < NH; h++) {
        if (!((SM >> h) & 1u)) continue;            // compile time: not this subset's
        if (!((wmask >> h) & 1u)) continue;         // uniform: known / unused
        const int k0 = 8 * u - 8 * G::dk8(h);       // uniform: the hypothesis' step at this unit
        if (G::LAGS && (k0 < 0 || k0 > 248)) continue;
        if (TAB) {
          const kr_f16 ph = *(const KR_CONST kr_f16 *)(tabw + (G::LAGS ? 0 : h * 2 * kPtabFloat2) + 2 * k0);
#pragma unroll
          for (int q = 0; q < 4; q++) {
            k4_mac<FAST>(inp[h], quad[h], xv[q].x, xv[q].y, ph[4 * q], ph[4 * q + 1]);       // cc:206-207
            k4_mac<FAST>(inp[h], quad[h], xv[q].z, xv[q].w, ph[4 * q + 2], ph[4 * q + 3]);
          }
        } else {
          const int hs = G::LAGS ? 0 : h;
#pragma unroll
          for (int q = 0; q < 4; q++) {
            k4_mac<FAST>(inp[h], quad[h], xv[q].x, xv[q].y, pc[h], psn[h]);                  // cc:206-207
            k4_rot<FAST>(pc[h], psn[h], cd[hs], sd[hs]);                                     // cc:193-195
            k4_mac<FAST>(inp[h], quad[h], xv[q].z, xv[q].w, pc[h], psn[h]);
            k4_rot<FAST>(pc[h], psn[h], cd[hs], sd[hs]);
          }
        }
      }
    }
  };

  gload(0); tload(0);
  if (G::LAGS && tabled && KIND != UWSPR_ROWS_S5) {   // the lag stages' one table, whole
    for (int e = tid; e < 4 * 128; e += NT) ptl[e] = tab4[e];
  }
  gstore(0); tstore(0);
  __syncthreads();
#ifdef KR_STAMPS
  st_pro = __builtin_amdgcn_s_memtime() - st_c0;
#endif
  // every wavefront of the workgroup executes the same number of barriers (one per chunk)
  auto go = [&](auto sub_tag) {
    if (!tabled) walk_lane(sub_tag, std::false_type{});
    else if constexpr (KIND == UWSPR_ROWS_S5) walk_lane(sub_tag, std::true_type{});   // (17 lags: the phased walk would be 50 KB of code)
    else if (known) walk_tab(sub_tag, std::true_type{});
    else walk_tab(sub_tag, std::false_type{});
  };
  if constexpr (HS == 1) go(std::integral_constant<int, 0>{});
  else if constexpr (HS == 2) { if (hsub == 0) go(std::integral_constant<int, 0>{}); else go(std::integral_constant<int, 1>{}); }
  else {
    switch (hsub) {
      case 0: go(std::integral_constant<int, 0>{}); break;
      case 1: go(std::integral_constant<int, 1>{}); break;
      case 2: go(std::integral_constant<int, HS >= 3 ? 2 : 0>{}); break;
      default: go(std::integral_constant<int, HS >= 4 ? 3 : 0>{}); break;
    }
  }

#ifdef KR_STAMPS
  st_end = __builtin_amdgcn_s_memtime();
#endif
  // tone magnitudes (cc:211)
  if (mine || virt) {
    const uint32_t smask = mask & (HS == 1 ? 0xffffffffu : HS == 2 ? G::subset(HS, hsub ? 1 : 0)
                                   : G::subset(HS, hsub == 0 ? 0 : hsub == 1 ? 1 : hsub == 2 ? (HS >= 3 ? 2 : 0) : (HS >= 4 ? 3 : 0)));
#pragma unroll
    for (int h = 0; h < NH; h++) {
      if (!((smask >> h) & 1u)) continue;
      const float pj = ieee_sqrtf(inp[h] * inp[h] + quad[h] * quad[h]);
      if (!virt) p_out[(((size_t)slot * NH + G::out_index(h)) * UWSPR_NSYM + sym) * 4 + tone] = pj;
      // the wrap: (lag 0, symbol i) is also (lag 4, symbol i - 1)
      if (KIND == UWSPR_ROWS_S0 && h == 0 && wrap && sym >= 1 && ((R.mask >> 4) & 1u))
        p_out[(((size_t)slot * NH + 4) * UWSPR_NSYM + sym - 1) * 4 + tone] = pj;
    }
  }
#ifdef KR_STAMPS
  if (KIND == KR_STAMP_KIND && lane == 0) {
    const unsigned gw = wg * (4 * HS) + wv;
    if (gw < (unsigned)KR_STAMP_WAVES) {
      unsigned long long *o = &g_kr_stamps[(size_t)gw * 8];
      o[0] = st_c0; o[1] = __builtin_amdgcn_s_memtime(); o[2] = st_r0; o[3] = __builtin_amdgcn_s_memrealtime();
      o[4] = blockIdx.x; o[5] = st_end - st_c0 - st_pro - st_bar; o[6] = st_bar;
      o[7] = (unsigned long long)__builtin_amdgcn_s_getreg((31 << 11) | 4) | ((unsigned long long)__builtin_amdgcn_s_getreg((31 << 11) | 20) << 32);
    }
  }
#endif
}

template <int KIND, int HS>
static void launch_rows_t(uwspr_ctx *c, prof_scope &ps, const float *frames, int B, const dev_hyp *hyps, int nslots, float4 *p) {
  const dim3 grid(3u * (unsigned)nslots), blk(256 * HS);
  const float2 *pt = c->use_ptab ? c->d_ptab : nullptr;
  if (c->fast_now && KIND != UWSPR_ROWS_S5)
    launch_timed(c, ps, (k4_rows<KIND, HS, (KIND != UWSPR_ROWS_S5)>), grid, blk, 0, (const float2 *)frames, c->fstride, c->np, B,
                 c->d_rows, hyps, nslots, pt, (float *)p);
  else
    launch_timed(c, ps, (k4_rows<KIND, HS, false>), grid, blk, 0, (const float2 *)frames, c->fstride, c->np, B,
                 c->d_rows, hyps, nslots, pt, (float *)p);
}

// stage `kind` for nslots candidate slots: rows[slot] (written by the schedule kernels) says what to compute
void launch_tonecorr_rows(uwspr_ctx *c, const float *frames, int B, int kind, const dev_hyp *hyps, int nslots,
                          int64_t nhyps, float4 *p) {
  if (nslots <= 0) return;
  prof_scope ps(c, UWSPR_K_TONECORR, nhyps, true);
  const int hs = c->rows_hs[kind];
  switch (kind) {
    case UWSPR_ROWS_S0: if (hs == 2) launch_rows_t<UWSPR_ROWS_S0, 2>(c, ps, frames, B, hyps, nslots, p); else launch_rows_t<UWSPR_ROWS_S0, 1>(c, ps, frames, B, hyps, nslots, p); break;
    case UWSPR_ROWS_S1: if (hs == 2) launch_rows_t<UWSPR_ROWS_S1, 2>(c, ps, frames, B, hyps, nslots, p); else launch_rows_t<UWSPR_ROWS_S1, 1>(c, ps, frames, B, hyps, nslots, p); break;
    case UWSPR_ROWS_S3: if (hs == 2) launch_rows_t<UWSPR_ROWS_S3, 2>(c, ps, frames, B, hyps, nslots, p); else launch_rows_t<UWSPR_ROWS_S3, 1>(c, ps, frames, B, hyps, nslots, p); break;
    case UWSPR_ROWS_S4: if (hs == 2) launch_rows_t<UWSPR_ROWS_S4, 2>(c, ps, frames, B, hyps, nslots, p); else launch_rows_t<UWSPR_ROWS_S4, 1>(c, ps, frames, B, hyps, nslots, p); break;
    default:
      if (hs == 4) launch_rows_t<UWSPR_ROWS_S5, 4>(c, ps, frames, B, hyps, nslots, p);
      else if (hs == 1) launch_rows_t<UWSPR_ROWS_S5, 1>(c, ps, frames, B, hyps, nslots, p);
      else launch_rows_t<UWSPR_ROWS_S5, 2>(c, ps, frames, B, hyps, nslots, p);
      break;
  }
}

}  // namespace uwspr

#ifdef KR_STAMPS
extern "C" int uwspr_debug_kr_stamps(unsigned long long *out, int nwaves) {
  if (nwaves > uwspr::KR_STAMP_WAVES) nwaves = uwspr::KR_STAMP_WAVES;
  (void)hipDeviceSynchronize();
  return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(uwspr::g_kr_stamps), (size_t)nwaves * 64);
}
#endif
