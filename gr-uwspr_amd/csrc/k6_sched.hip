// K6 -- the whole refinement schedule of a candidate (S0..S5) in ONE workgroup.
//
// Reference: sync_and_demodulate_impl::demodulate, lib/sync_and_demodulate_impl.cc:403-482
// (the six dependent calls of sync_and_demodulate(), cc:126-256, per candidate).
//
// The candidates of a batch are independent (cc:389 loops over them) and every
// stage of a candidate needs all of the previous stage's 162 symbols, so the
// natural unit is: one 16-wavefront workgroup = one candidate, running
// S0 -> S5 back to back with no kernel boundary, no tone magnitudes in HBM and
// no cross-workgroup traffic.  (The staged form -- k4_* + k5_fold_step, 17
// launches per batch -- stays as option "sched" = 0; both give the same bytes.)
//
// Mapping.  A lane owns a symbol window ("row") and ONE tone, and accumulates
// inp/quad for several hypotheses of the stage against it, every accumulator
// seeing exactly the reference's sequence of binary32 operations (cc:206-207: no
// FMA, no tree).  Sixteen wavefronts = four per SIMD (three per SIMD issue at 0.38
// instructions per cycle on gfx950, two or four at 0.5: tools/op_probe.hip):
// wavefront w has tone w & 3 and hypothesis slot w >> 2; a lane owns THREE rows (symbols lane,
// 54 + lane, 108 + lane) and every slot carries a quarter of the stage's hypotheses
// (k6_geom::slot_mask), so that each phasor run is fetched by exactly one wavefront.
//  * Samples: the stage streams the rows [L0 + 256 i, L0 + 256 i + 256 + span)
//    through a double-buffered LDS image, 16 samples per row and chunk, loaded
//    cooperatively with coalesced 8-byte loads (cc:205's n > 0 && n < np test is
//    applied by the loader: a skipped sample is a zero, which leaves inp/quad
//    unchanged).  A lane reads its samples once per chunk and uses them for every
//    hypothesis: a hypothesis whose lag is L0 + D sees stream position a as its
//    sample k = a - D ("sample-major" order), so lag sweeps cost no extra loads --
//    S5's 17 lags are one pass over 384 samples.
//  * Phasors.  When the per-symbol frequency does not depend on the symbol
//    (drift == 0 or the straight-line model with t = 0: the reference's `fplast`
//    cache hits for the same reason, cc:185) the sequence c[k], s[k] of cc:186-199
//    is one table per (frequency, tone): 20 lanes run the 256-step recurrences
//    (cc:193-195) once per candidate and table set, the table goes to HBM/L2, and
//    the correlating wavefronts -- whose tone is wave-uniform -- fetch it with
//    SCALAR loads (s_load_dwordx16 = 8 steps) and use it as SGPR operands: the walk
//    is 8 VALU operations per sample and hypothesis, nothing else.  Table set A
//    (f1 + {-2..2} 0.25 Hz) serves S0 and S1, set B (f1 + {-2..2} 0.05 Hz) S3, S4, S5.
//    With a per-symbol frequency (a drifting linear model: always in S2) every lane
//    runs its own recurrences, 14 operations per sample and hypothesis.
//  * Fold (cc:213-226, 240-254) from LDS: the per-symbol terms by all threads, then
//    the order-sensitive running sums (totp, ss, fsum, f2sum, rms) one hypothesis per
//    LANE, one kind of sum per wavefront; stage transitions (cc:227-231, 416-452) by
//    one thread; the stage winner's tone magnitudes are kept (2.6 KB) so that the
//    hypothesis a later stage repeats -- the middle one of S1/S3/S4 and the first
//    jiggered shift of S5 -- is not computed again.
//  * S0's last lag is one symbol after its first: (last lag, symbol i) is (first
//    lag, symbol i+1) when the frequency does not depend on the symbol; a 163rd
//    "virtual" row on an otherwise idle lane supplies (last lag, symbol 161).
#include <stdlib.h>

#include <type_traits>

#include "uwspr_internal.h"

#pragma clang fp contract(off)

namespace uwspr {

// K6_SLOTS: hypothesis slots per tone.  4 = the library's form (16 wavefronts: one workgroup fills a CU).  2 = round 6's
// EXPERIMENT build only (UWSPR_EXTRA_HIPFLAGS=-DK6_SLOTS=2, its own library file; with option sched_grid = 512): 8 wavefronts
// per candidate, two hypotheses per loaded sample in S0-S4, two workgroups per CU -- review item 3, measured in
// profiles/r06_k6_slots_ab.txt.  Everything the default build compiles is unchanged by the macro.
#ifndef K6_SLOTS
#define K6_SLOTS 4
#endif
static_assert(K6_SLOTS == 4 || K6_SLOTS == 2, "K6_SLOTS");
constexpr int K6_WAVES = 4 * K6_SLOTS;
constexpr int K6_THREADS = 64 * K6_WAVES;
constexpr int K6_LROWS = K6_THREADS / 16;     // rows one loader round covers: 64 / 32

constexpr int K6_TROWS = 54;                  // rows per row-role wavefront: 162 = 3 x 54
constexpr int K6_MAXROWS = UWSPR_NSYM + 1;    // + the virtual row of the S0 wrap
constexpr int K6_ROWDW = 36;                  // dwords per staged row: 16 samples x 8 B + 16 B pad
constexpr int K6_NTAB = 5;                    // frequencies per table set
constexpr int K6_TABSET = K6_NTAB * 4 * 512;  // floats per table set: [freq][tone][256](c, s)
constexpr int K6_PSLAB = K6_MAXROWS * 4;      // floats per hypothesis in the p image
constexpr int K6_FOLDH = 9;                   // hypotheses folded per round
constexpr int K6_CMS = UWSPR_NSYM + 2;        // stride of the per-hypothesis scratch rows
constexpr int K6_NLD = (K6_MAXROWS + K6_LROWS - 1) / K6_LROWS;   // loader rounds per chunk: 3 / 6

constexpr double kTwoPiDt6 = 2.0 * 3.14159265358979323846 * (double)(float)(1.0 / 375.0);  // cc:146,188

__device__ __constant__ uint32_t kPr3_6[6] = UWSPR_PR3_WORDS;
__device__ __forceinline__ bool pr3_6(int i) { return (kPr3_6[i >> 5] >> (i & 31)) & 1u; }

typedef float f16v __attribute__((ext_vector_type(16)));
#define K6_CONST __attribute__((address_space(4)))
// The passes and folds are real (not inlined) functions -- each gets its own register allocation
// instead of one 40 000-instruction body -- so LDS pointers cross the call as address-space-3
// pointers and stay ds_read / ds_write inside.
#define K6_LDS __attribute__((address_space(3)))
#define K6_GLOBAL __attribute__((address_space(1)))
typedef float v4f __attribute__((ext_vector_type(4)));
typedef float v2f __attribute__((ext_vector_type(2)));
typedef K6_LDS float lds_f;
typedef K6_LDS v4f lds_f4;
typedef K6_LDS v2f lds_f2;

__device__ __forceinline__ int uni(int v) { return __builtin_amdgcn_readfirstlane(v); }
template <typename T>
__device__ __forceinline__ T *uni_ptr(T *p) {   // a pointer that crossed a call: back into SGPRs
  const unsigned long long v = (unsigned long long)p;
  const unsigned lo = (unsigned)uni((int)(unsigned)v), hi = (unsigned)uni((int)(unsigned)(v >> 32));
  return (T *)(((unsigned long long)hi << 32) | lo);
}

// slmFrequencyDrift(m_nl, cf, t = 0), lib/slm.cc:36-73, in binary64 like the reference (`t` is
// read uninitialised at sync_and_demodulate_impl.cc:177-180; every observed build behaves as t = 0)
static __device__ float k6_slm_drift_t0(double V1, double V2, int p1, int p2, float cf) {
  const double q1 = V1 * 0.0 + (double)p1, q2 = V2 * 0.0 + (double)p2;
  const float sign = (float)(((q1 * V1 + q2 * V2) > 0) * 2 - 1);
  const double num = fabs(V1 * q1 + V2 * q2);
  const double den = sqrt(q1 * q1 + q2 * q2);
  if (den == 0) return 0.0f;
  return (float)((double)(-sign) * num / den * (double)cf / (double)1500.0f);
}

struct k6_args {
  const float2 *frames; int fstride; int np; int nframes;
  const uwspr_candidate *cands; const int32_t *npk; int cand_stride; int per_frame; int nslots;
  float cf; int reuse;
  int njig;                     // mode-2 tries to produce: 17, or fewer (lazy S5)
  float *tabs;                  // [gridDim.x][2][K6_TABSET]
  int *counter;                 // slot queue head
  uwspr_demod_out *out;
  cand_state *state;            // [nslots] final state (resume / diagnostics)
  float *pwin;                  // [nslots][162][4] winner magnitudes, kept for uwspr_demod_resume, or null
  const uint8_t *resume;        // resume pass: [nslots] nonzero = produce the remaining tries of that slot (state, pwin are inputs)
  unsigned long long *stamps;   // diagnostics: [nslots][64] wall-clock ticks at the phase boundaries, or null
};

// ---- one pass over the sample stream -----------------------------------------------------------
// The six stages as compile-time geometry: hypothesis h of stage KIND has lag L0 + 8 dk8(h) and
// either phasor table tq(h) of the stage's table set (TAB) or, on the per-lane path, frequency
// fc + (h - 2) fstep and drift drp (S2: drp for h = 0, drm for h = 1, both at fc).
enum { K6_S0 = 0, K6_S1 = 1, K6_S2 = 2, K6_S3 = 3, K6_S4 = 4, K6_S5 = 5 };
template <int KIND> struct k6_geom {
  static constexpr int HMAX = KIND == K6_S2 ? 2 : KIND == K6_S5 ? UWSPR_NJIG : 5;
  static constexpr bool LAGS = KIND == K6_S0 || KIND == K6_S3 || KIND == K6_S5;   // one frequency, several lags
  // Work split: wavefront w has tone w & 3 and hypothesis slot w >> 2; slot_mask(s) = the stage's
  // hypotheses of slot s, row_mask(s, h) = which of the lane's three rows (symbols lane, 54 + lane,
  // 108 + lane) it walks for them.  A stage of four computed hypotheses is one per slot; S5's
  // sixteen are four per slot, dealt so that every slot has an early, two middle and a late lag
  // (a lag is walked only while the stream position is inside its window); S2's two are split by rows so that three slots carry two
  // (row, hypothesis) pairs each.  Hypotheses that are usually known (the middle one of S1/S3/S4,
  // try 0 of S5, the wrapped last lag of S0) ride on a slot as an extra.
  __host__ __device__ static constexpr uint32_t slot_mask(int s) {
#if K6_SLOTS == 2
    // two slots: an early and a late lag each (S0), the halves of the frequency row (S1 / S4 / S3), S5's 8 + 9
    return KIND == K6_S0 ? (s == 0 ? 0x09u : 0x16u)
         : KIND == K6_S2 ? (s == 0 ? 0x1u : 0x2u)
         : KIND == K6_S5 ? (s == 0 ? 0x0cccdu : 0x13332u)
         : (s == 0 ? 0x07u : 0x18u);
#endif
    return KIND == K6_S0 ? (s == 3 ? 0x18u : 1u << s)
         : KIND == K6_S2 ? (s == 0 ? 0x1u : s == 1 ? 0x2u : s == 2 ? 0x3u : 0u)
         : KIND == K6_S5 ? (s == 0 ? 0x08485u : s == 1 ? 0x03030u : s == 2 ? 0x04848u : 0x10302u)
         : (s == 0 ? 0x05u : s == 1 ? 0x02u : s == 2 ? 0x08u : 0x10u);
  }
  __host__ __device__ static constexpr uint32_t row_mask(int s, int h) {
    if (K6_SLOTS == 2) return 0x7u;
    return KIND == K6_S2 ? (s == 2 ? 0x4u : 0x3u) : 0x7u;
  }
  __host__ __device__ static constexpr int dk8(int h) {
    return KIND == K6_S0 ? 8 * h                                            // shift1 - 128 + 64 h  (cc:409-411)
         : KIND == K6_S3 ? 2 * h                                            // shift1 - 32 + 16 h   (cc:444)
         : KIND == K6_S5 ? 8 + ((h & 1) ? -((h + 1) / 2) : (h + 1) / 2)     // shift1 + 8 ii(idt)   (cc:459-463)
         : 0;
  }
};

// index of the n-th set bit of M (compile time)
template <uint32_t M>
__host__ __device__ constexpr int nth_bit(int n) {
  int c = 0;
  for (int b = 0; b < 32; b++)
    if ((M >> b) & 1u) { if (c == n) return b; c++; }
  return 0;
}

// bit h of `mask` = hypothesis h is computed; tq0 = table of a lag sweep (LAGS); S1/S4 use table h.
template <int KIND, bool TAB>
__device__ __noinline__ void k6_pass(const float2 *__restrict__ fb, int np, int L0, int nrows,
                                     int nchunks, uint32_t mask, int tq0, float fc, float fstep,
                                     float drp, float drm, int m_type, float slmc,
                                     const float *tabset, lds_f *stage) {
  using G = k6_geom<KIND>;
  constexpr int HMAX = G::HMAX;
  constexpr bool SHARED = G::LAGS;
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wv = uni(tid >> 6);
  // arguments are workgroup-uniform; across the call they arrive in VGPRs
  // (and as generic pointers: the frame is read with global_load, not flat_load -- a flat load
  // counts as an LDS access too and would tie every LDS / scalar wait to HBM latency)
  const K6_GLOBAL v2f *fbg = (const K6_GLOBAL v2f *)uni_ptr(fb);
  tabset = uni_ptr(tabset);
  np = uni(np); L0 = uni(L0); nrows = uni(nrows); nchunks = uni(nchunks); mask = (uint32_t)uni((int)mask);
  tq0 = uni(tq0); m_type = uni(m_type);

  // ---- loader: element e = tid + 1024 n -> row tid / 16 + 64 n, sample tid % 16
  const int lr = tid >> 4, lj = tid & 15;
  const int sbase = lr * K6_ROWDW + 2 * lj;
  const bool interior = (L0 > 0) && (L0 + 256 * (nrows - 1) + 16 * nchunks < np);   // workgroup-uniform
  float2 greg[K6_NLD];
  auto gload = [&](int c) {
    if (interior) {
#pragma unroll
      for (int n = 0; n < K6_NLD; n++) {
        const int r = min(lr + K6_LROWS * n, nrows - 1);
        const v2f v = fbg[L0 + 256 * r + lj + 16 * c];
        greg[n] = make_float2(v.x, v.y);
      }
    } else {
#pragma unroll
      for (int n = 0; n < K6_NLD; n++) {
        const int r = min(lr + K6_LROWS * n, nrows - 1);
        const int ns = L0 + 256 * r + lj + 16 * c;
        const bool inr = (ns > 0) && (ns < np);                 // cc:205, sample 0 excluded
        const v2f v = fbg[min(max(ns, 0), np - 1)];
        greg[n] = inr ? make_float2(v.x, v.y) : make_float2(0.0f, 0.0f);
      }
    }
  };
  auto gstore = [&](int buf) {
#pragma unroll
    for (int n = 0; n < K6_NLD; n++)
      if (lr + K6_LROWS * n < nrows)
        *(lds_f2 *)(&stage[buf * K6_MAXROWS * K6_ROWDW + sbase + K6_LROWS * n * K6_ROWDW]) = v2f{greg[n].x, greg[n].y};
  };

  // ---- the walk of one hypothesis slot: three rows per lane, the hypotheses of slot_mask(SLOT)
  auto walk = [&](auto slot_tag) {
    constexpr int SLOT = decltype(slot_tag)::value;
    constexpr uint32_t SM = G::slot_mask(SLOT);
    constexpr int NR = 3;
    const int tone = wv & 3;
    // lanes 0..53: symbols lane, 54 + lane, 108 + lane; lane 54: the virtual row 162 (nrows == 163)
    const bool virt = (lane == K6_TROWS) && (nrows > UWSPR_NSYM);
    const int row0 = lane < K6_TROWS ? lane : 0;          // idle lanes shadow real rows (results discarded)

    float inp[NR][HMAX], quad[NR][HMAX];
#pragma unroll
    for (int r = 0; r < NR; r++)
#pragma unroll
      for (int h = 0; h < HMAX; h++) { inp[r][h] = 0.0f; quad[r][h] = 0.0f; }

    // per-lane path: phasor steps from the row's symbol frequency (cc:170-189)
    constexpr int NPH = TAB ? 1 : HMAX;
    constexpr int NST = TAB ? 1 : (SHARED ? 1 : HMAX);
    float pc[NR][NPH], psn[NR][NPH], cd[NR][NST], sd[NR][NST];
    if (!TAB) {
      const float delta = ((float)tone - 1.5f) * 1.46484375f;          // cc:148
#pragma unroll
      for (int r = 0; r < NR; r++) {
        const int own_i = min(row0 + K6_TROWS * r, UWSPR_NSYM - 1);
#pragma unroll
        for (int h = 0; h < NST; h++) {
          if (!SHARED && !((SM >> h) & 1u)) continue;
          if (!SHARED && !((G::row_mask(SLOT, h) >> r) & 1u)) continue;
          const float f0 = (KIND == K6_S2 || SHARED) ? fc : fc + (float)(h - 2) * fstep;   // cc:164
          const float dr = KIND == K6_S2 ? (h == 0 ? drp : drm) : drp;
          float fp;
          if (m_type == UWSPR_LINEAR)
            fp = (float)((double)f0 + ((double)dr / 2.0) * ((double)(float)own_i - 81.0) / 81.0);  // cc:173
          else
            fp = f0 + slmc;                                               // cc:179 (t = 0)
          double sn, cs;
          sincos(kTwoPiDt6 * (double)(fp + delta), &sn, &cs);             // cc:188-189
          cd[r][h] = (float)cs; sd[r][h] = (float)sn;
        }
#pragma unroll
        for (int h = 0; h < NPH; h++) { pc[r][h] = 1.0f; psn[r][h] = 0.0f; }
      }
    }
    const K6_CONST float *tabw = (const K6_CONST float *)tabset + tone * 512 + (SHARED ? tq0 * 2048 : 0);
    const uint32_t wmask = mask & SM;
    // rows this slot reads at all (S2's third slot walks only the last row of each lane)
    constexpr uint32_t ROWS = (KIND == K6_S2 && K6_SLOTS == 4) ? (SLOT == 2 ? 0x4u : SLOT == 3 ? 0u : 0x3u) : 0x7u;

    for (int c = 0; c < nchunks; c++) {
      gload(min(c + 1, nchunks - 1));           // in flight during the arithmetic
      const lds_f *rowp = &stage[(c & 1) * K6_MAXROWS * K6_ROWDW + (virt ? 0 : row0) * K6_ROWDW];
      // lane 54 (virtual row): its third row is row 162 = 108 + 54
      for (int half = 0; half < 2; half++) {   // a real loop: halves the code
        v4f xv[NR][4];
#pragma unroll
        for (int r = 0; r < NR; r++) {
          if (!((ROWS >> r) & 1u)) continue;
#pragma unroll
          for (int j = 0; j < 4; j++)
            xv[r][j] = *(const lds_f4 *)(rowp + (r * K6_TROWS + (virt ? K6_TROWS : 0)) * K6_ROWDW + 16 * half + 4 * j);
        }
#pragma unroll
        for (int h = 0; h < HMAX; h++) {
          if (!((SM >> h) & 1u)) continue;                   // compile time: not this slot's
          if (!((wmask >> h) & 1u)) continue;                // uniform
          const int k0 = 16 * c + 8 * half - 8 * G::dk8(h);  // uniform: this hypothesis' first step
          if (k0 < 0 || k0 > 248) continue;
          if (TAB) {
            // 8 steps of (c, s) by one scalar load, used for three rows.  (The scalar path moves
            // ~4 GB/s per CU when it streams -- tools/smem_probe.hip -- so every run is fetched
            // by exactly one wavefront.)
            const f16v ph = *(const K6_CONST f16v *)(tabw + (SHARED ? 0 : h * 2048) + 2 * k0);
#pragma unroll
            for (int r = 0; r < NR; r++) {
              if (!((G::row_mask(SLOT, h) >> r) & 1u)) continue;
#pragma unroll
              for (int j = 0; j < 4; j++) {
                const v4f x = xv[r][j];
                inp[r][h] = (inp[r][h] + x.x * ph[4 * j]) + x.y * ph[4 * j + 1];        // cc:206
                quad[r][h] = (quad[r][h] - x.x * ph[4 * j + 1]) + x.y * ph[4 * j];      // cc:207
                inp[r][h] = (inp[r][h] + x.z * ph[4 * j + 2]) + x.w * ph[4 * j + 3];
                quad[r][h] = (quad[r][h] - x.z * ph[4 * j + 3]) + x.w * ph[4 * j + 2];
              }
            }
          } else {
            const int hs = SHARED ? 0 : h;
#pragma unroll
            for (int r = 0; r < NR; r++) {
              if (!((G::row_mask(SLOT, h) >> r) & 1u)) continue;
#pragma unroll
              for (int j = 0; j < 4; j++) {
                const v4f x = xv[r][j];
#pragma unroll
                for (int e = 0; e < 2; e++) {
                  const float xx = e ? x.z : x.x, xy = e ? x.w : x.y;
                  inp[r][h] = (inp[r][h] + xx * pc[r][h]) + xy * psn[r][h];             // cc:206
                  quad[r][h] = (quad[r][h] - xx * psn[r][h]) + xy * pc[r][h];           // cc:207
                  const float nc = pc[r][h] * cd[r][hs] - psn[r][h] * sd[r][hs];        // cc:193-195
                  const float ns = pc[r][h] * sd[r][hs] + psn[r][h] * cd[r][hs];
                  pc[r][h] = nc; psn[r][h] = ns;
                }
              }
            }
          }
        }
      }
      gstore((c + 1) & 1);
      __syncthreads();
    }
    // tone magnitudes (cc:211) into the p image, which overlays the (now dead) staging buffers
    if (lane < K6_TROWS || virt) {
#pragma unroll
      for (int r = 0; r < NR; r++) {
        if (virt && r != 2) continue;
        const int row = virt ? UWSPR_NSYM : row0 + K6_TROWS * r;
#pragma unroll
        for (int h = 0; h < HMAX; h++)
          if (((SM >> h) & 1u) && ((G::row_mask(SLOT, h) >> r) & 1u) && ((wmask >> h) & 1u))
            stage[h * K6_PSLAB + row * 4 + tone] = ieee_sqrtf(inp[r][h] * inp[r][h] + quad[r][h] * quad[r][h]);
      }
    }
  };

  // ---- S2 when its two tries mirror each other (drm == -drp: the candidate came in without drift): symbol i of the
  // + try and symbol 162 - i of the - try have, bit for bit, the same tone frequency (k4_pair.hip has the argument), so
  // one recurrence serves both.  Slot s < 3 gives lane l the pair-row l + 54 s (lane 54 of slot 2: pair-row 162): its
  // P window is symbol r of the + try, its M window symbol 162 - r of the - try -- 22 instructions per sample for the
  // two where the split above spends 28 on two (row, hypothesis) pairs.  Slot 3 only loads.
  auto walk_pair = [&](auto slot_tag) {
    constexpr int SLOT = decltype(slot_tag)::value;
    const int tone = wv & 3;
    const bool act = SLOT < 3 && (lane < K6_TROWS || (SLOT == 2 && lane == K6_TROWS));
    const int rp = act ? lane + K6_TROWS * SLOT : 1;                       // idle lanes shadow pair-row 1
    const int prow = min(rp, UWSPR_NSYM - 1), mrow = min(UWSPR_NSYM - rp, UWSPR_NSYM - 1);
    const bool pok = act && rp < UWSPR_NSYM, mok = act && rp >= 1;
    float cd, sd;
    {
      const float delta = ((float)tone - 1.5f) * 1.46484375f;              // cc:148
      const float fp = (float)((double)fc + ((double)drp / 2.0) * ((double)(float)rp - 81.0) / 81.0);   // cc:173
      double sn, cs;
      sincos(kTwoPiDt6 * (double)(fp + delta), &sn, &cs);                  // cc:188-189
      cd = (float)cs; sd = (float)sn;
    }
    float pc = 1.0f, psn = 0.0f, inpP = 0.0f, quadP = 0.0f, inpM = 0.0f, quadM = 0.0f;
    for (int c = 0; c < nchunks; c++) {
      gload(min(c + 1, nchunks - 1));           // in flight during the arithmetic
      if (SLOT < 3) {
        const lds_f *bp = &stage[(c & 1) * K6_MAXROWS * K6_ROWDW];
        for (int half = 0; half < 2; half++) {
          v4f xp[4], xm[4];
#pragma unroll
          for (int j = 0; j < 4; j++) {
            xp[j] = *(const lds_f4 *)(bp + prow * K6_ROWDW + 16 * half + 4 * j);
            xm[j] = *(const lds_f4 *)(bp + mrow * K6_ROWDW + 16 * half + 4 * j);
          }
#pragma unroll
          for (int j = 0; j < 4; j++) {
#pragma unroll
            for (int e = 0; e < 2; e++) {
              const float px = e ? xp[j].z : xp[j].x, py = e ? xp[j].w : xp[j].y;
              const float mx = e ? xm[j].z : xm[j].x, my = e ? xm[j].w : xm[j].y;
              inpP = (inpP + px * pc) + py * psn;             // cc:206
              quadP = (quadP - px * psn) + py * pc;           // cc:207
              inpM = (inpM + mx * pc) + my * psn;
              quadM = (quadM - mx * psn) + my * pc;
              const float nc = pc * cd - psn * sd;            // cc:193-195
              const float ns = pc * sd + psn * cd;
              pc = nc; psn = ns;
            }
          }
        }
      }
      gstore((c + 1) & 1);
      __syncthreads();
    }
    // tone magnitudes (cc:211) into the p image, which overlays the (now dead) staging buffers
    if (pok) stage[0 * K6_PSLAB + prow * 4 + tone] = ieee_sqrtf(inpP * inpP + quadP * quadP);
    if (mok) stage[1 * K6_PSLAB + mrow * 4 + tone] = ieee_sqrtf(inpM * inpM + quadM * quadM);
  };

  gload(0);
  gstore(0);
  __syncthreads();
  // every slot executes the same number of barriers (one per chunk)
  bool paired = false;
#if K6_SLOTS == 2
  // (the experiment build walks S2's two tries one per slot: the mirrored pair walk wants 163 pair-rows on three slots)
  (void)paired; (void)walk_pair;
  if ((wv >> 2) == 0) walk(std::integral_constant<int, 0>{});
  else walk(std::integral_constant<int, 1>{});
#else
  if (KIND == K6_S2) paired = uni((m_type == UWSPR_LINEAR && drp == -drm && mask == 0x3u) ? 1 : 0) != 0;
  if (KIND == K6_S2 && paired) {
    switch (wv >> 2) {
      case 0: walk_pair(std::integral_constant<int, 0>{}); break;
      case 1: walk_pair(std::integral_constant<int, 1>{}); break;
      case 2: walk_pair(std::integral_constant<int, 2>{}); break;
      default: walk_pair(std::integral_constant<int, 3>{}); break;
    }
  } else {
    switch (wv >> 2) {
      case 0: walk(std::integral_constant<int, 0>{}); break;
      case 1: walk(std::integral_constant<int, 1>{}); break;
      case 2: walk(std::integral_constant<int, 2>{}); break;
      default: walk(std::integral_constant<int, 3>{}); break;
    }
  }
#endif
  __syncthreads();
}

// ---- fold: cc:213-226 (sync) and, SOFT, cc:216-224 + 240-254 (soft symbols) + cc:469-474 (rms) -
// NH hypotheses at a time; hypothesis j reads the [rows][4] magnitudes at slab[j] (LDS).
struct k6_fold_lds {
  float cm[K6_FOLDH][K6_CMS];          // signed cmet per symbol (cc:214-215); later symbol - 128 (cc:471)
  double q[2][K6_FOLDH][UWSPR_NSYM];   // fs/162, fs*fs/162 (cc:243-244)
  float sums[5][K6_FOLDH + 1];         // totp, ss, fsum, f2sum, sq per hypothesis
};

template <bool SOFT>
__device__ __noinline__ void k6_fold(int nh, const int *slab_off, const lds_f *lds0, K6_LDS k6_fold_lds *Fp,
                                     float *sync_out, uint8_t *sym0, float *rms_out) {
  // slab_off[j]: dword offset (from lds0) of hypothesis j's [rows][4] magnitudes; its soft symbols
  // go to sym0 + 162 j; sync_out / rms_out: generic pointers (LDS sy[] or the output record)
  K6_LDS k6_fold_lds &F = *Fp;
  nh = uni(nh);
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wv = uni(tid >> 6);
  // per-symbol terms, all threads
  for (int e = tid; e < nh * UWSPR_NSYM; e += K6_THREADS) {
    const int j = e / UWSPR_NSYM, i = e - j * UWSPR_NSYM;
    const v4f P = ((const lds_f4 *)(lds0 + slab_off[j]))[i];
    const bool bit = pr3_6(i);
    const float cmet = (P.y + P.w) - (P.x + P.z);   // cc:214
    F.cm[j][i] = bit ? cmet : -cmet;                // ss -/+ cmet == ss + (-/+ cmet)
    if (SOFT) {
      const float fs = bit ? P.w - P.y : P.z - P.x;          // cc:219,222
      F.q[0][j][i] = (double)fs / 162.0;                     // cc:243
      F.q[1][j][i] = (double)(fs * fs) / 162.0;              // cc:244
    }
  }
  __syncthreads();
  // the order-sensitive sums: one hypothesis per lane, one kind of sum per wavefront
  if (lane < nh) {
    if (wv == 0) {
      const lds_f4 *s4 = (const lds_f4 *)(lds0 + slab_off[lane]);
      float acc = 0.0f;
#pragma unroll 9
      for (int i = 0; i < UWSPR_NSYM; i++) {            // cc:213
        const v4f P = s4[i];
        acc = acc + P.x; acc = acc + P.y; acc = acc + P.z; acc = acc + P.w;
      }
      F.sums[0][lane] = acc;
    } else if (wv == 1) {
      float acc = 0.0f;
#pragma unroll 9
      for (int i = 0; i < UWSPR_NSYM; i++) acc = acc + F.cm[lane][i];   // cc:215
      F.sums[1][lane] = acc;
    } else if (SOFT && (wv == 2 || wv == 3)) {
      const K6_LDS double *q = F.q[wv - 2][lane];
      float acc = 0.0f;
#pragma unroll 9
      for (int i = 0; i < UWSPR_NSYM; i++) acc = (float)((double)acc + q[i]);
      F.sums[wv][lane] = acc;
    }
  }
  __syncthreads();
  if (tid < nh) sync_out[tid] = ieee_divf(F.sums[1][tid], F.sums[0][tid]);   // cc:226
  if (SOFT) {
    for (int e = tid; e < nh * UWSPR_NSYM; e += K6_THREADS) {
      const int j = e / UWSPR_NSYM, i = e - j * UWSPR_NSYM;
      const v4f P = ((const lds_f4 *)(lds0 + slab_off[j]))[i];
      const float fs = pr3_6(i) ? P.w - P.y : P.z - P.x;
      const float fsum = F.sums[2][j], f2sum = F.sums[3][j];
      const float fac = ieee_sqrtf(f2sum - fsum * fsum);   // cc:246
      float v = ieee_divf(50.0f * fs, fac);                // cc:248 (symfac = 50)
      if (v > 127.0f) v = 127.0f;
      if (v < -128.0f) v = -128.0f;
      v = v + 128.0f;
      const uint8_t b = (v != v) ? (uint8_t)0 : (uint8_t)(int)v;   // cc:251 (NaN -> 0)
      sym0[j * UWSPR_NSYM + i] = b;
      F.cm[j][i] = (float)((double)(float)b - 128.0);              // cc:471
    }
    __syncthreads();
    if (wv == 0 && lane < nh) {
      float sq = 0.0f;
#pragma unroll 9
      for (int i = 0; i < UWSPR_NSYM; i++) { const float y = F.cm[lane][i]; sq += y * y; }   // cc:472
      rms_out[lane] = (float)sqrt((double)sq / 162.0);                                      // cc:474
    }
  }
  __syncthreads();
}

// ---- table set: 5 frequencies x 4 tones x 256 steps of (c, s), cc:186-199 ----------------------
__device__ __forceinline__ void k6_build_tables(float *tabset, float fcentre, float fstep, int m_type,
                                                float drift, float slmc) {
  const int tid = threadIdx.x;
  if (tid < 4 * K6_NTAB) {
    const int q = tid >> 2, tone = tid & 3;
    const float f0 = fcentre + (float)(q - 2) * fstep;                                  // cc:164
    const float fp = (m_type == UWSPR_LINEAR)
                         ? (float)((double)f0 + ((double)drift / 2.0) * ((double)(float)0 - 81.0) / 81.0)
                         : f0 + slmc;                                                   // cc:173 / cc:179
    const float delta = ((float)tone - 1.5f) * 1.46484375f;                             // cc:148
    double sn, cs;
    sincos(kTwoPiDt6 * (double)(fp + delta), &sn, &cs);                                 // cc:188-189
    const float cdq = (float)cs, sdq = (float)sn;
    float c = 1.0f, s = 0.0f;
    float2 *t = reinterpret_cast<float2 *>(tabset) + (q * 4 + tone) * 256;
#pragma unroll 8
    for (int k = 0; k < 256; k++) {
      t[k] = make_float2(c, s);
      const float nc = c * cdq - s * sdq;   // cc:193-195
      const float ns = c * sdq + s * cdq;
      c = nc; s = ns;
    }
  }
  __builtin_amdgcn_s_waitcnt(0);            // the table stores have left this CU
  __syncthreads();
  __builtin_amdgcn_s_dcache_inv();          // scalar cache: no stale lines of an earlier table
}

__device__ __forceinline__ const float *launder(const float *p) {
  // the table is written earlier in this kernel: pin the reads behind the barrier above
  asm volatile("" : "+s"(p));
  return p;
}

#if K6_SLOTS == 2
__global__ __launch_bounds__(K6_THREADS, 4) void k6_sched(k6_args a) {     // 8 wavefronts at <= 128 VGPRs: two workgroups per CU
#else
__global__ __launch_bounds__(K6_THREADS) void k6_sched(k6_args a) {
#endif
  __shared__ __align__(16) float stage[2 * K6_MAXROWS * K6_ROWDW];   // staging, then p[h][163][4]
  __shared__ __align__(16) float pw[UWSPR_NSYM * 4];                 // the current winner's magnitudes
  __shared__ k6_fold_lds F;
  __shared__ cand_state st;
  __shared__ float sy[UWSPR_NJIG + 1];
  __shared__ int slab[K6_FOLDH];     // dword offsets from stage[] of the magnitudes being folded
  __shared__ int s_slot, s_wsrc, s_wroff, s_tq;

  const int tid = threadIdx.x;
  float *tabA = a.tabs + (size_t)blockIdx.x * 2 * K6_TABSET;
  float *tabB = tabA + K6_TABSET;
  const bool reuse = a.reuse != 0;

  for (;;) {
    __syncthreads();
    if (tid == 0) s_slot = atomicAdd(a.counter, 1);
    __syncthreads();
    const int slot = uni(s_slot);
    if (slot >= a.nslots) return;

    const bool resume = a.resume != nullptr;
    if (resume && !a.resume[slot]) continue;   // this slot keeps what the first pass wrote
    // ---- candidate -> state (k_sched_init; cc:404-407) ----
    if (resume) {
      if (tid == 0) st = a.state[slot];
      for (int e = tid; e < UWSPR_NSYM * 4; e += K6_THREADS) pw[e] = a.pwin[(size_t)slot * UWSPR_NSYM * 4 + e];
    } else if (tid == 0) {
      const int b = slot / a.per_frame, j = slot - b * a.per_frame;
      cand_state s0;
      const bool on = j < a.npk[b] && j < a.cand_stride;
      if (on) {
        const uwspr_candidate cnd = a.cands[(size_t)b * a.cand_stride + j];
        s0.frame = b;
        s0.m_type = cnd.m_type;
        s0.slmc = (cnd.m_type == UWSPR_NONLINEAR)
                      ? k6_slm_drift_t0(cnd.m_nonlinear.V1, cnd.m_nonlinear.V2, cnd.m_nonlinear.p1,
                                        cnd.m_nonlinear.p2, a.cf)
                      : 0.0f;
        s0.f1 = cnd.freq;
        s0.drift1 = (cnd.m_type == UWSPR_LINEAR) ? cnd.m_linear.drift : 0.0f;   // cc:405,373
        s0.shift1 = cnd.shift;
        s0.sync1 = cnd.sync;
      } else {
        s0.frame = -1; s0.m_type = 0; s0.slmc = 0.0f; s0.f1 = 0.0f; s0.drift1 = 0.0f;
        s0.shift1 = 0; s0.sync1 = 0.0f;
      }
      s0.worth = 0; s0.driftp = 0.0f; s0.driftm = 0.0f; s0.csync = 0.0f; s0.cknown = 0;
      st = s0;
    }
    __syncthreads();
    uwspr_demod_out *o = a.out + slot;
    const bool live = st.frame >= 0 && st.frame < a.nframes;
    if (resume && live && !st.worth) continue;   // nothing more to produce (cc:453-457)
    if (!live) {
      uint32_t *ow = reinterpret_cast<uint32_t *>(o);
      for (int e = tid; e < (int)(sizeof(uwspr_demod_out) / 4); e += K6_THREADS) ow[e] = 0u;
      if (tid == 0 && a.state) a.state[slot] = st;
      continue;
    }
    auto stamp = [&](int k) {
      if (a.stamps && tid == 0) a.stamps[(size_t)slot * 64 + k] = wall_clock64();
    };
    stamp(0);
    const float2 *fb = a.frames + (size_t)st.frame * a.fstride;
    const int m_type = uni(st.m_type);
    const float slmc = st.slmc;

    // mode 0/1 fold of a 5- (2-) hypothesis stage: hypotheses of `mask` from the p image;
    // `wrap_h` (S0): hypothesis 4 = hypothesis 0 one row later; the others are the known
    // hypothesis (the previous winner: metric carried in st.csync)
    auto fold_plain = [&](int nh, uint32_t mask, int wrap_h) {
      if (tid < nh)
        slab[tid] = ((mask >> tid) & 1u) ? tid * K6_PSLAB : (tid == wrap_h ? 4 : (int)(pw - stage));
      __syncthreads();
      k6_fold<false>(nh, slab, (const lds_f *)stage, (K6_LDS k6_fold_lds *)&F, sy, nullptr, nullptr);
      if (tid < nh && !((mask >> tid) & 1u) && tid != wrap_h) sy[tid] = st.csync;
      __syncthreads();
    };
    auto keep_winner = [&]() {
      // s_wsrc >= 0: hypothesis whose magnitudes become the winner's (s_wroff: row offset of the wrap)
      const int src = uni(s_wsrc), roff = uni(s_wroff);
      if (src >= 0)
        for (int e = tid; e < UWSPR_NSYM * 4; e += K6_THREADS) pw[e] = stage[src * K6_PSLAB + 4 * roff + e];
      __syncthreads();
    };

    // =========================== S0 (cc:409-415): lag = shift1-128..+128 step 64, mode 0
    bool tabled = (m_type != UWSPR_LINEAR) || (st.drift1 == 0.0f);
    if (!resume) {
    if (tabled) k6_build_tables(tabA, st.f1, 0.25f, m_type, st.drift1, slmc);
    stamp(1);
    {
      const float f0v = st.f1 + (float)0 * 0.0f;
      const int L0 = st.shift1 - 128;
      if (tabled) {
        k6_pass<K6_S0, true>(fb, a.np, L0, K6_MAXROWS, 28, 0x0fu, 2, f0v, 0.0f, st.drift1, 0.0f, m_type, slmc,
                             launder(tabA), (lds_f *)stage);
        fold_plain(5, 0x0fu, 4);
      } else {
        k6_pass<K6_S0, false>(fb, a.np, L0, UWSPR_NSYM, 32, 0x1fu, 2, f0v, 0.0f, st.drift1, 0.0f, m_type, slmc,
                              tabA, (lds_f *)stage);
        fold_plain(5, 0x1fu, -1);
      }
      if (tid == 0) {   // transition to S1 (sched_step_body<1>)
        float bs = -1e30f; int bshift = 0; float bf = 0.0f; int bq = -1;
        for (int q = 0; q < 5; q++)
          if (sy[q] > bs) { bs = sy[q]; bshift = L0 + 64 * q; bf = f0v; bq = q; }   // cc:227-231
        st.sync1 = bs; st.shift1 = bshift; st.f1 = bf;
        st.cknown = (reuse && bs > -1e30f) ? 1 : 0;
        st.csync = bs;
        s_wsrc = (bq == 4 && tabled) ? 0 : bq; s_wroff = (bq == 4 && tabled) ? 1 : 0;
      }
      __syncthreads();
      keep_winner();
    }

    stamp(2);
    // =========================== S1 (cc:416-419): f = f1 + ifreq 0.25, mode 1
    {
      const float fc = st.f1;
      // after S0 the frequency is unchanged unless no hypothesis won (f1 = 0.0 default): then table set A
      // (built around the candidate frequency) does not apply
      const bool tabs_ok = tabled && (st.sync1 > -1e30f);
      const uint32_t mask = st.cknown ? 0x1bu : 0x1fu;
      float f0[5];
#pragma unroll
      for (int q = 0; q < 5; q++) f0[q] = fc + (float)(q - 2) * 0.25f;
      const int L0 = st.shift1;
      if (tabs_ok) k6_pass<K6_S1, true>(fb, a.np, L0, UWSPR_NSYM, 16, mask, 0, fc, 0.25f, st.drift1, 0.0f, m_type, slmc, launder(tabA), (lds_f *)stage);
      else k6_pass<K6_S1, false>(fb, a.np, L0, UWSPR_NSYM, 16, mask, 0, fc, 0.25f, st.drift1, 0.0f, m_type, slmc, tabA, (lds_f *)stage);
      stamp(10);
      fold_plain(5, mask, -1);
      stamp(11);
      if (tid == 0) {   // transition to S2 (sched_step_body<2>)
        float bs = -1e30f; int bshift = 0; float bf = 0.0f; int bq = -1;
        for (int q = 0; q < 5; q++)
          if (sy[q] > bs) { bs = sy[q]; bshift = L0; bf = f0[q]; bq = q; }
        st.sync1 = bs; st.shift1 = bshift; st.f1 = bf;
        const int was_known = st.cknown;
        st.cknown = (reuse && bs > -1e30f) ? 1 : 0;
        st.csync = bs;
        st.driftp = (float)((double)st.drift1 + 0.5);
        st.driftm = (float)((double)st.drift1 - 0.5);
        s_wsrc = (bq == 2 && was_known) ? -1 : bq; s_wroff = 0;
      }
      __syncthreads();
      keep_winner();
    }

    stamp(3);
    // =========================== S2 (cc:423-441): linear only, drift1 +- 0.5 at (f1, shift1)
    if (m_type == UWSPR_LINEAR) {
      const float f0v = st.f1 + (float)0 * 0.0f;
      k6_pass<K6_S2, false>(fb, a.np, st.shift1, UWSPR_NSYM, 16, 0x3u, 0, f0v, 0.0f, st.driftp, st.driftm, m_type, slmc, tabA, (lds_f *)stage);
      fold_plain(2, 0x3u, -1);
      if (tid == 0) {   // sched_step_body<3>, first half (cc:434-441)
        float syncp = -1e30f, syncm = -1e30f;
        if (sy[0] > syncp) syncp = sy[0]; else { st.f1 = 0.0f; st.shift1 = 0; st.cknown = 0; }
        if (sy[1] > syncm) syncm = sy[1]; else { st.f1 = 0.0f; st.shift1 = 0; st.cknown = 0; }
        s_wsrc = -1; s_wroff = 0;
        if (syncp > st.sync1) { st.drift1 = st.driftp; st.sync1 = syncp; s_wsrc = 0; }
        else if (syncm > st.sync1) { st.drift1 = st.driftm; st.sync1 = syncm; s_wsrc = 1; }
      }
      __syncthreads();
      keep_winner();
    }
    if (tid == 0) {
      st.worth = (st.sync1 > 0.10f) ? 1 : 0;   // cc:443
      st.csync = st.sync1;
    }
    }   // !resume
    __syncthreads();

    stamp(4);
    const int njig = a.njig;
    if (st.worth) {
      // =========================== S3 (cc:444-447): lag = shift1-32..+32 step 16, mode 0
      tabled = (m_type != UWSPR_LINEAR) || (st.drift1 == 0.0f);
      // (resume: the table set is rebuilt around the FINAL f1, whose middle table is the S5 frequency)
      if (tabled) k6_build_tables(tabB, st.f1, 0.05f, m_type, st.drift1, slmc);
      if (resume && tid == 0) s_tq = 2;
      stamp(5);
      if (!resume) {
      {
        const float f0v = st.f1 + (float)0 * 0.0f;
        const int L0 = st.shift1 - 32;
        const uint32_t mask = st.cknown ? 0x1bu : 0x1fu;
        if (tabled) k6_pass<K6_S3, true>(fb, a.np, L0, UWSPR_NSYM, 20, mask, 2, f0v, 0.0f, st.drift1, 0.0f, m_type, slmc, launder(tabB), (lds_f *)stage);
        else k6_pass<K6_S3, false>(fb, a.np, L0, UWSPR_NSYM, 20, mask, 2, f0v, 0.0f, st.drift1, 0.0f, m_type, slmc, tabB, (lds_f *)stage);
        fold_plain(5, mask, -1);
        if (tid == 0) {   // sched_step_body<4>
          float bs = -1e30f; int bshift = 0; float bf = 0.0f; int bq = -1;
          for (int q = 0; q < 5; q++)
            if (sy[q] > bs) { bs = sy[q]; bshift = L0 + 16 * q; bf = f0v; bq = q; }
          st.sync1 = bs; st.shift1 = bshift; st.f1 = bf;
          const int was_known = st.cknown;
          st.cknown = (reuse && bs > -1e30f) ? 1 : 0;
          st.csync = bs;
          s_wsrc = (bq == 2 && was_known) ? -1 : bq; s_wroff = 0;
        }
        __syncthreads();
        keep_winner();
      }
      stamp(6);
      // =========================== S4 (cc:449-452): f = f1 + ifreq 0.05, mode 1
      {
        const float fc = st.f1;
        const bool tabs_ok = tabled && (st.sync1 > -1e30f);
        const uint32_t mask = st.cknown ? 0x1bu : 0x1fu;
        float f0[5];
#pragma unroll
        for (int q = 0; q < 5; q++) f0[q] = fc + (float)(q - 2) * 0.05f;
        const int L0 = st.shift1;
        if (tabs_ok) k6_pass<K6_S4, true>(fb, a.np, L0, UWSPR_NSYM, 16, mask, 0, fc, 0.05f, st.drift1, 0.0f, m_type, slmc, launder(tabB), (lds_f *)stage);
        else k6_pass<K6_S4, false>(fb, a.np, L0, UWSPR_NSYM, 16, mask, 0, fc, 0.05f, st.drift1, 0.0f, m_type, slmc, tabB, (lds_f *)stage);
        fold_plain(5, mask, -1);
        if (tid == 0) {   // sched_step_body<5>
          float bs = -1e30f; int bshift = 0; float bf = 0.0f; int bq = -1;
          for (int q = 0; q < 5; q++)
            if (sy[q] > bs) { bs = sy[q]; bshift = L0; bf = f0[q]; bq = q; }
          st.sync1 = bs; st.shift1 = bshift; st.f1 = bf;
          const int was_known = st.cknown;
          st.cknown = (reuse && bs > -1e30f) ? 1 : 0;
          st.csync = bs;
          s_wsrc = (bq == 2 && was_known) ? -1 : bq; s_wroff = 0;
          s_tq = bq;   // the winner's table (the S5 frequency) -- -1: none won
        }
        __syncthreads();
        keep_winner();
      }
      }   // !resume
      __syncthreads();
      stamp(7);
      // =========================== S5 (cc:457-482): the jiggered shifts, mode 2
      {
        // try idt has shift 8 ii(idt), ii = (+-)ceil(idt/2); lags ascend from L0 = shift1 - 64
        const int wq = uni(s_tq);
        const bool tab5 = tabled && wq >= 0;
        const uint32_t want = (resume || njig >= UWSPR_NJIG) ? 0x1ffffu : ((1u << njig) - 1u);
        uint32_t mask = want;
        if (st.cknown) mask &= ~1u;   // try 0 repeats the S4 winner: its magnitudes are in pw
        const int L0 = st.shift1 - 64;
        if (mask) {
          if (tab5) k6_pass<K6_S5, true>(fb, a.np, L0, UWSPR_NSYM, 24, mask, wq, st.f1, 0.0f, st.drift1, 0.0f, m_type, slmc, launder(tabB), (lds_f *)stage);
          else k6_pass<K6_S5, false>(fb, a.np, L0, UWSPR_NSYM, 24, mask, 0, st.f1, 0.0f, st.drift1, 0.0f, m_type, slmc, tabB, (lds_f *)stage);
        }
        stamp(8);
        for (int base = 0; base < UWSPR_NJIG; base += K6_FOLDH) {
          const int nh = min(K6_FOLDH, UWSPR_NJIG - base);
          const int nwant = resume ? nh : max(0, min(nh, njig - base));   // the tries wanted are a prefix
          if (tid < nh) {
            const int idt = base + tid;
            slab[tid] = ((mask >> idt) & 1u) ? idt * K6_PSLAB : (int)(pw - stage);
          }
          __syncthreads();
          if (nwant > 0)
            k6_fold<true>(nwant, slab, (const lds_f *)stage, (K6_LDS k6_fold_lds *)&F, &o->jig_sync[base],
                          &o->symbols[base][0], &o->jig_rms[base]);
          if (tid < nh) {
            const int idt = base + tid;
            int ii = (idt + 1) / 2;                      // cc:459-462
            if (idt % 2 == 1) ii = -ii;
            if (tid < nwant) o->jig_shift[idt] = st.shift1 + 8 * ii;
            else { o->jig_sync[idt] = 0.0f; o->jig_rms[idt] = 0.0f; o->jig_shift[idt] = 0; }
          }
          for (int e = tid; e < (nh - nwant) * UWSPR_NSYM; e += K6_THREADS)
            (&o->symbols[base + nwant][0])[e] = 0;
          __syncthreads();
        }
      }
    } else {
      // not worth a try: the reference computes no soft symbols (cc:453-457)
      uint32_t *ow = reinterpret_cast<uint32_t *>(o);
      for (int e = tid + 5; e < (int)(sizeof(uwspr_demod_out) / 4); e += K6_THREADS) ow[e] = 0u;
    }
    __syncthreads();
    stamp(9);
    if (tid == 0 && !resume) {
      o->f1 = st.f1; o->drift1 = st.drift1; o->sync1 = st.sync1; o->shift1 = st.shift1;
      o->worth_a_try = st.worth; o->_pad[0] = 0; o->_pad[1] = 0;
      if (a.state) a.state[slot] = st;
    }
    if (a.pwin && st.worth && !resume)
      for (int e = tid; e < UWSPR_NSYM * 4; e += K6_THREADS) a.pwin[(size_t)slot * UWSPR_NSYM * 4 + e] = pw[e];
  }
}

void launch_sched_fused(uwspr_ctx *c, const float *frames, int B, const uwspr_candidate *cands,
                        const int32_t *npk, int cand_stride, int per_frame, uwspr_demod_out *out,
                        int njig, const uint8_t *resume) {
  const int nslots = B * per_frame;
  if (nslots <= 0) return;
  prof_scope ps(c, UWSPR_K_TONECORR, (int64_t)nslots * 35, true);
  k6_args a;
  a.frames = (const float2 *)frames; a.fstride = c->fstride; a.np = c->np; a.nframes = B;
  a.cands = cands; a.npk = npk; a.cand_stride = cand_stride; a.per_frame = per_frame; a.nslots = nslots;
  a.cf = (float)c->p.cf; a.reuse = c->reuse_centre ? 1 : 0;
  a.njig = njig;
  a.tabs = c->d_tabs; a.counter = c->d_counter; a.out = out; a.state = c->d_state; a.resume = resume;
  // the winner's magnitudes are kept only for a lazy pass (run_schedule sized d_pwin for THIS batch) and
  // read only by a resume pass: an eager pass must not write a buffer sized for an earlier, smaller batch
  a.pwin = (njig < UWSPR_NJIG || resume) ? c->d_pwin : nullptr;
  a.stamps = c->d_sched_stamps;
  const int grid = nslots < c->sched_grid ? nslots : c->sched_grid;
  (void)hipMemsetAsync(c->d_counter, 0, sizeof(int), c->stream);
  launch_timed(c, ps, k6_sched, dim3(grid), dim3(K6_THREADS), 0, a);
}

}  // namespace uwspr
