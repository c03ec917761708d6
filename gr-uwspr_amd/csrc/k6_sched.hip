// K6 -- the whole refinement schedule of a candidate (S0..S5) in ONE workgroup.
//
// Reference: sync_and_demodulate_impl::demodulate, lib/sync_and_demodulate_impl.cc:403-482
// (the six dependent calls of sync_and_demodulate(), cc:126-256, per candidate).
//
// The candidates of a batch are independent (cc:389 loops over them) and every
// stage of a candidate needs all of the previous stage's 162 symbols, so the
// natural unit is: one 12-wavefront workgroup = one candidate, running
// S0 -> S5 back to back with no kernel boundary, no tone magnitudes in HBM and
// no cross-workgroup traffic.  (The staged form -- k4_* + k5_fold_step, 17
// launches per batch -- stays as UWSPR_SCHED_FUSED=0; both give the same bytes.)
//
// Mapping.  Wavefront w works on tone (w & 3) of the 54 symbols 54 (w >> 2) ..;
// a lane owns ONE symbol window ("row") and accumulates inp/quad for all the
// hypotheses of the stage against it, every accumulator seeing exactly the
// reference's sequence of binary32 operations (cc:206-207: no FMA, no tree).
//  * Samples: the stage streams the rows [L0 + 256 i, L0 + 256 i + 256 + span)
//    through a double-buffered LDS image, 16 samples per row and chunk, loaded
//    cooperatively with coalesced 8-byte loads (cc:205's n > 0 && n < np test is
//    applied by the loader: a skipped sample is a zero, which leaves inp/quad
//    unchanged).  A lane reads its 16 samples once per chunk (8 ds_read_b128)
//    and uses them for every hypothesis: a hypothesis whose lag is L0 + D sees
//    stream position a as its sample k = a - D ("sample-major" order), so lag
//    sweeps cost no extra loads -- S5's 17 lags are one pass over 384 samples.
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
//  * Fold (cc:213-226, 240-254) from LDS by up to six wavefronts, the order-sensitive
//    running sums on single lanes as in k5_fold_wave; stage transitions (cc:227-231,
//    416-452) by one thread; the stage winner's tone magnitudes are kept (2.6 KB) so
//    that the hypothesis a later stage repeats -- the middle one of S1/S3/S4 and the
//    first jiggered shift of S5 -- is not computed again.
//  * S0's last lag is one symbol after its first: (last lag, symbol i) is (first
//    lag, symbol i+1) when the frequency does not depend on the symbol; a 163rd
//    "virtual" row on an otherwise idle lane supplies (last lag, symbol 161).
#include <stdlib.h>

#include "uwspr_internal.h"

#pragma clang fp contract(off)

namespace uwspr {

constexpr int K6_WAVES = 12;
constexpr int K6_THREADS = 64 * K6_WAVES;
constexpr int K6_TROWS = 54;                  // rows per wavefront: 162 = 3 x 54
constexpr int K6_MAXROWS = UWSPR_NSYM + 1;    // + the virtual row of the S0 wrap
constexpr int K6_ROWDW = 36;                  // dwords per staged row: 16 samples x 8 B + 16 B pad
constexpr int K6_NTAB = 5;                    // frequencies per table set
constexpr int K6_TABSET = K6_NTAB * 4 * 512;  // floats per table set: [freq][tone][256](c, s)
constexpr int K6_FOLDW = 6;                   // wavefronts that fold concurrently
constexpr int K6_PSLAB = K6_MAXROWS * 4;      // floats per hypothesis in the p image

constexpr double kTwoPiDt6 = 2.0 * 3.14159265358979323846 * (double)(float)(1.0 / 375.0);  // cc:146,188

__device__ __constant__ uint32_t kPr3_6[6] = UWSPR_PR3_WORDS;
__device__ __forceinline__ bool pr3_6(int i) { return (kPr3_6[i >> 5] >> (i & 31)) & 1u; }

typedef float f16v __attribute__((ext_vector_type(16)));
#define K6_CONST __attribute__((address_space(4)))

__device__ __forceinline__ int uni(int v) { return __builtin_amdgcn_readfirstlane(v); }
__device__ __forceinline__ void k6_wave_fence() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
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

struct k6_scr {                 // per folding wavefront
  float cm[UWSPR_NSYM + 2];     // signed cmet per symbol (cc:214-215); later (symbol - 128) for the rms
  double q[2][UWSPR_NSYM];      // fs/162, fs*fs/162 (cc:243-244)
};

struct k6_args {
  const float2 *frames; int fl; int nframes;
  const uwspr_candidate *cands; const int32_t *npk; int cand_stride; int per_frame; int nslots;
  float cf; int reuse;
  int njig;                     // mode-2 tries to produce: 17, or fewer (lazy S5)
  float *tabs;                  // [gridDim.x][2][K6_TABSET]
  int *counter;                 // slot queue head
  uwspr_demod_out *out;
  cand_state *state;            // [nslots] final state (resume / diagnostics)
  float *pwin;                  // [nslots][162][4] winner magnitudes (resume), or null
};

// ---- one pass over the sample stream -----------------------------------------------------------
// The six stages as compile-time geometry: hypothesis h of stage KIND has lag L0 + 8 dk8(h) and
// either phasor table tq(h) of the stage's table set (TAB) or, on the per-lane path, frequency
// fc + (h - 2) fstep and drift drp (S2: drp for h = 0, drm for h = 1, both at fc).
enum { K6_S0 = 0, K6_S1 = 1, K6_S2 = 2, K6_S3 = 3, K6_S4 = 4, K6_S5 = 5 };
template <int KIND> struct k6_geom {
  static constexpr int HMAX = KIND == K6_S2 ? 2 : KIND == K6_S5 ? UWSPR_NJIG : 5;
  static constexpr bool LAGS = KIND == K6_S0 || KIND == K6_S3 || KIND == K6_S5;   // one frequency, several lags
  __host__ __device__ static constexpr int dk8(int h) {
    return KIND == K6_S0 ? 8 * h                                            // shift1 - 128 + 64 h  (cc:409-411)
         : KIND == K6_S3 ? 2 * h                                            // shift1 - 32 + 16 h   (cc:444)
         : KIND == K6_S5 ? 8 + ((h & 1) ? -((h + 1) / 2) : (h + 1) / 2)     // shift1 + 8 ii(idt)   (cc:459-463)
         : 0;
  }
};

// bit h of `mask` = hypothesis h is computed; tq0 = table of a lag sweep (LAGS); S1/S4 use table h.
template <int KIND, bool TAB>
__device__ __forceinline__ void k6_pass(const float2 *__restrict__ fb, int fl, int L0, int nrows,
                                        int nchunks, uint32_t mask, int tq0, float fc, float fstep,
                                        float drp, float drm, int m_type, float slmc,
                                        const float *tabset, float *stage) {
  using G = k6_geom<KIND>;
  constexpr int HMAX = G::HMAX;
  constexpr bool SHARED = G::LAGS;
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wv = uni(tid >> 6);
  const int tone = wv & 3, third = wv >> 2;
  // lanes 0..53: the wave's symbols; lane 54 of the last third: the virtual row (nrows == 163)
  int row = K6_TROWS * third + lane;
  const bool has_row = (lane < K6_TROWS) || (third == 2 && lane == K6_TROWS && nrows > UWSPR_NSYM);
  if (!has_row) row = K6_TROWS * third;   // idle lanes shadow a real row (results discarded)

  // ---- loader: element e = tid + 768 n -> row tid / 16 + 48 n, sample tid % 16
  const int lr = tid >> 4, lj = tid & 15;
  const int sbase = lr * K6_ROWDW + 2 * lj;
  const bool interior = (L0 > 0) && (L0 + 256 * (nrows - 1) + 16 * nchunks < fl);   // workgroup-uniform
  float2 greg[4];
  auto gload = [&](int c) {
    if (interior) {
#pragma unroll
      for (int n = 0; n < 4; n++) {
        const int r = min(lr + 48 * n, nrows - 1);
        greg[n] = fb[L0 + 256 * r + lj + 16 * c];
      }
    } else {
#pragma unroll
      for (int n = 0; n < 4; n++) {
        const int r = min(lr + 48 * n, nrows - 1);
        const int ns = L0 + 256 * r + lj + 16 * c;
        const bool inr = (ns > 0) && (ns < fl);                 // cc:205, sample 0 excluded
        const float2 v = fb[min(max(ns, 0), fl - 1)];
        greg[n] = inr ? v : make_float2(0.0f, 0.0f);
      }
    }
  };
  auto gstore = [&](int buf) {
#pragma unroll
    for (int n = 0; n < 4; n++)
      if (lr + 48 * n < nrows)
        *reinterpret_cast<float2 *>(&stage[buf * K6_MAXROWS * K6_ROWDW + sbase + 48 * n * K6_ROWDW]) = greg[n];
  };

  float inp[HMAX], quad[HMAX];
#pragma unroll
  for (int h = 0; h < HMAX; h++) { inp[h] = 0.0f; quad[h] = 0.0f; }

  // per-lane path: phasor steps from this lane's symbol frequency (cc:170-189)
  constexpr int NPH = TAB ? 1 : HMAX;
  constexpr int NST = TAB ? 1 : (SHARED ? 1 : HMAX);
  float pc[NPH], psn[NPH], cd[NST], sd[NST];
  if (!TAB) {
    const float delta = ((float)tone - 1.5f) * 1.46484375f;          // cc:148
    const int own_i = min(row, UWSPR_NSYM - 1);
#pragma unroll
    for (int h = 0; h < NST; h++) {
      const float f0 = (KIND == K6_S2 || SHARED) ? fc : fc + (float)(h - 2) * fstep;   // cc:164
      const float dr = KIND == K6_S2 ? (h == 0 ? drp : drm) : drp;
      float fp;
      if (m_type == UWSPR_LINEAR)
        fp = (float)((double)f0 + ((double)dr / 2.0) * ((double)(float)own_i - 81.0) / 81.0);  // cc:173
      else
        fp = f0 + slmc;                                               // cc:179 (t = 0)
      double sn, cs;
      sincos(kTwoPiDt6 * (double)(fp + delta), &sn, &cs);             // cc:188-189
      cd[h] = (float)cs; sd[h] = (float)sn;
    }
#pragma unroll
    for (int h = 0; h < NPH; h++) { pc[h] = 1.0f; psn[h] = 0.0f; }
  }
  const K6_CONST float *tabw = (const K6_CONST float *)tabset + tone * 512 + (SHARED ? tq0 * 2048 : 0);

  gload(0);
  gstore(0);
  __syncthreads();
  for (int c = 0; c < nchunks; c++) {
    gload(min(c + 1, nchunks - 1));           // in flight during the arithmetic
    const float *rowp = &stage[(c & 1) * K6_MAXROWS * K6_ROWDW + row * K6_ROWDW];
#pragma unroll
    for (int half = 0; half < 2; half++) {
      float4 xv[4];
#pragma unroll
      for (int j = 0; j < 4; j++) xv[j] = *reinterpret_cast<const float4 *>(rowp + 16 * half + 4 * j);
#pragma unroll
      for (int h = 0; h < HMAX; h++) {
        if (!((mask >> h) & 1u)) continue;                 // uniform
        const int k0 = 16 * c + 8 * half - 8 * G::dk8(h);  // uniform: this hypothesis' first step
        if (k0 < 0 || k0 > 248) continue;
        if (TAB) {
          const f16v ph = *(const K6_CONST f16v *)(tabw + (SHARED ? 0 : h * 2048) + 2 * k0);
#pragma unroll
          for (int j = 0; j < 4; j++) {
            const float4 x = xv[j];
            inp[h] = (inp[h] + x.x * ph[4 * j]) + x.y * ph[4 * j + 1];        // cc:206
            quad[h] = (quad[h] - x.x * ph[4 * j + 1]) + x.y * ph[4 * j];      // cc:207
            inp[h] = (inp[h] + x.z * ph[4 * j + 2]) + x.w * ph[4 * j + 3];
            quad[h] = (quad[h] - x.z * ph[4 * j + 3]) + x.w * ph[4 * j + 2];
          }
        } else {
          const int hs = SHARED ? 0 : h;
#pragma unroll
          for (int j = 0; j < 4; j++) {
            const float4 x = xv[j];
#pragma unroll
            for (int e = 0; e < 2; e++) {
              const float xx = e ? x.z : x.x, xy = e ? x.w : x.y;
              inp[h] = (inp[h] + xx * pc[h]) + xy * psn[h];                   // cc:206
              quad[h] = (quad[h] - xx * psn[h]) + xy * pc[h];                 // cc:207
              const float nc = pc[h] * cd[hs] - psn[h] * sd[hs];              // cc:193-195
              const float ns = pc[h] * sd[hs] + psn[h] * cd[hs];
              pc[h] = nc; psn[h] = ns;
            }
          }
        }
      }
    }
    gstore((c + 1) & 1);
    __syncthreads();
  }
  // tone magnitudes (cc:211) into the p image, which overlays the (now dead) staging buffers
  if (has_row) {
#pragma unroll
    for (int h = 0; h < HMAX; h++)
      if ((mask >> h) & 1u)
        stage[h * K6_PSLAB + row * 4 + tone] = ieee_sqrtf(inp[h] * inp[h] + quad[h] * quad[h]);
  }
  __syncthreads();
}

// ---- fold of one hypothesis by one wavefront (cc:213-226; SOFT: cc:216-224, 240-254) -----------
// slab: [rows][4] tone magnitudes in LDS.  Returns sync in every lane.
template <bool SOFT>
__device__ __forceinline__ float k6_fold(const float *slab, k6_scr &S, float symfac, uint8_t *sym_out,
                                         float *rms_out) {
  const int lane = threadIdx.x & 63;
  const float4 *s4 = reinterpret_cast<const float4 *>(slab);
  float fsr[3];
#pragma unroll
  for (int r = 0; r < 3; r++) {
    const int i = lane + 64 * r;
    fsr[r] = 0.0f;
    if (i < UWSPR_NSYM) {
      const float4 P = s4[i];
      const bool bit = pr3_6(i);
      const float cmet = (P.y + P.w) - (P.x + P.z);   // cc:214
      S.cm[i] = bit ? cmet : -cmet;                   // ss -/+ cmet == ss + (-/+ cmet)
      fsr[r] = bit ? P.w - P.y : P.z - P.x;           // cc:219,222
      if (SOFT) {
        S.q[0][i] = (double)fsr[r] / 162.0;                  // cc:243
        S.q[1][i] = (double)(fsr[r] * fsr[r]) / 162.0;       // cc:244
      }
    }
  }
  k6_wave_fence();
  float acc = 0.0f;
  if (lane == 0) {
#pragma unroll 9
    for (int i = 0; i < UWSPR_NSYM; i++) {            // cc:213
      const float4 P = s4[i];
      acc = acc + P.x; acc = acc + P.y; acc = acc + P.z; acc = acc + P.w;
    }
  } else if (lane == 1) {
#pragma unroll 9
    for (int i = 0; i < UWSPR_NSYM; i++) acc = acc + S.cm[i];   // cc:215
  } else if (SOFT && lane < 4) {
    const double *q = S.q[lane - 2];
#pragma unroll 9
    for (int i = 0; i < UWSPR_NSYM; i++) acc = (float)((double)acc + q[i]);
  }
  const float totp = __shfl(acc, 0), ss = __shfl(acc, 1);
  const float sync = ieee_divf(ss, totp);   // cc:226
  if (SOFT) {
    const float fsum = __shfl(acc, 2), f2sum = __shfl(acc, 3);
    const float fac = ieee_sqrtf(f2sum - fsum * fsum);   // cc:246
    k6_wave_fence();   // cm[] is read no more
#pragma unroll
    for (int r = 0; r < 3; r++) {
      const int i = lane + 64 * r;
      if (i < UWSPR_NSYM) {
        float v = ieee_divf(symfac * fsr[r], fac);   // cc:248
        if (v > 127.0f) v = 127.0f;
        if (v < -128.0f) v = -128.0f;
        v = v + 128.0f;
        const uint8_t b = (v != v) ? (uint8_t)0 : (uint8_t)(int)v;   // cc:251 (NaN -> 0)
        sym_out[i] = b;
        S.cm[i] = (float)((double)(float)b - 128.0);                 // cc:471
      }
    }
    k6_wave_fence();
    if (lane == 0) {
      float sq = 0.0f;
#pragma unroll 9
      for (int i = 0; i < UWSPR_NSYM; i++) { const float y = S.cm[i]; sq += y * y; }   // cc:472
      *rms_out = (float)sqrt((double)sq / 162.0);                                     // cc:474
    }
  }
  k6_wave_fence();
  return sync;
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

__global__ __launch_bounds__(K6_THREADS) void k6_sched(k6_args a) {
  __shared__ __align__(16) float stage[2 * K6_MAXROWS * K6_ROWDW];   // staging, then p[h][163][4]
  __shared__ __align__(16) float pw[UWSPR_NSYM * 4];                 // the current winner's magnitudes
  __shared__ k6_scr scr[K6_FOLDW];
  __shared__ cand_state st;
  __shared__ float sy[UWSPR_NJIG];
  __shared__ int s_slot, s_wsrc, s_wroff, s_tq;
  extern __shared__ char k6_pad[];   // launch-time padding: bounds the workgroups per CU
  (void)k6_pad;

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wv = uni(tid >> 6);
  float *tabA = a.tabs + (size_t)blockIdx.x * 2 * K6_TABSET;
  float *tabB = tabA + K6_TABSET;
  const bool reuse = a.reuse != 0;

  for (;;) {
    __syncthreads();
    if (tid == 0) s_slot = atomicAdd(a.counter, 1);
    __syncthreads();
    const int slot = uni(s_slot);
    if (slot >= a.nslots) return;

    // ---- candidate -> state (k_sched_init; cc:404-407) ----
    if (tid == 0) {
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
    if (!live) {
      uint32_t *ow = reinterpret_cast<uint32_t *>(o);
      for (int e = tid; e < (int)(sizeof(uwspr_demod_out) / 4); e += K6_THREADS) ow[e] = 0u;
      if (tid == 0 && a.state) a.state[slot] = st;
      continue;
    }
    const float2 *fb = a.frames + (size_t)st.frame * a.fl;
    const int m_type = uni(st.m_type);
    const float slmc = st.slmc;

    // the fold + winner bookkeeping shared by the stages
    auto fold_plain = [&](int nh, uint32_t mask, int wrap_h) {
      // hypotheses of `mask` from the p image; `wrap_h` (S0): hypothesis 4 = hypothesis 0 one row later
      if (wv < K6_FOLDW) {
        for (int h = wv; h < nh; h += K6_FOLDW) {
          float s;
          if ((mask >> h) & 1u) s = k6_fold<false>(&stage[h * K6_PSLAB], scr[wv], 50.0f, nullptr, nullptr);
          else if (h == wrap_h) s = k6_fold<false>(&stage[4], scr[wv], 50.0f, nullptr, nullptr);
          else s = st.csync;   // the known hypothesis: the previous winner, metric carried
          if (lane == 0) sy[h] = s;
        }
      }
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
    if (tabled) k6_build_tables(tabA, st.f1, 0.25f, m_type, st.drift1, slmc);
    {
      const float f0v = st.f1 + (float)0 * 0.0f;
      const int L0 = st.shift1 - 128;
      if (tabled) {
        k6_pass<K6_S0, true>(fb, a.fl, L0, K6_MAXROWS, 28, 0x0fu, 2, f0v, 0.0f, st.drift1, 0.0f, m_type, slmc,
                             launder(tabA), stage);
        fold_plain(5, 0x0fu, 4);
      } else {
        k6_pass<K6_S0, false>(fb, a.fl, L0, UWSPR_NSYM, 32, 0x1fu, 2, f0v, 0.0f, st.drift1, 0.0f, m_type, slmc,
                              tabA, stage);
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
      if (tabs_ok) k6_pass<K6_S1, true>(fb, a.fl, L0, UWSPR_NSYM, 16, mask, 0, fc, 0.25f, st.drift1, 0.0f, m_type, slmc, launder(tabA), stage);
      else k6_pass<K6_S1, false>(fb, a.fl, L0, UWSPR_NSYM, 16, mask, 0, fc, 0.25f, st.drift1, 0.0f, m_type, slmc, tabA, stage);
      fold_plain(5, mask, -1);
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

    // =========================== S2 (cc:423-441): linear only, drift1 +- 0.5 at (f1, shift1)
    if (m_type == UWSPR_LINEAR) {
      const float f0v = st.f1 + (float)0 * 0.0f;
      k6_pass<K6_S2, false>(fb, a.fl, st.shift1, UWSPR_NSYM, 16, 0x3u, 0, f0v, 0.0f, st.driftp, st.driftm, m_type, slmc, tabA, stage);
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
    __syncthreads();

    const int njig = a.njig;
    if (st.worth) {
      // =========================== S3 (cc:444-447): lag = shift1-32..+32 step 16, mode 0
      tabled = (m_type != UWSPR_LINEAR) || (st.drift1 == 0.0f);
      if (tabled) k6_build_tables(tabB, st.f1, 0.05f, m_type, st.drift1, slmc);
      {
        const float f0v = st.f1 + (float)0 * 0.0f;
        const int L0 = st.shift1 - 32;
        const uint32_t mask = st.cknown ? 0x1bu : 0x1fu;
        if (tabled) k6_pass<K6_S3, true>(fb, a.fl, L0, UWSPR_NSYM, 20, mask, 2, f0v, 0.0f, st.drift1, 0.0f, m_type, slmc, launder(tabB), stage);
        else k6_pass<K6_S3, false>(fb, a.fl, L0, UWSPR_NSYM, 20, mask, 2, f0v, 0.0f, st.drift1, 0.0f, m_type, slmc, tabB, stage);
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
      // =========================== S4 (cc:449-452): f = f1 + ifreq 0.05, mode 1
      {
        const float fc = st.f1;
        const bool tabs_ok = tabled && (st.sync1 > -1e30f);
        const uint32_t mask = st.cknown ? 0x1bu : 0x1fu;
        float f0[5];
#pragma unroll
        for (int q = 0; q < 5; q++) f0[q] = fc + (float)(q - 2) * 0.05f;
        const int L0 = st.shift1;
        if (tabs_ok) k6_pass<K6_S4, true>(fb, a.fl, L0, UWSPR_NSYM, 16, mask, 0, fc, 0.05f, st.drift1, 0.0f, m_type, slmc, launder(tabB), stage);
        else k6_pass<K6_S4, false>(fb, a.fl, L0, UWSPR_NSYM, 16, mask, 0, fc, 0.05f, st.drift1, 0.0f, m_type, slmc, tabB, stage);
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
      // =========================== S5 (cc:457-482): the jiggered shifts, mode 2
      {
        // try idt has shift ii = 8 (+-)ceil(idt/2); lags ascend from L0 = shift1 - 64
        const int wq = uni(s_tq);
        const bool tab5 = tabled && wq >= 0;
        uint32_t mask = njig >= UWSPR_NJIG ? 0x1ffffu : ((1u << njig) - 1u);
        if (st.cknown) mask &= ~1u;   // try 0 repeats the S4 winner: its magnitudes are in pw
        const int L0 = st.shift1 - 64;
        if (mask) {
          if (tab5) k6_pass<K6_S5, true>(fb, a.fl, L0, UWSPR_NSYM, 24, mask, wq, st.f1, 0.0f, st.drift1, 0.0f, m_type, slmc, launder(tabB), stage);
          else k6_pass<K6_S5, false>(fb, a.fl, L0, UWSPR_NSYM, 24, mask, 0, st.f1, 0.0f, st.drift1, 0.0f, m_type, slmc, tabB, stage);
        }
        if (wv < K6_FOLDW) {
          for (int idt = wv; idt < UWSPR_NJIG; idt += K6_FOLDW) {
            if (idt < njig) {
              const float *slab = ((mask >> idt) & 1u) ? &stage[idt * K6_PSLAB] : pw;
              const float s = k6_fold<true>(slab, scr[wv], 50.0f, &o->symbols[idt][0], &o->jig_rms[idt]);
              int ii = (idt + 1) / 2;                      // cc:459-462
              if (idt % 2 == 1) ii = -ii;
              if (lane == 0) { o->jig_sync[idt] = s; o->jig_shift[idt] = st.shift1 + 8 * ii; }
            } else {
              for (int i = lane; i < UWSPR_NSYM; i += 64) o->symbols[idt][i] = 0;
              if (lane == 0) { o->jig_sync[idt] = 0.0f; o->jig_rms[idt] = 0.0f; o->jig_shift[idt] = 0; }
            }
          }
        }
      }
    } else {
      // not worth a try: the reference computes no soft symbols (cc:453-457)
      uint32_t *ow = reinterpret_cast<uint32_t *>(o);
      for (int e = tid + 5; e < (int)(sizeof(uwspr_demod_out) / 4); e += K6_THREADS) ow[e] = 0u;
    }
    if (tid == 0) {
      o->f1 = st.f1; o->drift1 = st.drift1; o->sync1 = st.sync1; o->shift1 = st.shift1;
      o->worth_a_try = st.worth; o->_pad[0] = 0; o->_pad[1] = 0;
      if (a.state) a.state[slot] = st;
    }
    if (a.pwin && st.worth)
      for (int e = tid; e < UWSPR_NSYM * 4; e += K6_THREADS) a.pwin[(size_t)slot * UWSPR_NSYM * 4 + e] = pw[e];
  }
}

void launch_sched_fused(uwspr_ctx *c, const float *frames, int B, const uwspr_candidate *cands,
                        const int32_t *npk, int cand_stride, int per_frame, uwspr_demod_out *out,
                        int njig) {
  const int nslots = B * per_frame;
  if (nslots <= 0) return;
  prof_scope ps(c, UWSPR_K_TONECORR, nslots, true);
  k6_args a;
  a.frames = (const float2 *)frames; a.fl = c->fc.fl; a.nframes = B;
  a.cands = cands; a.npk = npk; a.cand_stride = cand_stride; a.per_frame = per_frame; a.nslots = nslots;
  a.cf = (float)c->p.cf; a.reuse = c->reuse_centre ? 1 : 0;
  a.njig = njig;
  a.tabs = c->d_tabs; a.counter = c->d_counter; a.out = out; a.state = c->d_state; a.pwin = nullptr;
  const int grid = nslots < c->sched_grid ? nslots : c->sched_grid;
  // with no more candidates than CUs, keep them one per CU (a CU has room for two of these workgroups)
  const size_t pad = (nslots <= c->num_cus && !c->sched_nopad) ? 40 * 1024 : 0;
  (void)hipMemsetAsync(c->d_counter, 0, sizeof(int), c->stream);
  launch_timed(c, ps, k6_sched, dim3(grid), dim3(K6_THREADS), pad, a);
}

}  // namespace uwspr
