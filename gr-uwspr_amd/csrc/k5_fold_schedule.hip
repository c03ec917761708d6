// K5 -- per-hypothesis fold of the 162x4 tone magnitudes into the sync metric
// and the soft symbols, plus the tiny kernels that chain the refinement
// schedule stage to stage entirely in HBM.
//
// Reference:
//   fold           sync_and_demodulate_impl.cc:213-226 (totp, cmet, ss, ss/totp)
//   soft symbols   cc:216-224 and the mode-2 epilogue cc:240-254
//   schedule       sync_and_demodulate_impl::demodulate cc:403-482 (S0..S5)
//   best-of rule   cc:227-231 (strict >, first wins; -1e30 / 0 / 0.0 defaults)
#include "uwspr_internal.h"

#pragma clang fp contract(off)

namespace uwspr {

__device__ __constant__ uint32_t kPr3[6] = UWSPR_PR3_WORDS;
__device__ __forceinline__ bool pr3_rt(int i) { return (kPr3[i >> 5] >> (i & 31)) & 1u; }

// ------------------------------------------------------------------- K5 fold
__device__ __forceinline__ void k5_wave_fence() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}
// Lanes form, for launches with tens of thousands of hypotheses (the configs[2]
// sweep): one lane per hypothesis walks its 162 symbols in order -- the reference's
// accumulation order for totp / ss / fsum / f2sum -- with every lane busy, where the
// wave form below spends a whole wavefront on four serial sums.  The rows are 2.6 KB
// apart, so a wavefront brings 9 symbols of its 64 hypotheses into LDS at a time
// with 144-byte runs (576 float4, 9 per lane) and each lane then reads its own 9
// (row stride 36 dwords: conflict-free ds_read_b128); the soft-symbol bytes go
// back the same way (LDS image, then 10 368 contiguous bytes per wavefront).
constexpr int K5L_CH = 9;                       // symbols per staged chunk: 162 = 18 x 9
// pwin / per_slot (stage 5 of the schedule, else null): a hypothesis marked known (frame <= -2: try 0 repeating
// the stage-4 winner) is live and takes its 162 rows from pwin[h / per_slot] (the magnitudes carried for it)
__global__ __launch_bounds__(64) void k5_fold(const dev_hyp *__restrict__ hyps,
                                              const float4 *__restrict__ p, int H,
                                              float symfac, float *__restrict__ sync,
                                              uint8_t *__restrict__ symbols,
                                              const float4 *__restrict__ pwin, int per_slot) {
  UWSPR_SET_PRIO(UWSPR_SMALL_PRIO);
  __shared__ __align__(16) float4 tile[64 * K5L_CH];          // [hyp][9]
  __shared__ __align__(16) uint8_t bytes[64 * UWSPR_NSYM];    // [hyp][162]
  __shared__ int rowoff[64];                                  // float4 index of each hypothesis' row 0, or -1: in pwin
  __shared__ int rowslot[64];
  const int lane = threadIdx.x;
  const int h0 = blockIdx.x * 64;
  const int nh = min(64, H - h0);
  const int h = h0 + min(lane, nh - 1);
  const int fr = hyps[h].frame;
  const bool known = pwin != nullptr && fr <= -2;
  const bool live = lane < nh && (fr >= 0 || known);
  const bool soft = symbols != nullptr;
  const float4 *base = p + (size_t)h0 * UWSPR_NSYM;
  rowoff[lane] = known ? -1 : min(lane, nh - 1) * UWSPR_NSYM;
  rowslot[lane] = known ? h / per_slot : 0;

  auto stage = [&](int c) {   // chunk c of the 64 rows -> tile
    k5_wave_fence();
#pragma unroll
    for (int j = 0; j < K5L_CH; j++) {
      const int idx = lane + 64 * j;                 // 0..575: hyp = idx / 9, sym = idx % 9
      const int hy = idx / K5L_CH, sy = idx - hy * K5L_CH;
      const int ro = rowoff[hy];
      tile[idx] = ro >= 0 ? base[(size_t)ro + c * K5L_CH + sy]
                          : pwin[(size_t)rowslot[hy] * UWSPR_NSYM + c * K5L_CH + sy];
    }
    k5_wave_fence();
  };

  float ss = 0.0f, totp = 0.0f, fsum = 0.0f, f2sum = 0.0f;
  for (int c = 0; c < UWSPR_NSYM / K5L_CH; c++) {
    stage(c);
#pragma unroll
    for (int j = 0; j < K5L_CH; j++) {
      const float4 P = tile[lane * K5L_CH + j];
      const bool bit = pr3_rt(c * K5L_CH + j);
      totp = totp + P.x; totp = totp + P.y; totp = totp + P.z; totp = totp + P.w;  // cc:213
      const float cmet = (P.y + P.w) - (P.x + P.z);                               // cc:214
      ss = bit ? ss + cmet : ss - cmet;                                           // cc:215
      if (soft) {
        const float fs = bit ? P.w - P.y : P.z - P.x;                             // cc:219,222
        fsum = (float)((double)fsum + (double)fs / 162.0);                        // cc:243
        f2sum = (float)((double)f2sum + (double)(fs * fs) / 162.0);               // cc:244
      }
    }
  }
  if (lane < nh) sync[h] = live ? ieee_divf(ss, totp) : -1e30f;  // cc:226
  if (!soft) return;

  const float fac = ieee_sqrtf(f2sum - fsum * fsum);  // cc:246
  for (int c = 0; c < UWSPR_NSYM / K5L_CH; c++) {
    stage(c);
#pragma unroll
    for (int j = 0; j < K5L_CH; j++) {
      const float4 P = tile[lane * K5L_CH + j];
      const int i = c * K5L_CH + j;
      const bool bit = pr3_rt(i);
      float v = bit ? P.w - P.y : P.z - P.x;
      v = ieee_divf(symfac * v, fac);  // cc:248
      if (v > 127.0f) v = 127.0f;
      if (v < -128.0f) v = -128.0f;
      v = v + 128.0f;
      bytes[lane * UWSPR_NSYM + i] = (!live || v != v) ? (uint8_t)0 : (uint8_t)(int)v;  // cc:251 (NaN -> 0)
    }
  }
  k5_wave_fence();
  // 64 x 162 bytes are contiguous in the output: 4 bytes per lane and pass
  uint8_t *out = symbols + (size_t)h0 * UWSPR_NSYM;
  const int nbytes = nh * UWSPR_NSYM;
  for (int o = 4 * lane; o < nbytes; o += 256) {
    if (o + 4 <= nbytes && ((reinterpret_cast<uintptr_t>(out) + o) & 3) == 0) {
      *reinterpret_cast<uint32_t *>(out + o) = *reinterpret_cast<const uint32_t *>(&bytes[o]);
    } else {
      for (int q = 0; q < 4 && o + q < nbytes; q++) out[o + q] = bytes[o + q];
    }
  }
}

// Wave-per-hypothesis form of the same fold, for launches too small to fill the
// chip with one lane per hypothesis (the schedule stages: 5..17 hypotheses per
// candidate).  All loads/stores are coalesced; the per-symbol terms are formed
// by 162 lanes in parallel and only the order-sensitive running sums are
// serial: lane 0 adds the 648 magnitudes (cc:213), lane 1 the 162 signed
// metrics (cc:215), lanes 2/3 the binary64-stepped fsum / f2sum (cc:243-244).
constexpr int K5W_WAVES = 4;

struct k5_wave_lds {
  float4 str[2][UWSPR_NSYM];   // [0]: magnitudes, [1]: (0,0,0,+-cmet)
};
struct k5_wave_lds_soft {
  double q[2][UWSPR_NSYM];     // fs/162, fs*fs/162
};

// One wavefront folds hypothesis h; returns its sync metric in every lane.
// prow: the hypothesis' 162 tone magnitudes, or null = p + 162 h.  (A mode-2 hypothesis marked known --
// frame <= -2: try 0 of stage 5 repeating the stage-4 winner -- is folded from the magnitudes the schedule
// kept for it; see k5_fold_wave.)
template <bool SOFT>
__device__ __forceinline__ float fold_wave(const dev_hyp *__restrict__ hyps,
                                           const float4 *__restrict__ p, int h, float symfac,
                                           uint8_t *__restrict__ symbols, k5_wave_lds &L,
                                           k5_wave_lds_soft *Q, const float4 *__restrict__ prow = nullptr) {
  const int lane = threadIdx.x & 63;
  if (hyps[h].frame < 0 && !prow) {
    if (SOFT)
      for (int i = lane; i < UWSPR_NSYM; i += 64) symbols[(size_t)h * UWSPR_NSYM + i] = 0;
    return -1e30f;
  }
  const float4 *src = prow ? prow : p + (size_t)h * UWSPR_NSYM;
  float fs[3];
#pragma unroll
  for (int r = 0; r < 3; r++) {
    const int i = lane + 64 * r;
    fs[r] = 0.0f;
    if (i < UWSPR_NSYM) {
      const float4 P = src[i];
      const bool bit = pr3_rt(i);
      L.str[0][i] = P;
      const float cmet = (P.y + P.w) - (P.x + P.z);   // cc:214
      // ss -/+ cmet == ss + (-/+cmet); the three +0 terms leave ss unchanged
      L.str[1][i] = make_float4(0.0f, 0.0f, 0.0f, bit ? cmet : -cmet);
      fs[r] = bit ? P.w - P.y : P.z - P.x;            // cc:219,222
      if (SOFT) {
        Q->q[0][i] = (double)fs[r] / 162.0;                // cc:243
        Q->q[1][i] = (double)(fs[r] * fs[r]) / 162.0;      // cc:244
      }
    }
  }
  k5_wave_fence();
  float acc = 0.0f;
  if (lane < 2) {
    // lane 0: totp = (((totp+p0)+p1)+p2)+p3 per symbol (cc:213); lane 1: ss (cc:215)
    const float4 *st = L.str[lane];
#pragma unroll 9
    for (int i = 0; i < UWSPR_NSYM; i++) {
      const float4 P = st[i];
      acc = acc + P.x; acc = acc + P.y; acc = acc + P.z; acc = acc + P.w;
    }
  } else if (SOFT && lane < 4) {
    const double *q = Q->q[lane - 2];
#pragma unroll 9
    for (int i = 0; i < UWSPR_NSYM; i++) acc = (float)((double)acc + q[i]);
  }
  const float totp = __shfl(acc, 0), ss = __shfl(acc, 1);
  const float sync = ieee_divf(ss, totp);  // cc:226
  if (SOFT) {
    const float fsum = __shfl(acc, 2), f2sum = __shfl(acc, 3);
    const float fac = ieee_sqrtf(f2sum - fsum * fsum);  // cc:246
#pragma unroll
    for (int r = 0; r < 3; r++) {
      const int i = lane + 64 * r;
      if (i < UWSPR_NSYM) {
        float v = ieee_divf(symfac * fs[r], fac);  // cc:248
        if (v > 127.0f) v = 127.0f;
        if (v < -128.0f) v = -128.0f;
        v = v + 128.0f;
        symbols[(size_t)h * UWSPR_NSYM + i] = (v != v) ? (uint8_t)0 : (uint8_t)(int)v;
      }
    }
  }
  k5_wave_fence();  // LDS image may be reused by the caller
  return sync;
}

// pwin / per_slot (stage 5 of the schedule, else null): hypothesis h belongs to slot h / per_slot; one marked
// known (frame <= -2) repeats the stage-4 winner and is folded from pwin[slot] (its magnitudes, carried)
template <bool SOFT>
__global__ __launch_bounds__(64 * K5W_WAVES) void k5_fold_wave(
    const dev_hyp *__restrict__ hyps, const float4 *__restrict__ p, int H, float symfac,
    float *__restrict__ sync, uint8_t *__restrict__ symbols, const float4 *__restrict__ pwin, int per_slot) {
  UWSPR_SET_PRIO(UWSPR_SMALL_PRIO);
  __shared__ k5_wave_lds L[K5W_WAVES];
  __shared__ k5_wave_lds_soft Q[SOFT ? K5W_WAVES : 1];
  const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int h = blockIdx.x * K5W_WAVES + wv;
  if (h >= H) return;  // wave-uniform
  const float4 *prow = (pwin && hyps[h].frame <= -2) ? pwin + (size_t)(h / per_slot) * UWSPR_NSYM : nullptr;
  const float s = fold_wave<SOFT>(hyps, p, h, symfac, symbols, L[wv], SOFT ? &Q[wv] : nullptr, prow);
  if ((threadIdx.x & 63) == 0) sync[h] = s;
}

void launch_fold(uwspr_ctx *c, const dev_hyp *hyps, const float4 *p, int H, float *sync,
                 uint8_t *symbols, const float4 *pwin, int per_slot) {
  if (H <= 0) return;
  prof_scope ps(c, UWSPR_K_FOLD, H);
  // lanes form from 32768 hypotheses up (option "k5_lanes" = 0 / 1 forces one or the other).  The schedule's stage 5
  // (4352 tries for 256 slots) stays on the wave form: through the lanes form it is 68 wavefronts and a twentieth of
  // the instructions, but each walks 162 symbols serially -- 169 us instead of 64 on the lane's critical path,
  // -4 % frames/s under three streams (round-3 A/B)
  const int forced = c->opt[UWSPR_OPT_K5_LANES];
  const bool lanes_form = forced >= 0 ? forced != 0 : H >= 32768;
  if (per_slot < 1) per_slot = 1;
  if (!lanes_form) {
    dim3 g((H + K5W_WAVES - 1) / K5W_WAVES), b(64 * K5W_WAVES);
    if (symbols) hipLaunchKernelGGL(k5_fold_wave<true>, g, b, 0, c->stream, hyps, p, H, 50.0f, sync, symbols, pwin, per_slot);
    else hipLaunchKernelGGL(k5_fold_wave<false>, g, b, 0, c->stream, hyps, p, H, 50.0f, sync, symbols, pwin, per_slot);
  } else {
    hipLaunchKernelGGL(k5_fold, dim3((H + 63) / 64), dim3(64), 0, c->stream, hyps, p, H, 50.0f,
                       sync, symbols, pwin, per_slot);
  }
}

// ----------------------------------------------------- ABI hyp -> device hyp
// slmFrequencyDrift(m_nl, cf, t = 0), lib/slm.cc:36-73, in binary64 like the
// reference.  (`t` is read uninitialised at sync_and_demodulate_impl.cc:177-180;
// every observed build behaves as t = 0)
__device__ float slm_drift_t0(double V1, double V2, int p1, int p2, float cf) {
  const double q1 = V1 * 0.0 + (double)p1, q2 = V2 * 0.0 + (double)p2;
  const float sign = (float)(((q1 * V1 + q2 * V2) > 0) * 2 - 1);
  const double num = fabs(V1 * q1 + V2 * q2);
  const double den = sqrt(q1 * q1 + q2 * q2);
  if (den == 0) return 0.0f;
  return (float)((double)(-sign) * num / den * (double)cf / (double)1500.0f);
}

__global__ void k_prep_hyps(const uwspr_hyp *__restrict__ in, dev_hyp *__restrict__ out, int H,
                            float cf) {
  const int h = blockIdx.x * 256 + threadIdx.x;
  if (h >= H) return;
  const uwspr_hyp a = in[h];
  dev_hyp d;
  d.frame = a.frame; d.lag = a.lag; d.f0 = a.f0; d.drift = a.drift; d.m_type = a.m_type;
  d.slmc = (a.m_type == UWSPR_NONLINEAR) ? slm_drift_t0(a.V1, a.V2, a.p1, a.p2, cf) : 0.0f;
  out[h] = d;
}

void launch_prep_hyps(uwspr_ctx *c, const uwspr_hyp *abi, dev_hyp *out, int H) {
  if (H <= 0) return;
  prof_scope ps(c, UWSPR_K_SCHED, H);
  hipLaunchKernelGGL(k_prep_hyps, dim3((H + 255) / 256), dim3(256), 0, c->stream, abi, out, H,
                     (float)c->p.cf);
}

// ------------------------------------------------------------ the schedule
// Stage s consumes the metrics of the hypotheses stage s-1 generated and
// emits the next ones; hypotheses per candidate: S0 5, S1 5, S2 2, S3 5, S4 5,
// S5 17 (cc:409-482).  One thread per candidate slot.

// known: the hypothesis repeats the previous stage's winner -- frame <= -2 tells K4 and the
// fold to leave it alone (its metric is cand_state::csync)
__device__ inline void emit(dev_hyp *h, const cand_state &st, bool on, int lag, float f0,
                            float drift, bool known = false) {
  h->frame = on ? (known ? -2 - st.frame : st.frame) : -1;
  h->lag = lag; h->f0 = f0; h->drift = drift; h->slmc = st.slmc; h->m_type = st.m_type;
}

// the same hypotheses as a lag group (shared phasors in K4)
__device__ inline void emit_group(dev_grp *g, const cand_state &st, bool on, float f0, float drift,
                                  int hyp_base, const int *lags, int n, uint32_t hmap = 0x76543210u) {
  g->frame = on ? st.frame : -1;
  g->m_type = st.m_type; g->f0 = f0; g->drift = drift; g->slmc = st.slmc;
  g->nvalid = n; g->hyp_base = hyp_base; g->hmap = hmap;
  for (int l = 0; l < 8; l++) g->lag[l] = l < n ? lags[l] : lags[0];
}

// the same stage as a grid centre (K4 grid form: one symbol window per candidate and symbol,
// shared by the stage's frequencies / drifts)
__device__ inline void emit_centre(uwspr_candidate *ce, int32_t *cframe, const cand_state &st, bool on) {
  ce->freq = st.f1; ce->snr = 0.0f; ce->drift = 0.0f; ce->sync = 0.0f; ce->shift = st.shift1;
  ce->m_nonlinear.V1 = 0.0; ce->m_nonlinear.V2 = 0.0; ce->m_nonlinear.p1 = 0; ce->m_nonlinear.p2 = 0;
  if (st.m_type == UWSPR_LINEAR) { ce->m_type = UWSPR_LINEAR; ce->m_linear.drift = st.drift1; }
  else { ce->m_type = 2; ce->m_linear.drift = st.slmc; }   // internal: precomputed SLM constant
  *cframe = on ? st.frame : -1;
}

// cc:227-231 over a list scanned in order: strict >, defaults -1e30 / 0 / 0.0
struct best3 { float sync; int shift; float f; int q; };   // q: the winning hypothesis, -1 = none won
__device__ inline best3 best_of(const float *sy, const dev_hyp *hy, int n) {
  best3 b{-1e30f, 0, 0.0f, -1};
  for (int q = 0; q < n; q++)
    if (sy[q] > b.sync) { b.sync = sy[q]; b.shift = hy[q].lag; b.f = hy[q].f0; b.q = q; }
  return b;
}

// ---- phasor tables of the lag stages (staged form) ------------------------------------------------------
// When the per-symbol frequency does not depend on the symbol (drift == 0 or the straight-line model with
// t = 0: the reference's `fplast` cache hits for the same reason, cc:185) the sequence c[k], s[k] of cc:186-199 is
// the same for all 162 symbols of a (frequency, tone).  The lag kernels (k4_group, k4_ring) otherwise run that
// recurrence in every lane -- six instruction slots per sample step and wavefront; with the table in HBM/L2 they
// fetch 16 steps per chunk into LDS and read them from there.  Same values: the table IS the recurrence,
// computed once (constant kTwoPiDt5 etc. as in k4_tonecorr.hip).
constexpr double kTwoPiDt5 = 2.0 * 3.14159265358979323846 * (double)(float)(1.0 / 375.0);   // cc:146,188

// lanes 0 .. 4 nq - 1 of the calling wavefront: table (q0 + lane / 4) for tone lane & 3, frequency
// fc + (q - 2) fstep for q = qfirst + lane / 4
__device__ __forceinline__ void ptab_build(float2 *__restrict__ tabs, int nq, int qfirst, float fc, float fstep,
                                           int m_type, float drift, float slmc) {
  const int lane = threadIdx.x & 63;
  if (lane >= 4 * nq) return;
  const int qi = lane >> 2, tone = lane & 3, q = qfirst + qi;
  const float f0 = fc + (float)(q - 2) * fstep;                                       // cc:164
  const float fp = (m_type == UWSPR_LINEAR)
                       ? (float)((double)f0 + ((double)drift / 2.0) * ((double)(float)0 - 81.0) / 81.0)
                       : f0 + slmc;                                                   // cc:173 / cc:179
  const float delta = ((float)tone - 1.5f) * 1.46484375f;                             // cc:148
  double sn, cs;
  sincos(kTwoPiDt5 * (double)(fp + delta), &sn, &cs);                                 // cc:188-189
  const float cd = (float)cs, sd = (float)sn;
  float c = 1.0f, sv = 0.0f;
  float4 *t = reinterpret_cast<float4 *>(tabs + (size_t)qi * kPtabFloat2 + tone * 256);   // two steps per 16-byte store
  for (int k = 0; k < 128; k += 4) {   // eight steps per trip: the recurrence runs ahead of its stores
    float4 v[4];
#pragma unroll
    for (int j = 0; j < 4; j++) {
      v[j].x = c; v[j].y = sv;
      float nc = c * cd - sv * sd;   // cc:193-195
      float ns = c * sd + sv * cd;
      c = nc; sv = ns;
      v[j].z = c; v[j].w = sv;
      nc = c * cd - sv * sd;
      ns = c * sd + sv * cd;
      c = nc; sv = ns;
    }
#pragma unroll
    for (int j = 0; j < 4; j++) t[k + j] = v[j];
  }
}

// Table selector: -1 = none (every lane runs the recurrence), else the table index within the slot.  A table applies
// only if it was built and the frequency is bit for bit the one it was built for (set A: tabA_f + (q - 2) 0.25f,
// set B: tabB_f + (q - 2) 0.05f -- the expressions ptab_build and the stage emitters use).
__device__ __forceinline__ int ptab_index(const cand_state &st, int set, float f0, float drift, bool on) {
  if (!on) return -1;
  if (st.m_type == UWSPR_LINEAR && drift != 0.0f) return -1;
  if (!(set == 0 ? st.tabA_ok : st.tabB_ok)) return -1;
  const float fc = set == 0 ? st.tabA_f : st.tabB_f, step = set == 0 ? 0.25f : 0.05f;
  for (int q = 0; q < 5; q++)
    if (__float_as_int(fc + (float)(q - 2) * step) == __float_as_int(f0)) return (set == 0 ? 0 : kPtabSetB) + q;
  return -1;
}
// the same for a frequency stage (five hypotheses fc' + (q - 2) step): the set applies if the stage's centre is the
// set's centre -- returns the set's first table or -1
__device__ __forceinline__ int ptab_set(const cand_state &st, int set, float centre, float drift, bool on) {
  if (!on) return -1;
  if (st.m_type == UWSPR_LINEAR && drift != 0.0f) return -1;
  if (!(set == 0 ? st.tabA_ok : st.tabB_ok)) return -1;
  if (__float_as_int(centre) != __float_as_int(set == 0 ? st.tabA_f : st.tabB_f)) return -1;
  return set == 0 ? 0 : kPtabSetB;
}
// dev_grp::nvalid bits 16..23 of a lag group (the round-3 lag kernels): 0 = none, else 1 + table index
__device__ __forceinline__ int ptab_select(const cand_state &st, int set, float f0, float drift, bool on) {
  return ptab_index(st, set, f0, drift, on) + 1;
}

__global__ void k_sched_init(const uwspr_candidate *__restrict__ cands,
                             const int32_t *__restrict__ npk, int cand_stride, int B,
                             int per_frame, float cf, cand_state *__restrict__ state,
                             dev_hyp *__restrict__ hyps, dev_grp *__restrict__ grps, float2 *__restrict__ ptab) {
  UWSPR_SET_PRIO(UWSPR_SMALL_PRIO);
  // one wavefront per slot: every lane derives the same state (lane 0 writes it), lanes 0..19 build table set A
  const int slot = blockIdx.x;
  if (slot >= B * per_frame) return;
  const int b = slot / per_frame, j = slot - b * per_frame;
  cand_state st;
  const bool on = j < npk[b] && j < cand_stride;
  if (on) {
    const uwspr_candidate cnd = cands[(size_t)b * cand_stride + j];
    st.frame = b;
    st.m_type = cnd.m_type;
    st.slmc = (cnd.m_type == UWSPR_NONLINEAR)
                  ? slm_drift_t0(cnd.m_nonlinear.V1, cnd.m_nonlinear.V2, cnd.m_nonlinear.p1,
                                 cnd.m_nonlinear.p2, cf)
                  : 0.0f;
    st.f1 = cnd.freq;                                               // cc:404
    st.drift1 = (cnd.m_type == UWSPR_LINEAR) ? cnd.m_linear.drift : 0.0f;  // cc:405,373
    st.shift1 = cnd.shift;                                          // cc:406
    st.sync1 = cnd.sync;                                            // cc:407
  } else {
    st.frame = -1; st.m_type = 0; st.slmc = 0.0f; st.f1 = 0.0f; st.drift1 = 0.0f;
    st.shift1 = 0; st.sync1 = 0.0f;
  }
  st.worth = 0; st.driftp = 0.0f; st.driftm = 0.0f; st.csync = 0.0f; st.cknown = 0;
  // S0 (cc:409-415): mode 0, lag = shift1-128 .. shift1+128 step 64, f0 = f1 + 0*0.0f
  const float f0 = st.f1 + (float)0 * 0.0f;
  // table set A: f0 + {-2..2} 0.25 Hz (S0's lag sweep at the middle one, S1's five frequencies), when the frequency
  // does not depend on the symbol.  Built around f0 -- the frequency S0 runs at and S1 is centred on -- not f1: the
  // two differ in the sign of a zero.
  st.tabA_f = f0; st.tabB_f = 0.0f; st.tabB_ok = 0;
  st.tabA_ok = (ptab != nullptr && on && (st.m_type != UWSPR_LINEAR || st.drift1 == 0.0f)) ? 1 : 0;
  // (only the middle one of the five is read: S0; k4_fpack generates S1's phasors itself)
  if (st.tabA_ok) ptab_build(ptab + ((size_t)slot * kPtabPerSlot + 2) * kPtabFloat2, 1, 2, f0, 0.25f, st.m_type, st.drift1, st.slmc);
  if (threadIdx.x != 0) return;
  state[slot] = st;
  dev_hyp *h = hyps + (size_t)slot * 5;
  int lags[5];
  for (int q = 0; q < 5; q++) {
    lags[q] = st.shift1 - 128 + 64 * q;
    emit(&h[q], st, on, lags[q], f0, st.drift1);
  }
  emit_group(&grps[slot], st, on, f0, st.drift1, slot * 5, lags, 5);
  grps[slot].nvalid |= ptab_select(st, 0, f0, st.drift1, on) << 16;
}

template <int STAGE>
__device__ __forceinline__ void sched_step_body(int slot, cand_state *__restrict__ state,
                                                const dev_hyp *__restrict__ hin,
                                                const float *sync_of_slot,
                                                dev_hyp *__restrict__ hout,
                                                dev_grp *__restrict__ grps,
                                                uwspr_candidate *__restrict__ cent,
                                                int32_t *__restrict__ cframe, bool reuse, int team = 0,
                                                int njig = UWSPR_NJIG, int *wsrc = nullptr, bool tabs = false,
                                                bool fast = false) {
  // *wsrc (written by team 0): the input hypothesis whose tone magnitudes are now those of the state's
  // (f1, shift1, drift1) -- the stage winner -- or -1: the winner is the hypothesis that was marked known
  // (its magnitudes are the ones already kept) or nobody won.  The workgroup copies them to the slot's kept
  // row, so that try 0 of stage 5 (which repeats the stage-4 winner) is never correlated again.
  int ws = -1;
  // team (stage 5 only): lanes 0..19 of the first wavefront all derive the same new state, in
  // lockstep, and share the emission -- lane t < 17 writes try t, lanes 17..19 the lag groups
  cand_state st = state[slot];
  const bool live = st.frame >= 0;
  constexpr int NIN = STAGE == 1 ? 5 : STAGE == 2 ? 5 : STAGE == 3 ? 2 : STAGE == 4 ? 5 : 5;
  constexpr int NOUT = STAGE == 1 ? 5 : STAGE == 2 ? 2 : STAGE == 3 ? 5 : STAGE == 4 ? 5 : 17;
  const dev_hyp *hi = hin + (size_t)slot * NIN;
  const float *sy = sync_of_slot;
  // stage 5 with lazy tries (njig < 17): only tries idt < njig, packed njig per slot, no lag groups
  dev_hyp *ho = hout + (size_t)slot * (STAGE == 5 ? njig : NOUT);

  if (STAGE == 1) {
    // after S0 (mode 0) -> S1 (cc:416-419): mode 1, f = f1 + ifreq*0.25, lag = shift1
    if (live) { best3 b = best_of(sy, hi, 5); st.sync1 = b.sync; st.shift1 = b.shift; st.f1 = b.f; ws = b.q; }
    // q = 2 is (f1 + 0, shift1, drift1): the S0 hypothesis that just won (if one did)
    st.cknown = (reuse && live && st.sync1 > -1e30f) ? 1 : 0;
    st.csync = st.sync1;
    for (int q = 0; q < 5; q++)
      emit(&ho[q], st, live, st.shift1, st.f1 + (float)(q - 2) * 0.25f, st.drift1, q == 2 && st.cknown);
    emit_centre(&cent[slot], &cframe[slot], st, live);
  } else if (STAGE == 2) {
    // after S1 -> S2 (cc:423-433): linear only, drift1 +- 0.5 at (f1, shift1)
    if (live) {
      best3 b = best_of(sy, hi, 5); st.sync1 = b.sync; st.shift1 = b.shift; st.f1 = b.f;
      ws = (b.q == 2 && hi[2].frame <= -2) ? -1 : b.q;
    }
    st.cknown = (reuse && live && st.sync1 > -1e30f) ? 1 : 0;   // (f1, shift1, drift1) has metric sync1
    st.csync = st.sync1;
    const bool lin = live && st.m_type == UWSPR_LINEAR;
    st.driftp = (float)((double)st.drift1 + 0.5);
    st.driftm = (float)((double)st.drift1 - 0.5);
    const float f0 = st.f1 + (float)0 * 0.0f;
    emit(&ho[0], st, lin, st.shift1, f0, st.driftp);
    emit(&ho[1], st, lin, st.shift1, f0, st.driftm);
    emit_centre(&cent[slot], &cframe[slot], st, lin);
  } else if (STAGE == 3) {
    // after S2 (cc:434-441), gate (cc:443) -> S3 (cc:444-447): lag = shift1-32..+32 step 16
    if (live && st.m_type == UWSPR_LINEAR) {
      // each mode-1 call writes *f1/*shift1 back (cc:236-237): unchanged unless
      // its metric failed to beat -1e30 (NaN), where the defaults 0 / 0.0 land
      float syncp = -1e30f, syncm = -1e30f;
      if (sy[0] > syncp) syncp = sy[0]; else { st.f1 = 0.0f; st.shift1 = 0; st.cknown = 0; }
      if (sy[1] > syncm) syncm = sy[1]; else { st.f1 = 0.0f; st.shift1 = 0; st.cknown = 0; }
      if (syncp > st.sync1) { st.drift1 = st.driftp; st.sync1 = syncp; ws = 0; }
      else if (syncm > st.sync1) { st.drift1 = st.driftm; st.sync1 = syncm; ws = 1; }
    }
    st.worth = (live && st.sync1 > 0.10f) ? 1 : 0;
    // q = 2 is (f1, shift1, drift1): the winner of S1, or of S2 when a drift try beat it
    st.csync = st.sync1;
    const float f0 = st.f1 + (float)0 * 0.0f;
    int lags[5];
    for (int q = 0; q < 5; q++) {
      lags[q] = st.shift1 - 32 + 16 * q;
      emit(&ho[q], st, st.worth != 0, lags[q], f0, st.drift1, q == 2 && st.cknown);
    }
    emit_group(&grps[slot], st, st.worth != 0, f0, st.drift1, slot * 5, lags, 5);
    if (st.cknown) grps[slot].nvalid |= 0x100;   // lag slot 2 is known: K4 skips it
    // table set B is built around this f0 by the workgroup right after this (k5_fold_step<3>)
    st.tabB_f = f0;
    st.tabB_ok = (tabs && st.worth && (st.m_type != UWSPR_LINEAR || st.drift1 == 0.0f)) ? 1 : 0;
    grps[slot].nvalid |= ptab_select(st, 1, f0, st.drift1, st.worth != 0) << 16;
  } else if (STAGE == 4) {
    // after S3 -> S4 (cc:449-452): f = f1 + ifreq*0.05
    if (st.worth) {
      best3 b = best_of(sy, hi, 5); st.sync1 = b.sync; st.shift1 = b.shift; st.f1 = b.f;
      ws = (b.q == 2 && hi[2].frame <= -2) ? -1 : b.q;
    }
    st.cknown = (reuse && st.worth && st.sync1 > -1e30f) ? 1 : 0;   // q = 2 repeats the S3 winner
    st.csync = st.sync1;
    for (int q = 0; q < 5; q++)
      emit(&ho[q], st, st.worth != 0, st.shift1, st.f1 + (float)(q - 2) * 0.05f, st.drift1,
           q == 2 && st.cknown);
    emit_centre(&cent[slot], &cframe[slot], st, st.worth != 0);
  } else {
    // after S4 -> S5 (cc:457-468): 17 jiggered shifts, mode 2
    if (st.worth) {
      best3 b = best_of(sy, hi, 5); st.sync1 = b.sync; st.shift1 = b.shift; st.f1 = b.f;
      ws = (b.q == 2 && hi[2].frame <= -2) ? -1 : b.q;
    }
    // try 0 is (f1, shift1, drift1): the hypothesis that just won stage 4 (cc:457-463 with idt = 0) -- known
    // when one did: its magnitudes are the kept ones, K4 leaves it out, the fold reads the kept row
    // (option "fast_search": the kept magnitudes come from the fused-multiply-add stages; stage 5 is always the
    // reference's arithmetic, so try 0 is correlated again there)
    const bool known0 = reuse && !fast && st.worth && st.sync1 > -1e30f;
    st.cknown = known0 ? 1 : 0;
    if (team < njig) {
      const int idt = team;
      int ii = (idt + 1) / 2;
      if (idt % 2 == 1) ii = -ii;
      ii = 8 * ii;
      emit(&ho[idt], st, st.worth != 0, st.shift1 + ii, st.f1, st.drift1, idt == 0 && known0);
    }
    // The 17 jiggered shifts as three lag groups of 6, 6, 5 with ASCENDING, evenly spaced
    // lags (shift1 - 64 + 8 m, m = 0..16), so that a group's windows overlap and K4 can keep
    // them in one LDS ring.  Shift m is try idt = 2|m-8| - (m < 8): -64 -> 15, 0 -> 0, +64 -> 16.
    for (int g = 0; g < 3; g++) {
      if (team != UWSPR_NJIG + g) continue;
      const int n = g < 2 ? 6 : 5;
      int gl[6], idt0 = 99;
      int id[6];
      for (int l = 0; l < n; l++) {
        const int m = 6 * g + l, d = m - 8;
        id[l] = d == 0 ? 0 : (d < 0 ? -2 * d - 1 : 2 * d);
        gl[l] = st.shift1 + 8 * d;
        idt0 = id[l] < idt0 ? id[l] : idt0;
      }
      uint32_t hmap = 0;
      for (int l = 0; l < n; l++) hmap |= (uint32_t)(id[l] - idt0) << (4 * l);
      emit_group(&grps[slot * 3 + g], st, st.worth != 0 && njig >= UWSPR_NJIG, st.f1, st.drift1,
                 slot * UWSPR_NJIG + idt0, gl, n, hmap);
      if (g == 1 && known0) grps[slot * 3 + g].nvalid |= 0x100;   // its lag slot 2 (m = 8) is try 0: known
      grps[slot * 3 + g].nvalid |= ptab_select(st, 1, st.f1, st.drift1, st.worth != 0 && njig >= UWSPR_NJIG) << 16;
    }
  }
  if (team == 0) { state[slot] = st; if (wsrc) *wsrc = ws; }
}

// Fold of a candidate's NIN hypotheses (one wavefront each) fused with the
// schedule transition that consumes them: one workgroup per candidate slot.
// Option "fast_search", stages S0..S4: the per-hypothesis metric by wavefront shuffle-tree sums -- each lane
// adds its (up to three) symbols' terms, then six butterfly steps; a different summation ORDER than
// cc:213-215, so the metric agrees with the reference only to rounding (~1e-6 relative).
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o >= 1; o >>= 1) v += __shfl_xor(v, o);
  return v;
}
__device__ __forceinline__ float fold_wave_fast(const dev_hyp *__restrict__ hyps, const float4 *__restrict__ p, int h) {
  const int lane = threadIdx.x & 63;
  if (hyps[h].frame < 0) return -1e30f;
  float tp = 0.0f, sm = 0.0f;
#pragma unroll
  for (int r = 0; r < 3; r++) {
    const int i = lane + 64 * r;
    if (i < UWSPR_NSYM) {
      const float4 P = p[(size_t)h * UWSPR_NSYM + i];
      tp += (P.x + P.y) + (P.z + P.w);
      const float cmet = (P.y + P.w) - (P.x + P.z);
      sm += pr3_rt(i) ? cmet : -cmet;
    }
  }
  return ieee_divf(wave_sum(sm), wave_sum(tp));
}

template <int STAGE, bool FAST = false>
__global__ void k5_fold_step(cand_state *__restrict__ state, const dev_hyp *__restrict__ hin,
                             const float4 *__restrict__ p, float *__restrict__ sync,
                             dev_hyp *__restrict__ hout, dev_grp *__restrict__ grps,
                             uwspr_candidate *__restrict__ cent, int32_t *__restrict__ cframe, int nslots,
                             int reuse, int njig, float4 *__restrict__ pwin, float2 *__restrict__ ptab) {
  UWSPR_SET_PRIO(UWSPR_SMALL_PRIO);
  constexpr int NIN = STAGE == 3 ? 2 : 5;
  __shared__ k5_wave_lds L[FAST ? 1 : NIN];
  __shared__ float sy[NIN];
  __shared__ int s_wsrc;
  const int slot = blockIdx.x;
  const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  {
    const int q = wv;
    const int h = slot * NIN + q;
    // a hypothesis marked known (frame <= -2) repeats the previous winner: its metric is in the state
    float s;
    if (hin[h].frame <= -2) s = state[slot].csync;
    else if (FAST) s = fold_wave_fast(hin, p, h);
    else s = fold_wave<false>(hin, p, h, 50.0f, nullptr, L[FAST ? 0 : q], nullptr);
    if ((threadIdx.x & 63) == 0) { sy[q] = s; sync[h] = s; }
  }
  __syncthreads();
  if (STAGE == 5) {
    if (threadIdx.x < UWSPR_NJIG + 3)
      sched_step_body<STAGE>(slot, state, hin, sy, hout, grps, cent, cframe, reuse != 0, (int)threadIdx.x, njig, &s_wsrc, ptab != nullptr, FAST);
  } else if (threadIdx.x == 0) {
    sched_step_body<STAGE>(slot, state, hin, sy, hout, grps, cent, cframe, reuse != 0, 0, njig, &s_wsrc, ptab != nullptr, FAST);
  }
  __syncthreads();
  if (STAGE == 3 && ptab && threadIdx.x < 64) {   // table set B around the f1 the fine stages start from
    const cand_state st = state[slot];
    if (st.tabB_ok)
      ptab_build(ptab + ((size_t)slot * kPtabPerSlot + kPtabSetB) * kPtabFloat2, 5, 0, st.tabB_f, 0.05f, st.m_type, st.drift1, st.slmc);
  }
  // the stage winner's tone magnitudes become the slot's kept row (cf. k6_sched's keep_winner)
  const int ws = s_wsrc;
  if (pwin && ws >= 0)
    for (int e = threadIdx.x; e < UWSPR_NSYM; e += blockDim.x)
      pwin[(size_t)slot * UWSPR_NSYM + e] = p[((size_t)slot * NIN + ws) * UWSPR_NSYM + e];
}

// out[slot]: state + per-try sync / rms / shift / symbols (cc:465-475)
// slab != null (uwspr_pipeline_slabs): the workgroup of a frame's top candidate also writes the frame's gather slab
// (k_pack_slabs' layout) -- the tail of the slab is this slot's own (f1, drift1, sync1, shift1), nothing is read
// from another workgroup's record
__global__ void k_sched_finish(const cand_state *__restrict__ state,
                               const dev_hyp *__restrict__ h5, const float *__restrict__ sync5,
                               const uint8_t *__restrict__ sym5, uwspr_demod_out *__restrict__ out,
                               int nslots, int njig, const uwspr_candidate *__restrict__ cands,
                               const int32_t *__restrict__ npk, int maxfreqs, int per_frame, int K,
                               uint8_t *__restrict__ slab) {
  UWSPR_SET_PRIO(UWSPR_SMALL_PRIO);
  const int slot = blockIdx.x;
  if (slot >= nslots) return;
  const int tid = threadIdx.x;
  // the slot's njig x 162 symbol bytes are requested first, whether or not the slot is on (they exist either way: masked
  // below), so that they travel beside the state instead of behind it (round 5)
  constexpr int K5F_HALFWORDS = (UWSPR_NJIG * UWSPR_NSYM + 2) / 2, K5F_ROUNDS = (K5F_HALFWORDS + 255) / 256;
  uint16_t symv[K5F_ROUNDS];
  {
    const uint16_t *src = reinterpret_cast<const uint16_t *>(sym5 + (size_t)slot * njig * UWSPR_NSYM);   // (even offset)
    const int have = njig * UWSPR_NSYM;
#pragma unroll
    for (int j = 0; j < K5F_ROUNDS; j++) {
      const int e = tid + 256 * j;
      symv[j] = (e < K5F_HALFWORDS && 2 * e < have) ? src[e] : (uint16_t)0;
    }
  }
  const cand_state st = state[slot];
  uwspr_demod_out *o = out + slot;
  if (tid == 0) {
    o->f1 = st.f1; o->drift1 = st.drift1; o->sync1 = st.sync1; o->shift1 = st.shift1;
    o->worth_a_try = st.worth;
  }
  const bool on = st.frame >= 0 && st.worth;
  // lazy tries: the njig produced tries are packed njig per slot; the others read as zero (cc:457-490
  // would not have produced them either unless the earlier ones failed to decode)
  // the slot's soft symbols through LDS: one coalesced read, then the per-try rms (162 serial terms each, cc:471-474)
  // and the record's symbol block from there
  static_assert((UWSPR_NJIG * UWSPR_NSYM) % 2 == 0 && UWSPR_NSYM % 2 == 0, "two-byte accesses below");
  static_assert(offsetof(uwspr_demod_out, symbols) % 4 == 0 && sizeof(uwspr_demod_out) % 4 == 0 &&
                offsetof(uwspr_demod_out, _pad) == offsetof(uwspr_demod_out, symbols) + UWSPR_NJIG * UWSPR_NSYM &&
                sizeof(((uwspr_demod_out *)0)->_pad) == 2, "symbols + pad are written as 32-bit words below");
  __shared__ __align__(4) uint8_t sy_s[UWSPR_NJIG * UWSPR_NSYM + 2];
  const int nby = on ? njig * UWSPR_NSYM : 0;
  {
    uint16_t *dst = reinterpret_cast<uint16_t *>(sy_s);
#pragma unroll
    for (int j = 0; j < K5F_ROUNDS; j++) {
      const int e = tid + 256 * j;
      if (e < K5F_HALFWORDS) dst[e] = 2 * e < nby ? symv[j] : (uint16_t)0;
    }
  }
  __syncthreads();
  if (tid < UWSPR_NJIG) {
    const bool have = on && tid < njig;
    const size_t q = (size_t)slot * njig + tid;
    float rms = 0.0f;
    if (have) {
      float sq = 0.0f;
#pragma unroll 6
      for (int i = 0; i < UWSPR_NSYM; i++) {
        const float y = (float)((double)(float)sy_s[tid * UWSPR_NSYM + i] - 128.0);  // cc:471
        sq += y * y;
      }
      rms = (float)sqrt((double)sq / 162.0);  // cc:474
    }
    o->jig_sync[tid] = have ? sync5[q] : 0.0f;
    o->jig_rms[tid] = rms;
    o->jig_shift[tid] = have ? h5[q].lag : 0;
  }
  {   // symbols + the two pad bytes (zero) as 689 words
    uint32_t *dst = reinterpret_cast<uint32_t *>(&o->symbols[0][0]);
    const uint32_t *src = reinterpret_cast<const uint32_t *>(sy_s);
    for (int e = tid; e < (UWSPR_NJIG * UWSPR_NSYM + 2) / 4; e += blockDim.x) dst[e] = src[e];
  }
  if (slab != nullptr && slot % per_frame == 0) {
    const int b = slot / per_frame;
    const int slab_bytes = 16 + K * 48 + 16, words = slab_bytes / 4;
    uint32_t *so = reinterpret_cast<uint32_t *>(slab + (size_t)b * slab_bytes);
    const uint32_t *cw = reinterpret_cast<const uint32_t *>(cands + (size_t)b * maxfreqs);
    const int n = npk[b];
    for (int w = tid; w < words; w += blockDim.x) {
      uint32_t v = 0;
      if (w == 0) v = (uint32_t)n;
      else if (w >= 4 && w < 4 + K * 12) { const int k = (w - 4) / 12; v = (k < n && k < maxfreqs) ? cw[w - 4] : 0u; }
      else if (w >= 4 + K * 12) {
        const int t = w - 4 - K * 12;
        v = t == 0 ? __float_as_uint(st.f1) : t == 1 ? __float_as_uint(st.drift1) : t == 2 ? __float_as_uint(st.sync1) : (uint32_t)st.shift1;
      }
      so[w] = v;
    }
  }
}

// Per-frame slab for the multi-GPU gather: {npk, pad[3]} | candidate_t[K] |
// {f1, drift1, sync1, shift1} of the frame's top candidate.
__global__ void k_pack_slabs(const uwspr_candidate *__restrict__ cands,
                             const int32_t *__restrict__ npk, int maxfreqs,
                             const uwspr_demod_out *__restrict__ dout, int per_frame, int K,
                             uint8_t *__restrict__ slab, int B) {
  UWSPR_SET_PRIO(UWSPR_SMALL_PRIO);
  const int b = blockIdx.x;
  if (b >= B) return;
  const int slab_bytes = 16 + K * 48 + 16;
  uint32_t *out = reinterpret_cast<uint32_t *>(slab + (size_t)b * slab_bytes);
  const int words = slab_bytes / 4;
  const uint32_t *cw = reinterpret_cast<const uint32_t *>(cands + (size_t)b * maxfreqs);
  const uint32_t *dw = reinterpret_cast<const uint32_t *>(dout + (size_t)b * per_frame);
  const int n = npk[b];
  for (int w = threadIdx.x; w < words; w += blockDim.x) {
    uint32_t v = 0;
    if (w == 0) v = (uint32_t)n;
    else if (w >= 4 && w < 4 + K * 12) { const int k = (w - 4) / 12; v = (k < n && k < maxfreqs) ? cw[w - 4] : 0u; }
    else if (w >= 4 + K * 12) v = dw[w - 4 - K * 12];
    out[w] = v;
  }
}

void launch_pack_slabs(uwspr_ctx *c, const uwspr_candidate *cands, const int32_t *npk,
                       const uwspr_demod_out *dout, int per_frame, int K, uint8_t *slab, int B) {
  prof_scope ps(c, UWSPR_K_SCHED, B);
  hipLaunchKernelGGL(k_pack_slabs, dim3(B), dim3(128), 0, c->stream, cands, npk, c->fc.maxfreqs,
                     dout, per_frame, K, slab, B);
}

void launch_sched_init(uwspr_ctx *c, const uwspr_candidate *cands, const int32_t *npk,
                       int cand_stride, int B, int per_frame) {
  const int nslots = B * per_frame;
  prof_scope ps(c, UWSPR_K_SCHED, nslots);
  hipLaunchKernelGGL(k_sched_init, dim3(nslots), dim3(64), 0, c->stream, cands,
                     npk, cand_stride, B, per_frame, (float)c->p.cf, c->d_state, c->d_hyps, c->d_grps,
                     c->use_ptab ? c->d_ptab : nullptr);
}

// hyps of consecutive stages ping-pong between the two halves of d_hyps;
// stage s consumes the tone magnitudes K4 just wrote for the hypotheses of stage s-1
void launch_fold_step(uwspr_ctx *c, int stage, int nslots, int njig) {
  prof_scope ps(c, UWSPR_K_FOLD, (int64_t)nslots * (stage == 3 ? 2 : 5));
  dev_hyp *half0 = c->d_hyps, *half1 = c->d_hyps + (size_t)nslots * UWSPR_NJIG;
  dev_hyp *hin = (stage & 1) ? half0 : half1;
  dev_hyp *hout = (stage & 1) ? half1 : half0;
  dim3 g(nslots);
  const int reuse = c->reuse_centre ? 1 : 0;
  auto go = [&](auto kern, int threads) {
    hipLaunchKernelGGL(kern, g, dim3(threads), 0, c->stream, c->d_state, hin, c->d_p, c->d_sync, hout, c->d_grps,
                       c->d_cent, c->d_cent_frame, nslots, reuse, njig, (float4 *)c->d_pwin,
                       c->use_ptab ? c->d_ptab : nullptr);
  };
  if (c->fast_now) {
    switch (stage) {
      case 1: go(k5_fold_step<1, true>, 320); break;
      case 2: go(k5_fold_step<2, true>, 320); break;
      case 3: go(k5_fold_step<3, true>, 128); break;
      case 4: go(k5_fold_step<4, true>, 320); break;
      default: go(k5_fold_step<5, true>, 320); break;
    }
  } else {
    switch (stage) {
      case 1: go(k5_fold_step<1>, 320); break;
      case 2: go(k5_fold_step<2>, 320); break;
      case 3: go(k5_fold_step<3>, 128); break;
      case 4: go(k5_fold_step<4>, 320); break;
      default: go(k5_fold_step<5>, 320); break;
    }
  }
}

// lazy tries: try 0's magnitudes are what uwspr_demod_resume starts from -- already kept when try 0 was the
// known hypothesis, copied here when it was correlated (no stage-4 winner to repeat)
__global__ void k_keep_try0(const dev_hyp *__restrict__ h5, const float4 *__restrict__ p, float4 *__restrict__ pwin,
                            int njig, int nslots) {
  UWSPR_SET_PRIO(UWSPR_SMALL_PRIO);
  const int slot = blockIdx.x;
  if (slot >= nslots || h5[(size_t)slot * njig].frame < 0) return;   // known (<= -2) or dead (-1): nothing to copy
  for (int e = threadIdx.x; e < UWSPR_NSYM; e += blockDim.x)
    pwin[(size_t)slot * UWSPR_NSYM + e] = p[(size_t)slot * njig * UWSPR_NSYM + e];
}

void launch_keep_try0(uwspr_ctx *c, int nslots, int njig) {
  dev_hyp *h5 = c->d_hyps + (size_t)nslots * UWSPR_NJIG;
  hipLaunchKernelGGL(k_keep_try0, dim3(nslots), dim3(192), 0, c->stream, h5, c->d_p, (float4 *)c->d_pwin, njig, nslots);
}

void launch_sched_finish(uwspr_ctx *c, int nslots, int njig) {
  prof_scope ps(c, UWSPR_K_SCHED, nslots);
  // stage-5 hyps live in the half selected by (5 & 1) -> half1
  dev_hyp *h5 = c->d_hyps + (size_t)nslots * UWSPR_NJIG;
  uint8_t *slab = c->cands_from_fdr ? c->next_slab : nullptr;   // (uwspr_pipeline_slabs: this batch's slabs in the same launch)
  hipLaunchKernelGGL(k_sched_finish, dim3(nslots), dim3(256), 0, c->stream, c->d_state, h5,
                     c->d_sync, c->d_sym, c->cur_dout, nslots, njig, c->cur_cands, c->cur_npk, c->fc.maxfreqs,
                     c->sched_per_frame, c->next_slab_K, slab);
  if (slab) c->next_slab_done = true;
}

}  // namespace uwspr
