// k4_common.h -- arithmetic shared by the tone-correlation kernels (k4_tonecorr.hip, k4_pair.hip).
// Reference: sync_and_demodulate_impl.cc:186-211.
#pragma once

#include "uwspr_internal.h"

namespace uwspr {

// 2*pi*dt with dt = (float)(1/375) -- cc:146,188: `2*M_PI*dt*(fp+delta[j])`
constexpr double kTwoPiDt = 2.0 * 3.14159265358979323846 * (double)(float)(1.0 / 375.0);

__device__ __forceinline__ void wave_lds_fence() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// Equal-priority wavefronts are served oldest first: of the workgroups b, b + 256, b + 512 that the
// dispatcher puts on one CU, the youngest is starved while the others run and then finishes alone, two wavefronts on
// each SIMD.  Rotating the issue priority over the three from chunk to chunk lets them finish together.  `turn` counts
// the chunks; the class is the position of the workgroup in its CU's queue.  (s_setprio takes an immediate.)
// Used by k4_fpack.  In k4_lag0 and k4_ring it bought < 1 % alone and cost 0.1-0.3 % of the three-stream
// rate (profiles/r04_prio_rotation_ab.txt), so they stay at the default priority.
#ifndef UWSPR_K4_PRIO_ROTATION
#define UWSPR_K4_PRIO_ROTATION 1
#endif
__device__ __forceinline__ unsigned k4_prio_class() { return (blockIdx.x >> 8) % 3u; }
__device__ __forceinline__ void k4_rotate_priority(unsigned turn, unsigned prio_class) {
#if UWSPR_K4_PRIO_ROTATION
  switch ((turn + prio_class) % 3u) {   // wavefront-uniform
    case 0: __builtin_amdgcn_s_setprio(0); break;
    case 1: __builtin_amdgcn_s_setprio(1); break;
    default: __builtin_amdgcn_s_setprio(2); break;
  }
#endif
}

// The correlation step (cc:206-207) and the phasor rotation (cc:193-195).  FAST = false is the
// reference's arithmetic: every product and every sum rounded on its own, left to right.  FAST = true
// (the fast-search option, stages S0..S4 only, never the soft symbols) contracts them into fused
// multiply-adds: half the instructions, one rounding per product-sum -- metrics agree to ~1e-6 relative,
// which may move an argmax at a near-tie (tools/fast_search_eval.py measures how often).
template <bool FAST>
__device__ __forceinline__ void k4_mac(float &inp, float &quad, float xx, float xy, float c, float s) {
#pragma clang fp contract(off)
  if (FAST) {
    inp = __builtin_fmaf(xy, s, __builtin_fmaf(xx, c, inp));
    quad = __builtin_fmaf(xy, c, __builtin_fmaf(-xx, s, quad));
  } else {
    inp = (inp + xx * c) + xy * s;     // cc:206
    quad = (quad - xx * s) + xy * c;   // cc:207
  }
}
template <bool FAST>
__device__ __forceinline__ void k4_rot(float &c, float &s, float cd, float sd) {
#pragma clang fp contract(off)
  float nc, ns;
  if (FAST) {
    nc = __builtin_fmaf(c, cd, -(s * sd));
    ns = __builtin_fmaf(c, sd, s * cd);
  } else {
    nc = c * cd - s * sd;              // cc:193-195
    ns = c * sd + s * cd;
  }
  c = nc; s = ns;
}

// per-symbol frequency of a hypothesis (cc:170-183; the nonlinear model with t = 0)
__device__ __forceinline__ float k4_symbol_freq(int m_type, float f0, float drift, float slmc, int sym) {
  if (m_type == UWSPR_LINEAR)
    return (float)((double)f0 + ((double)drift / 2.0) * ((double)(float)sym - 81.0) / 81.0);   // cc:173
  return f0 + slmc;                                                                            // cc:179
}
// (cos, sin) of the per-sample phase step of a tone, in binary64 rounded to binary32 (cc:188-189)
__device__ __forceinline__ void k4_tone_step(float fp, int tone, float &cd, float &sd) {
  const float delta = ((float)tone - 1.5f) * 1.46484375f;   // cc:148: {-1.5,-0.5,0.5,1.5} * (float)(375/256), exact
  double sn, cs;
  sincos(kTwoPiDt * (double)(fp + delta), &sn, &cs);
  cd = (float)cs;
  sd = (float)sn;
}

// Stage S2's two tries of a slot (hyps 2 s: drift1 + 0.5, 2 s + 1: drift1 - 0.5; cc:425, 429) mirror each other when
// the candidate had no drift: same frame, lag and f0, opposite drifts.  Then symbol i of the first has, bit for bit,
// the tone frequency of symbol 162 - i of the second (k4_pair.hip), and k4_dpair computes both with one recurrence.
__device__ __forceinline__ bool k4_drift_pair(const dev_hyp &p, const dev_hyp &m, int nframes) {
  return p.frame >= 0 && p.frame < nframes && m.frame == p.frame && p.m_type == UWSPR_LINEAR && m.m_type == UWSPR_LINEAR &&
         p.lag == m.lag && __float_as_uint(p.f0) == __float_as_uint(m.f0) && p.drift == -m.drift;
}

}  // namespace uwspr
