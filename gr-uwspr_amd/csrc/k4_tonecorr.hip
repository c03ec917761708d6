// K4 -- the fine-grid correlation sweep: for every (freq, time-lag, drift)
// hypothesis and every one of the 162 symbols, the non-coherent correlation of
// the 256-sample symbol window against the four 4-FSK tone phasors.
//
// Reference: sync_and_demodulate_impl::sync_and_demodulate hot loop 3,
// lib/sync_and_demodulate_impl.cc:167-212 (per-symbol frequency cc:170-183,
// phasor recurrence cc:186-199, correlation cc:200-211).
//
// Mapping: the (hypothesis, symbol) pairs are flattened, g = 162*h + i.  A lane
// owns one pair and T of its 4 tones (T = 1, 2 or 4; a wavefront covers 16*T
// pairs).  The lane derives its symbol frequency in binary64 exactly as
// cc:173/cc:179 do, takes cos/sin of the per-sample phase step in binary64
// (cc:188-189), then walks the 256 samples in order, advancing its tone phasors
// by the reference's binary32 rotation recurrence (cc:193-195) and accumulating
// inp/quad in the reference's operand order (cc:206-207).  Every accumulator
// sees the reference's exact sequence of binary32 operations (no FMA, no tree
// reduction), so p[] is bit-identical.  T trades latency for overhead: T=1 gives
// 4x the wavefronts and a 4x shorter serial chain (the 5..17-hypothesis stages of
// the refinement schedule are latency-bound), T=4 amortises the sample reads
// over all four tones (the 200-hypothesis sweep is throughput-bound).
//
// Samples reach the lanes through LDS: a wavefront's symbol windows are runs of
// 2 KB in HBM; per 16-sample chunk the wave loads them cooperatively (16 lanes x
// 8 B = one 128-B run per window) into a per-wave LDS image whose rows are
// padded to 136 B, so the per-lane column reads (ds_read_b64) are bank-conflict
// free (lanes of one pair read the same address: broadcast).  The next chunk's
// global loads are in flight while the current chunk is computed.
//
// Work per pair-sample: 32 correlation + 24 phasor binary32 ops = 56 VALU ops;
// the kernel is FP32-VALU bound; HBM sees each frame about once
// (L2 / Infinity Cache serve the re-reads by the other hypotheses of the frame).
#include <stdlib.h>

#include "uwspr_internal.h"

#pragma clang fp contract(off)

namespace uwspr {

constexpr int K4_WAVES = 4;
constexpr int K4_ROWDW = 34;  // dwords per staged row: 16 samples x 8 B + 8 B pad

// 2*pi*dt with dt = (float)(1/375) -- cc:146,188: `2*M_PI*dt*(fp+delta[j])`
constexpr double kTwoPiDt = 2.0 * 3.14159265358979323846 * (double)(float)(1.0 / 375.0);

__device__ __forceinline__ void wave_lds_fence() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

template <int T>
__global__ __launch_bounds__(64 * K4_WAVES) void k4_tonecorr(
    const float2 *__restrict__ frames, int fl, int nframes, const dev_hyp *__restrict__ hyps,
    int H, float *__restrict__ p_out) {
  constexpr int PPW = 16 * T;        // pairs per wavefront
  constexpr int LPP = 4 / T;         // lanes per pair
  constexpr int NLD = PPW / 4;       // cooperative loads per lane per chunk
  __shared__ float lds_all[K4_WAVES][PPW * K4_ROWDW];

  const int lane = threadIdx.x & 63;
  const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  float *lds = lds_all[wv];

  const long long total = (long long)H * UWSPR_NSYM;
  const long long g0 = ((long long)blockIdx.x * K4_WAVES + wv) * PPW;
  if (g0 >= total) return;  // wave-uniform; no workgroup barrier is used below

  // A wave's pairs span at most two hypotheses (162 > 64 >= PPW).
  const int hA = (int)(g0 / UWSPR_NSYM);
  const int iA0 = (int)(g0 - (long long)hA * UWSPR_NSYM);
  const int sb = min(PPW, UWSPR_NSYM - iA0);  // pairs < sb belong to hA
  dev_hyp A = hyps[hA];
  dev_hyp Bh = hyps[min(hA + 1, H - 1)];
  const bool okA = A.frame >= 0 && A.frame < nframes;
  const bool okB = (hA + 1 < H) && Bh.frame >= 0 && Bh.frame < nframes;
  // hypotheses that are skipped still own LDS rows: point them at safe samples
  if (!okA) { A.frame = 0; A.lag = 1 - 256 * iA0; }
  if (!okB) { Bh.frame = 0; Bh.lag = 1; }

  const int pr = lane / LPP;           // this lane's pair within the wave
  const int tone0 = (lane % LPP) * T;  // first of its T tones
  const bool mineA = pr < sb;
  const int own_i = mineA ? iA0 + pr : pr - sb;
  const bool own_ok = mineA ? okA : (okB && (g0 + pr) < total);
  const int own_nb = (mineA ? A.lag : Bh.lag) + 256 * own_i;  // first sample index
  const bool interior = __all((own_nb > 0) && (own_nb + 255 < fl));

  // ---- per-symbol tone phasor steps (binary64 angle, cc:173-189) ----------
  float cd[T], sd[T];
  {
    const dev_hyp &hy = mineA ? A : Bh;
    float fp;
    if (hy.m_type == UWSPR_LINEAR) {
      fp = (float)((double)hy.f0 +
                   ((double)hy.drift / 2.0) * ((double)(float)own_i - 81.0) / 81.0);
    } else {
      fp = hy.f0 + hy.slmc;
    }
#pragma unroll
    for (int j = 0; j < T; j++) {
      // delta[] = {-1.5,-0.5,0.5,1.5} * (float)(375/256), cc:148 (exact in binary32)
      const float delta = ((float)(tone0 + j) - 1.5f) * 1.46484375f;
      const double ang = kTwoPiDt * (double)(fp + delta);
      double sn, cs;
      sincos(ang, &sn, &cs);
      cd[j] = (float)cs;
      sd[j] = (float)sn;
    }
  }

  // ---- cooperative loader geometry ---------------------------------------
  // load t (0..NLD-1) of a chunk: window seg = 4t + lane/16, sample kk = lane%16
  const int kk = lane & 15;
  const int segq = lane >> 4;
  const float2 *src[NLD];  // sample 0 (+kk) of that window, fast path
#pragma unroll
  for (int t = 0; t < NLD; t++) {
    const int seg = 4 * t + segq;
    const bool sA = seg < sb;
    const int fr = sA ? A.frame : Bh.frame;
    const int nb = sA ? A.lag + 256 * (iA0 + seg) : Bh.lag + 256 * (seg - sb);
    src[t] = frames + ((long long)fr * fl + nb + kk);
  }

  float2 stage[NLD];
  auto load_chunk = [&](int c) {
    if (interior) {
#pragma unroll
      for (int t = 0; t < NLD; t++) stage[t] = src[t][16 * c];
    } else {
#pragma unroll
      for (int t = 0; t < NLD; t++) {
        const int seg = 4 * t + segq;
        const bool sA = seg < sb;
        const long long fb = (long long)(sA ? A.frame : Bh.frame) * fl;
        const int n = (sA ? A.lag + 256 * (iA0 + seg) : Bh.lag + 256 * (seg - sb)) + 16 * c + kk;
        const bool inr = (n > 0) && (n < fl);  // cc:205, sample 0 excluded
        float2 v = frames[fb + min(max(n, 0), fl - 1)];
        // a skipped sample contributes nothing: x*c with x = 0 leaves inp/quad unchanged
        stage[t] = inr ? v : make_float2(0.0f, 0.0f);
      }
    }
  };

  float c[T], s[T], inp[T], quad[T];
#pragma unroll
  for (int j = 0; j < T; j++) { c[j] = 1.0f; s[j] = 0.0f; inp[j] = 0.0f; quad[j] = 0.0f; }

  load_chunk(0);
  for (int ch = 0; ch < 16; ch++) {
    wave_lds_fence();  // previous chunk's reads are done before rows are rewritten
#pragma unroll
    for (int t = 0; t < NLD; t++)
      *reinterpret_cast<float2 *>(&lds[(4 * t + segq) * K4_ROWDW + 2 * kk]) = stage[t];
    wave_lds_fence();
    if (ch < 15) load_chunk(ch + 1);
#pragma unroll
    for (int k = 0; k < 16; k++) {
      const float2 x = *reinterpret_cast<const float2 *>(&lds[pr * K4_ROWDW + 2 * k]);
#pragma unroll
      for (int j = 0; j < T; j++) {
        // cc:206-207, left to right
        inp[j] = (inp[j] + x.x * c[j]) + x.y * s[j];
        quad[j] = (quad[j] - x.x * s[j]) + x.y * c[j];
        // cc:193-195
        const float nc = c[j] * cd[j] - s[j] * sd[j];
        const float ns = c[j] * sd[j] + s[j] * cd[j];
        c[j] = nc; s[j] = ns;
      }
    }
  }

  if (g0 + pr < total) {
    float *out = p_out + (g0 + pr) * 4 + tone0;  // p[pair][tone]
#pragma unroll
    for (int j = 0; j < T; j++) {
      const float pj = ieee_sqrtf(inp[j] * inp[j] + quad[j] * quad[j]);  // cc:211
      out[j] = own_ok ? pj : 0.0f;
    }
  }
}

static int k4_choose_t(long long pairs) {
  static int forced = -1;
  if (forced < 0) {
    const char *e = getenv("UWSPR_K4_T");
    forced = e ? atoi(e) : 0;
  }
  if (forced == 1 || forced == 2 || forced == 4) return forced;
  // enough wavefronts to fill 256 CUs x 4 SIMDs several times over -> amortise
  if (pairs >= 4LL * 1024 * 1024) return 4;
  if (pairs >= 1024 * 1024) return 2;
  return 1;
}

void launch_tonecorr(uwspr_ctx *c, const float *frames, int B, const dev_hyp *hyps, int H,
                     float4 *p) {
  if (H <= 0) return;
  prof_scope ps(c, UWSPR_K_TONECORR, H);
  const long long total = (long long)H * UWSPR_NSYM;
  const int T = k4_choose_t(total);
  const long long waves = (total + 16 * T - 1) / (16 * T);
  const unsigned blocks = (unsigned)((waves + K4_WAVES - 1) / K4_WAVES);
  const float2 *fr = (const float2 *)frames;
  float *po = (float *)p;
  dim3 blk(64 * K4_WAVES);
  if (T == 1) hipLaunchKernelGGL(k4_tonecorr<1>, dim3(blocks), blk, 0, c->stream, fr, c->fc.fl, B, hyps, H, po);
  else if (T == 2) hipLaunchKernelGGL(k4_tonecorr<2>, dim3(blocks), blk, 0, c->stream, fr, c->fc.fl, B, hyps, H, po);
  else hipLaunchKernelGGL(k4_tonecorr<4>, dim3(blocks), blk, 0, c->stream, fr, c->fc.fl, B, hyps, H, po);
}

}  // namespace uwspr
