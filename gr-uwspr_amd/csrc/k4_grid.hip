// K4, grid form -- the (freq, drift, lag) sweep of BASELINE configs[2] (uwspr_sync_grid).
//
// Reference: sync_and_demodulate_impl::sync_and_demodulate hot loop 3, lib/sync_and_demodulate_impl.cc:167-212.
#include <string.h>

#include <algorithm>

#include "k4_common.h"

#pragma clang fp contract(off)

// ---------------------------------------------------------------------------
// Grid form: the (freq, drift, lag) sweep around one centre per frame
// (BASELINE configs[2]).  All hypotheses of a frame and symbol read the same
// samples, and hypotheses that differ only in lag share their tone phasors, so:
//  * a wavefront takes 16 consecutive (symbol, freq x drift combination) pairs of
//    ONE frame (symbol-major order => at most a few distinct symbols per wave) and
//    loads each needed symbol window [lagmin + 256 i, lagmax + 256 i + 256) ONCE,
//    whole, into LDS with coalesced 8-byte loads (the n > 0 && n < np test of
//    cc:205 is applied here: skipped samples become zeros, which leave inp/quad
//    unchanged);
//  * a lane owns one pair and one tone: one phasor recurrence, NL accumulator
//    pairs (one per lag), reading sample k + (lag_l - lagmin) of its window;
//  * after the window load the 256-sample walk touches no global memory at all.
// Arithmetic per accumulator is the reference's sequence (cc:193-195, 206-207).
namespace uwspr {

struct grid_args {
  int nf, ndrift, ncombo, nlag_total, lag_base;  // this launch covers lags lag_base..lag_base+NL-1
  int wlen, wstride, wmax;                       // window length / LDS stride (samples), windows per wave
  int dlag_min;                                  // min over ALL lag offsets of the launch block
  int off[8];                                    // dlag[l] - dlag_min for the NL lags
  int nvalid;                                    // lags in use (<= NL)
  float df[32], ddrift[32];
};

constexpr int K4GR_WAVES = 4;

template <int NL>
__global__ __launch_bounds__(64 * K4GR_WAVES) void k4_grid(
    const float2 *__restrict__ frames, int fstride, int np, int nframes,
    const uwspr_candidate *__restrict__ centres, const int32_t *__restrict__ cframe,
    grid_args ga, float cf, float *__restrict__ p_out) {
  extern __shared__ __align__(16) float lds_dyn[];
  const int lane = threadIdx.x & 63;
  const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  float2 *win = reinterpret_cast<float2 *>(lds_dyn) + (size_t)wv * ga.wmax * ga.wstride;

  const int b = blockIdx.y;                      // centre index
  const int npairs = UWSPR_NSYM * ga.ncombo;
  const int nwv = blockDim.x >> 6;
  const int q0 = (blockIdx.x * nwv + wv) * 16;
  if (q0 >= npairs) return;  // wave-uniform
  const int frame = cframe ? cframe[b] : b;     // several centres may share a frame
  if (frame < 0 || frame >= nframes) return;    // dead centre: its hypotheses are marked skipped
  const uwspr_candidate ce = centres[b];
  const int i_first = q0 / ga.ncombo;
  const int i_last = min(q0 + 15, npairs - 1) / ga.ncombo;
  const int nwin = i_last - i_first + 1;  // <= ga.wmax by construction

  // ---- load the symbol windows (whole) ------------------------------------
  {
    const float2 *fb = frames + (long long)frame * fstride;
    const int n0 = ce.shift + ga.dlag_min + 256 * i_first;
    const int tot = nwin * ga.wlen;
    for (int e = lane; e < tot; e += 64) {
      const int w = e / ga.wlen, r = e - w * ga.wlen;
      const int n = n0 + 256 * w + r;
      const bool inr = (n > 0) && (n < np);  // cc:205, sample 0 excluded
      const float2 v = fb[min(max(n, 0), np - 1)];
      win[w * ga.wstride + r] = inr ? v : make_float2(0.0f, 0.0f);
    }
  }
  wave_lds_fence();

  const int pr = lane >> 2;
  const int tone = lane & 3;
  const int q = q0 + pr;
  const bool ok = q < npairs;
  const int i = ok ? q / ga.ncombo : i_first;
  const int cb = ok ? q - i * ga.ncombo : 0;
  const int fi = cb / ga.ndrift, di = cb - fi * ga.ndrift;

  // ---- this lane's tone phasor step (binary64 angle, cc:164-189) -----------
  float cd, sd;
  {
    const float f0 = ce.freq + ga.df[fi];  // cc:164
    float fp;
    if (ce.m_type == UWSPR_LINEAR) {
      const float drift = ce.m_linear.drift + ga.ddrift[di];
      fp = (float)((double)f0 + ((double)drift / 2.0) * ((double)(float)i - 81.0) / 81.0);
    } else if (ce.m_type == 2) {
      fp = f0 + ce.m_linear.drift;  // internal: nonlinear centre with its SLM constant precomputed
    } else {
      // slmFrequencyDrift(m_nl, cf, t = 0), lib/slm.cc:36-73
      const double q1 = (double)ce.m_nonlinear.p1, q2 = (double)ce.m_nonlinear.p2;
      const double V1 = ce.m_nonlinear.V1, V2 = ce.m_nonlinear.V2;
      const float sign = (float)(((q1 * V1 + q2 * V2) > 0) * 2 - 1);
      const double num = fabs(V1 * q1 + V2 * q2), den = sqrt(q1 * q1 + q2 * q2);
      const float slmc = den == 0 ? 0.0f : (float)((double)(-sign) * num / den * (double)cf / (double)1500.0f);
      fp = f0 + slmc;
    }
    const float delta = ((float)tone - 1.5f) * 1.46484375f;
    double sn, cs;
    sincos(kTwoPiDt * (double)(fp + delta), &sn, &cs);
    cd = (float)cs;
    sd = (float)sn;
  }

  float c = 1.0f, s = 0.0f;
  float inp[NL], quad[NL];
#pragma unroll
  for (int l = 0; l < NL; l++) { inp[l] = 0.0f; quad[l] = 0.0f; }

  // dword offsets of the NL lag columns inside this wave's windows (integers, so the reads stay
  // ds_read with compile-time column offsets); the reads of step k+1 are issued before the
  // arithmetic of step k and pinned there
  const float *winf = reinterpret_cast<const float *>(win);
  int ao[NL];
#pragma unroll
  for (int l = 0; l < NL; l++) ao[l] = 2 * ((i - i_first) * ga.wstride + ga.off[l]);
  for (int k0 = 0; k0 < 256; k0 += 16) {
    float2 xc[NL], xn[NL];
#pragma unroll
    for (int l = 0; l < NL; l++) { xc[l] = *reinterpret_cast<const float2 *>(&winf[ao[l]]); xn[l] = xc[l]; }
#pragma unroll
    for (int k = 0; k < 16; k++) {
      if (k < 15) {
#pragma unroll
        for (int l = 0; l < NL; l++) xn[l] = *reinterpret_cast<const float2 *>(&winf[ao[l] + 2 * (k + 1)]);
      }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int l = 0; l < NL; l++) {
        inp[l] = (inp[l] + xc[l].x * c) + xc[l].y * s;     // cc:206
        quad[l] = (quad[l] - xc[l].x * s) + xc[l].y * c;   // cc:207
      }
      const float nc = c * cd - s * sd;            // cc:193-195
      const float ns = c * sd + s * cd;
      c = nc; s = ns;
#pragma unroll
      for (int l = 0; l < NL; l++) xc[l] = xn[l];
    }
#pragma unroll
    for (int l = 0; l < NL; l++) ao[l] += 32;      // next 16 samples
  }

  if (ok) {
#pragma unroll
    for (int l = 0; l < NL; l++) {
      if (l < ga.nvalid) {
        const long long hyp = ((long long)b * ga.ncombo + cb) * ga.nlag_total + ga.lag_base + l;
        p_out[(hyp * UWSPR_NSYM + i) * 4 + tone] = ieee_sqrtf(inp[l] * inp[l] + quad[l] * quad[l]);  // cc:211
      }
    }
  }
}

// flat list of the grid's hypotheses (for the fold, and as the definition of the order)
__global__ void k_grid_hyps(const uwspr_candidate *__restrict__ centres, grid_args ga, int nlag,
                            const int *__restrict__ dlag, float cf, dev_hyp *__restrict__ out, int B) {
  const long long h = (long long)blockIdx.x * 256 + threadIdx.x;
  const long long per = (long long)ga.ncombo * nlag;
  if (h >= per * B) return;
  const int b = (int)(h / per);
  const int r = (int)(h - (long long)b * per);
  const int cb = r / nlag, l = r - cb * nlag;
  const int fi = cb / ga.ndrift, di = cb - fi * ga.ndrift;
  const uwspr_candidate ce = centres[b];
  dev_hyp d;
  d.frame = b; d.lag = ce.shift + dlag[l]; d.f0 = ce.freq + ga.df[fi]; d.m_type = ce.m_type;
  d.drift = (ce.m_type == UWSPR_LINEAR) ? ce.m_linear.drift + ga.ddrift[di] : 0.0f;
  d.slmc = 0.0f;
  if (ce.m_type == UWSPR_NONLINEAR) {  // slmFrequencyDrift(m_nl, cf, t = 0), lib/slm.cc:36-73
    const double q1 = (double)ce.m_nonlinear.p1, q2 = (double)ce.m_nonlinear.p2;
    const double V1 = ce.m_nonlinear.V1, V2 = ce.m_nonlinear.V2;
    const float sign = (float)(((q1 * V1 + q2 * V2) > 0) * 2 - 1);
    const double num = fabs(V1 * q1 + V2 * q2), den = sqrt(q1 * q1 + q2 * q2);
    d.slmc = den == 0 ? 0.0f : (float)((double)(-sign) * num / den * (double)cf / (double)1500.0f);
  }
  out[h] = d;
}

template <int NL>
static void launch_grid_t(uwspr_ctx *c, prof_scope &ps, const float2 *fr, int nframes, int ncentres,
                          const uwspr_candidate *centres, const int32_t *cframe, const grid_args &ga,
                          int wpw, float *po) {
  const int npairs = UWSPR_NSYM * ga.ncombo;
  const int waves = (npairs + 15) / 16;
  dim3 grid((waves + wpw - 1) / wpw, ncentres);
  const size_t lds = (size_t)wpw * ga.wmax * ga.wstride * sizeof(float2);
  launch_timed(c, ps, k4_grid<NL>, grid, dim3(64 * wpw), lds, fr, c->fstride, c->np, nframes, centres,
                     cframe, ga, (float)c->p.cf, po);
}

// waves per workgroup such that the symbol windows fit 64 KB of LDS (0 = does not fit)
static int grid_waves_per_wg(const grid_args &ga) {
  for (int w = K4GR_WAVES; w >= 1; w >>= 1)
    if ((size_t)w * ga.wmax * ga.wstride * sizeof(float2) <= 64 * 1024) return w;
  return 0;
}

// One lag block (<= 8 lags) of a grid around `ncentres` centres; false = does not fit LDS.
bool launch_grid_block(uwspr_ctx *c, const float *frames, int nframes, int ncentres,
                       const uwspr_candidate *centres, const int32_t *cframe, grid_args &ga,
                       const int *dlag, int nv, int64_t units, float4 *p) {
  int lo = dlag[0], hi = dlag[0];
  for (int l = 1; l < nv; l++) { lo = std::min(lo, dlag[l]); hi = std::max(hi, dlag[l]); }
  ga.nvalid = nv; ga.dlag_min = lo;
  ga.wlen = 256 + (hi - lo);
  ga.wstride = ga.wlen | 1;  // odd stride: windows of a wave start on different bank pairs
  for (int l = 0; l < 8; l++) ga.off[l] = dlag[std::min(l, nv - 1)] - lo;
  const int wpw = grid_waves_per_wg(ga);
  if (wpw == 0) return false;
  prof_scope ps(c, UWSPR_K_TONECORR, units, true);
  const float2 *fr = (const float2 *)frames;
  float *po = (float *)p;
  const int NL = nv <= 1 ? 1 : nv <= 2 ? 2 : nv <= 4 ? 4 : nv <= 5 ? 5 : nv <= 6 ? 6 : 8;
  switch (NL) {
    case 1: launch_grid_t<1>(c, ps, fr, nframes, ncentres, centres, cframe, ga, wpw, po); break;
    case 2: launch_grid_t<2>(c, ps, fr, nframes, ncentres, centres, cframe, ga, wpw, po); break;
    case 4: launch_grid_t<4>(c, ps, fr, nframes, ncentres, centres, cframe, ga, wpw, po); break;
    case 5: launch_grid_t<5>(c, ps, fr, nframes, ncentres, centres, cframe, ga, wpw, po); break;
    case 6: launch_grid_t<6>(c, ps, fr, nframes, ncentres, centres, cframe, ga, wpw, po); break;
    default: launch_grid_t<8>(c, ps, fr, nframes, ncentres, centres, cframe, ga, wpw, po); break;
  }
  return true;
}

static void grid_args_init(grid_args &ga, int nf, const float *df, int ndrift, const float *ddrift, int nlag) {
  memset(&ga, 0, sizeof(ga));
  ga.nf = nf; ga.ndrift = ndrift; ga.ncombo = nf * ndrift; ga.nlag_total = nlag;
  for (int i = 0; i < nf; i++) ga.df[i] = df[i];
  for (int i = 0; i < ndrift; i++) ga.ddrift[i] = ddrift[i];
  ga.wmax = std::min(16, (15 + ga.ncombo - 1) / ga.ncombo + 1);
}

// returns false when the grid does not fit the on-chip window scheme (caller falls back)
bool launch_tonecorr_grid(uwspr_ctx *c, const float *frames, int B, const uwspr_candidate *centres,
                          int nf, const float *df, int ndrift, const float *ddrift, int nlag,
                          const int *dlag_host, const int *dlag_dev, dev_hyp *hyps, float4 *p) {
  if (nf < 1 || nf > 32 || ndrift < 1 || ndrift > 32 || nlag < 1) return false;
  grid_args ga;
  grid_args_init(ga, nf, df, ndrift, ddrift, nlag);
  {
    prof_scope ps(c, UWSPR_K_SCHED, (int64_t)B * ga.ncombo * nlag);
    const long long H = (long long)B * ga.ncombo * nlag;
    hipLaunchKernelGGL(k_grid_hyps, dim3((unsigned)((H + 255) / 256)), dim3(256), 0, c->stream, centres, ga,
                       nlag, dlag_dev, (float)c->p.cf, hyps, B);
  }
  // lag blocks of up to 8; each block has its own window span
  for (int base = 0; base < nlag; base += 8) {
    const int nv = std::min(8, nlag - base);
    ga.lag_base = base;
    if (!launch_grid_block(c, frames, B, B, centres, nullptr, ga, dlag_host + base, nv,
                           (int64_t)B * ga.ncombo * nv, p))
      return false;
  }
  return true;
}

}  // namespace uwspr

