// K0 -- 12 kS/s real audio -> 375 S/s complex baseband.
//
// In the reference this stage is not gr-uwspr code at all: the flowgraph
// (examples/WaveFilePlusNoiseDecode.grc:840-916,1767-1768) chains GNU Radio's
// float_to_complex, two freq_xlating_fft_filter_ccc (band-pass 1500 +- 10 Hz, then
// translate by 1500 Hz + low-pass) and rational_resampler(decim 32), with taps from
// firdes / the resampler's own designer -- third-party, version-dependent, unpinned.
// This kernel is OUR single-stage equivalent, specified here and checked against a
// float64 numpy restatement of the same formula (tests/test_frontend.py):
//
//     y[m] = sum_{k=0}^{NT-1} h[k] * x[32 m + D - k] * exp(-j*pi*(32 m + D - k)/4),   D = (NT-1)/2
//
// h = Hamming-windowed sinc low-pass (cutoff 100 Hz at 12 kS/s, NT = 1025 taps, unit
// DC gain); x = 0 outside the record.  The mixer has period 8 and 8 | 32, so its
// phase depends only on (D - k) mod 8 and folds into complex taps g[k]:
//     y[m] = sum_k g[k] * x[32 m + D - k]   -> two real FIRs sharing the sample reads.
//
// Mapping: a 256-thread workgroup produces 256 consecutive outputs; the 9216 input
// samples it needs are staged in LDS in POLYPHASE order [n mod 32][n / 32], so that
// at every tap the 64 lanes of a wave read 64 consecutive words (conflict free);
// taps come from LDS as broadcast float2 reads.  46 M MAC per 2-minute frame:
// negligible beside K3/K4; HBM-bound (5.8 MB in, 0.36 MB out per frame).
#include <math.h>

#include <vector>

#include "uwspr_internal.h"

namespace uwspr {

constexpr int K0_NT = 1025;            // taps
constexpr int K0_D = (K0_NT - 1) / 2;  // group delay (samples at 12 kS/s)
constexpr int K0_DEC = 32;
constexpr int K0_OUT = 256;            // outputs per workgroup
constexpr int K0_SPAN = K0_OUT * K0_DEC + K0_NT - 1;     // input samples a workgroup touches (9216)
constexpr int K0_COLS = (K0_SPAN + K0_DEC - 1) / K0_DEC; // 288 polyphase columns

__global__ __launch_bounds__(K0_OUT) void k0_frontend(const float *__restrict__ audio, int nin,
                                                      const float2 *__restrict__ taps,
                                                      float2 *__restrict__ out, int nout) {
  __shared__ float xs[K0_DEC][K0_COLS + 1];
  __shared__ float2 gs[K0_NT + 7];
  const int b = blockIdx.y, tid = threadIdx.x;
  const int m0 = blockIdx.x * K0_OUT;
  const float *x = audio + (size_t)b * nin;
  // first input sample this workgroup needs: n0 = 32*m0 + D - (NT-1)
  const int n0 = K0_DEC * m0 + K0_D - (K0_NT - 1);
  for (int e = tid; e < K0_SPAN; e += K0_OUT) {
    const int n = n0 + e;
    const float v = (n >= 0 && n < nin) ? x[n] : 0.0f;
    xs[e & (K0_DEC - 1)][e >> 5] = v;  // n0 is a multiple of 32 (D = NT-1 - D, 32 | NT-1)
  }
  for (int k = tid; k < K0_NT; k += K0_OUT) gs[k] = taps[k];
  __syncthreads();
  const int m = m0 + tid;
  float re = 0.0f, im = 0.0f;
  // tap k multiplies x[32 m + D - k] = local sample e = 32*tid + (NT-1) - k
#pragma unroll 8
  for (int k = 0; k < K0_NT; k++) {
    const int e = K0_DEC * tid + (K0_NT - 1) - k;
    const float v = xs[e & (K0_DEC - 1)][e >> 5];
    const float2 g = gs[k];
    re = fmaf(g.x, v, re);
    im = fmaf(g.y, v, im);
  }
  if (m < nout) out[(size_t)b * nout + m] = make_float2(re, im);
}

// h[k] * exp(-j*pi*(D-k)/4): Hamming-windowed sinc, cutoff 100 Hz, unit DC gain
void frontend_taps(std::vector<float> &g) {
  std::vector<double> h(K0_NT);
  const double fc = 100.0 / 12000.0;
  double sum = 0.0;
  for (int k = 0; k < K0_NT; k++) {
    const double t = (double)(k - K0_D);
    const double sinc = t == 0.0 ? 2.0 * fc : sin(2.0 * M_PI * fc * t) / (M_PI * t);
    const double w = 0.54 - 0.46 * cos(2.0 * M_PI * (double)k / (double)(K0_NT - 1));
    h[k] = sinc * w;
    sum += h[k];
  }
  // exp(-j*pi*q/4) for q mod 8, exact octant values
  const double r = sqrt(0.5);
  const double cs[8] = {1, r, 0, -r, -1, -r, 0, r}, sn[8] = {0, r, 1, r, 0, -r, -1, -r};
  g.resize(2 * K0_NT);
  for (int k = 0; k < K0_NT; k++) {
    const int q = (((K0_D - k) % 8) + 8) % 8;
    g[2 * k] = (float)(h[k] / sum * cs[q]);
    g[2 * k + 1] = (float)(-h[k] / sum * sn[q]);
  }
}

void launch_frontend(uwspr_ctx *c, const float *audio, int B, int nin, float2 *out, int nout) {
  prof_scope ps(c, UWSPR_K_SPECTROGRAM, B);
  dim3 grid((nout + K0_OUT - 1) / K0_OUT, B);
  hipLaunchKernelGGL(k0_frontend, grid, dim3(K0_OUT), 0, c->stream, audio, nin,
                     (const float2 *)c->d_fe_taps, out, nout);
}

}  // namespace uwspr
