// K0 -- 12 kS/s real audio -> 375 S/s complex baseband.
//
// In the reference this stage is not gr-uwspr code: the flowgraph (examples/WaveFilePlusNoiseDecode.grc) chains GNU
// Radio's own blocks --
//     float_to_complex                                                             grc:527
//  -> freq_xlating_fft_filter_ccc(decim 1, band_pass(1, 12000, 1490, 1510, 10, HAMMING) real taps, centre 0)     grc:303-352, 840-893
//  -> freq_xlating_fft_filter_ccc(decim 1, low_pass(1, 12000, 1510, 10, HAMMING), centre 1500 Hz)                grc:358-400, 903-956
//  -> rational_resampler_ccc(interp 1, decim 32, no taps, fbw 0 -> its own Kaiser design)                       grc:1767-1808
// Mode 0 ("grc", the default) is THAT chain: the three tap sets are designed here on the host in binary64 from GNU
// Radio 3.7's published window-method formulas (gr-filter/lib/firdes.cc, gr-fft/lib/window.cc,
// gr-filter/python/filter/rational_resampler.py, freq_xlating_fft_filter.py; rounded to binary32 where the reference
// stores them so), and because every stage is linear and the 1500-Hz mixer has period 8 | 32 the whole chain is, at
// the decimated instants, ONE complex FIR on the real input:
//
//     y[m] = sum_k g[k] x[32 m + D - k],     g = h1 * (h2[k] e^{+j k pi/4}) * (h3[k] e^{+j k pi/4}),  6831 taps,  D = 0
//
// so the kernel works at the OUTPUT rate: 213 multiply-adds per input sample instead of the chain's 2 x 2891 complex
// ones at 12 kS/s.  Mode 1 ("compact") is the single-stage filter of rounds 1-4 (mix by -1500 Hz, 1025-tap Hamming
// low-pass at 100 Hz, D = 512): same kernel, other taps.  GNU Radio is absent from the reference tree and from this
// image: **parity unpinned by construction**; the independent float64 restatement is oracle/frontend_grc.py.
//
// Mapping.  Taps are split by phase p = k mod 32 (J = ceil(NT / 32) taps per phase, padded to a multiple of 8), the
// input is staged in LDS in polyphase order (row = n mod 32).  A 1024-thread workgroup makes 512 consecutive outputs:
// wavefront w owns phases 2w and 2w + 1, a lane owns EIGHT consecutive outputs, so one staged sample read feeds 16
// fused multiply-adds (the sliding window of a phase's column lives in registers: 8 new samples per 8 taps), the
// complex taps come as wave-uniform LDS reads, and the 16 partial sums per output are added in wavefront order at the
// end.  Within a row the columns are stored [column mod 8][column div 8] so that the 64 lanes of a read hit 64
// consecutive words.  LDS: 32 x (8 L + 1) samples + 32 J taps = 149 KB for the grc mode: one workgroup, 4 wavefronts
// per SIMD.  1.2 GFLOP and 5.8 MB in per 2-minute frame (213 flop / B): FP32-FMA bound.
#include <math.h>

#include <complex>
#include <vector>

#include "uwspr_internal.h"

namespace uwspr {

constexpr int K0_DEC = 32;
constexpr int K0_WG = 1024;             // 16 wavefronts x 2 tap phases
constexpr int K0_R = 8;                 // outputs per lane
constexpr int K0_OUT = 64 * K0_R;       // outputs per workgroup

__host__ __device__ constexpr int k0_L(int J) { return 64 + J / 8; }          // column groups per row segment
__host__ __device__ constexpr int k0_row(int J) { return 8 * k0_L(J) + 1; }   // floats per polyphase row (+1: the loader's lanes hit 32 banks)
static size_t k0_lds_bytes(int J) { return (size_t)K0_DEC * k0_row(J) * sizeof(float) + (size_t)K0_DEC * J * sizeof(float2); }

// 8 taps of one phase against the 15 samples they touch: acc[i] += g[u] * S[i + u], S = A[0..7] then B[0..7]
__device__ __forceinline__ void k0_block(float (&ar)[K0_R], float (&ai)[K0_R], const float (&A)[8], const float (&B)[8],
                                         const float2 *__restrict__ g) {
#pragma unroll
  for (int u = 0; u < 8; u++) {
    const float2 t = g[u];
#pragma unroll
    for (int i = 0; i < K0_R; i++) {
      const float s = (i + u < 8) ? A[i + u] : B[i + u - 8];
      ar[i] = fmaf(t.x, s, ar[i]);
      ai[i] = fmaf(t.y, s, ai[i]);
    }
  }
}

// taps: [32][J] float2, taps[p][jj] = g[32 (J - 1 - jj) + p] (zero beyond the filter); dcols = D / 32
__global__ __launch_bounds__(K0_WG) void k0_frontend(const float *__restrict__ audio, int nin,
                                                     const float2 *__restrict__ taps, float2 *__restrict__ out,
                                                     int nout, int J, int dcols) {
  extern __shared__ __align__(16) float k0_lds[];
  const int L = k0_L(J), ROW = k0_row(J);
  float *xs = k0_lds;
  float2 *tp = reinterpret_cast<float2 *>(k0_lds + K0_DEC * ROW);   // (32 ROW floats: a multiple of 8 bytes)
  const int b = blockIdx.y, tid = threadIdx.x;
  const int m0 = blockIdx.x * K0_OUT;
  const float *x = audio + (size_t)b * nin;
  // Output m, tap k = 32 j + p reads n = 32 (m - j + dcols) - p: row (32 - p) mod 32, column m - j + dcols - (p > 0).
  // With cb = m0 - J + dcols the workgroup's rows start at column cb; row 0 is stored one column late, so that for
  // every phase the sample of (output m0 + o, tap jj = J - 1 - j) sits at position o + jj of its row.
  const int cb = m0 - J + dcols;
  const int span = K0_DEC * (K0_OUT + J);
  for (int e = tid; e < span; e += K0_WG) {
    const int r = e & (K0_DEC - 1);
    const int pos = (e >> 5) - (r == 0 ? 1 : 0);
    if (pos < 0) continue;
    const long n = (long)K0_DEC * cb + e;
    const float v = (n >= 0 && n < nin) ? x[n] : 0.0f;
    xs[r * ROW + (pos & 7) * L + (pos >> 3)] = v;
  }
  for (int i = tid; i < K0_DEC * J; i += K0_WG) tp[i] = taps[i];
  __syncthreads();

  const int w = tid >> 6, l = tid & 63;
  float ar[K0_R], ai[K0_R];
#pragma unroll
  for (int i = 0; i < K0_R; i++) { ar[i] = 0.0f; ai[i] = 0.0f; }
  const int nblk = J / 8;
  for (int h = 0; h < 2; h++) {
    const int p = 2 * w + h;
    const float *xr = xs + ((K0_DEC - p) & (K0_DEC - 1)) * ROW + l;
    const float2 *g = tp + p * J;
    float A[8], B[8];
#pragma unroll
    for (int v = 0; v < 8; v++) A[v] = xr[v * L];
    int t = 0;
    for (; t + 2 <= nblk; t += 2) {           // two blocks per trip: the window's halves swap roles, no copies
#pragma unroll
      for (int v = 0; v < 8; v++) B[v] = xr[v * L + t + 1];
      k0_block(ar, ai, A, B, g + 8 * t);
#pragma unroll
      for (int v = 0; v < 8; v++) A[v] = xr[v * L + t + 2];
      k0_block(ar, ai, B, A, g + 8 * t + 8);
    }
    if (t < nblk) {
#pragma unroll
      for (int v = 0; v < 8; v++) B[v] = xr[v * L + t + 1];
      k0_block(ar, ai, A, B, g + 8 * t);
    }
  }
  __syncthreads();                             // every wavefront is done with the samples: their LDS becomes the partial sums
  float2 *part = reinterpret_cast<float2 *>(xs);      // [16 wavefronts][8 i][64 lanes]
#pragma unroll
  for (int i = 0; i < K0_R; i++) part[(w * K0_R + i) * 64 + l] = make_float2(ar[i], ai[i]);
  __syncthreads();
  if (tid < K0_OUT) {
    const int i = tid >> 6;                    // thread (i, l) finishes output 8 l + i
    float re = 0.0f, im = 0.0f;
#pragma unroll
    for (int ww = 0; ww < K0_WG / 64; ww++) {
      const float2 q = part[(ww * K0_R + i) * 64 + l];
      re += q.x; im += q.y;
    }
    const int m = m0 + K0_R * l + i;
    if (m < nout) out[(size_t)b * nout + m] = make_float2(re, im);
  }
}

// ---------------------------------------------------------------- tap designs (host, binary64)
// GNU Radio 3.7 gr-fft/lib/window.cc
static double gr_izero(double x) {             // Izero(): I0 by its power series, terms down to 1e-21 of the sum
  double sum = 1.0, u = 1.0;
  const double halfx = x / 2.0;
  int n = 1;
  do {
    double temp = halfx / (double)n;
    n += 1;
    temp *= temp;
    u *= temp;
    sum += u;
  } while (u >= 1e-21 * sum);
  return sum;
}
enum { WIN_HAMMING = 0, WIN_KAISER = 1 };
static std::vector<float> gr_window(int type, int ntaps, double beta) {
  std::vector<float> w(ntaps);
  if (type == WIN_HAMMING) {                   // window::hamming: `float M = ntaps - 1`
    const float M = (float)(ntaps - 1);
    for (int n = 0; n < ntaps; n++) w[n] = (float)(0.54 - 0.46 * cos((2 * M_PI * n) / M));
  } else {                                     // window::kaiser
    const double ibeta = 1.0 / gr_izero(beta), inm1 = 1.0 / (double)(ntaps - 1);
    for (int i = 0; i < ntaps; i++) {
      const double temp = 2 * i * inm1 - 1;
      w[i] = (float)(gr_izero(beta * sqrt(1.0 - temp * temp)) * ibeta);
    }
  }
  return w;
}
// gr-filter/lib/firdes.cc: compute_ntaps with window::max_attenuation (Hamming 53 dB, Kaiser beta / 0.1102 + 8.7)
static int gr_compute_ntaps(double fs, double tw, int type, double beta) {
  const double a = type == WIN_HAMMING ? 53.0 : beta / 0.1102 + 8.7;
  int ntaps = (int)(a * fs / (22.0 * tw));
  if ((ntaps & 1) == 0) ntaps++;
  return ntaps;
}
// firdes::low_pass (lo < 0) / firdes::band_pass: truncated ideal response x window, unit gain at 0 Hz / the band centre
static std::vector<float> gr_firdes(double gain, double fs, double lo, double hi, double tw, int type, double beta) {
  const int ntaps = gr_compute_ntaps(fs, tw, type, beta);
  std::vector<float> taps(ntaps), w = gr_window(type, ntaps, beta);
  const int M = (ntaps - 1) / 2;
  const bool band = lo >= 0.0;
  const double fwT0 = band ? 2 * M_PI * lo / fs : 0.0, fwT1 = 2 * M_PI * hi / fs;
  for (int n = -M; n <= M; n++) {
    if (n == 0) taps[n + M] = (float)((fwT1 - fwT0) / M_PI * w[n + M]);
    else if (band) taps[n + M] = (float)((sin(n * fwT1) - sin(n * fwT0)) / (n * M_PI) * w[n + M]);
    else taps[n + M] = (float)(sin(n * fwT1) / (n * M_PI) * w[n + M]);
  }
  double fmax = taps[0 + M];
  for (int n = 1; n <= M; n++) fmax += band ? 2 * taps[n + M] * cos(n * (fwT0 + fwT1) * 0.5) : 2 * taps[n + M];
  gain /= fmax;
  for (int i = 0; i < ntaps; i++) taps[i] = (float)(taps[i] * gain);
  return taps;
}
// gr-filter/python/filter/rational_resampler.py: design_filter(interp, decim, fractional_bw = 0.4 when none is given)
static std::vector<float> gr_resampler_taps(int interp, int decim) {
  const double fractional_bw = 0.4, beta = 7.0, halfband = 0.5;
  const double rate = (double)interp / (double)decim;
  double tw, mid;
  if (rate >= 1.0) { tw = halfband - fractional_bw; mid = halfband - tw / 2.0; }
  else { tw = rate * (halfband - fractional_bw); mid = rate * halfband - tw / 2.0; }
  return gr_firdes(interp, interp, -1.0, mid, tw, WIN_KAISER, beta);
}

typedef std::complex<double> cplx;
static std::vector<cplx> conv(const std::vector<cplx> &a, const std::vector<cplx> &b) {
  std::vector<cplx> c(a.size() + b.size() - 1, cplx(0, 0));
  for (size_t i = 0; i < a.size(); i++)
    for (size_t j = 0; j < b.size(); j++) c[i + j] += a[i] * b[j];
  return c;
}
// e^{+j k pi/4}: exact octant values (the mixer's period is 8 samples)
static cplx octant(long k) {
  static const double r = sqrt(0.5);
  static const double cs[8] = {1, r, 0, -r, -1, -r, 0, r}, sn[8] = {0, r, 1, r, 0, -r, -1, -r};
  const int q = (int)(((k % 8) + 8) % 8);
  return cplx(cs[q], sn[q]);
}

// stage 0: the composite complex taps g (pairs) and the read-ahead D; stages 1..3 (grc mode): the three real designs
int frontend_design(int mode, int stage, std::vector<double> &outv, int *delay) {
  outv.clear();
  if (delay) *delay = 0;
  if (mode == UWSPR_FRONTEND_COMPACT) {
    if (stage != 0) return UWSPR_ERR_ARG;
    // h[k] e^{-j pi (D - k) / 4}: Hamming-windowed sinc, cutoff 100 Hz, 1025 taps, unit DC gain, D = 512
    const int NT = 1025, D = 512;
    std::vector<double> h(NT);
    const double fc = 100.0 / 12000.0;
    double sum = 0.0;
    for (int k = 0; k < NT; k++) {
      const double t = (double)(k - D);
      const double sinc = t == 0.0 ? 2.0 * fc : sin(2.0 * M_PI * fc * t) / (M_PI * t);
      h[k] = sinc * (0.54 - 0.46 * cos(2.0 * M_PI * (double)k / (double)(NT - 1)));
      sum += h[k];
    }
    for (int k = 0; k < NT; k++) {
      const cplx g = (h[k] / sum) * std::conj(octant(D - k));
      outv.push_back(g.real()); outv.push_back(g.imag());
    }
    if (delay) *delay = D;
    return UWSPR_OK;
  }
  if (mode != UWSPR_FRONTEND_GRC) return UWSPR_ERR_ARG;
  const std::vector<float> h1 = gr_firdes(1.0, 12000.0, 1500.0 - 10.0, 1500.0 + 10.0, 10.0, WIN_HAMMING, 6.76);
  const std::vector<float> h2 = gr_firdes(1.0, 12000.0, -1.0, 1500.0 + 10.0, 10.0, WIN_HAMMING, 6.76);
  const std::vector<float> h3 = gr_resampler_taps(1, 32);
  if (stage >= 1 && stage <= 3) {
    const std::vector<float> &h = stage == 1 ? h1 : stage == 2 ? h2 : h3;
    for (float v : h) outv.push_back((double)v);
    return UWSPR_OK;
  }
  if (stage != 0) return UWSPR_ERR_ARG;
  std::vector<cplx> a(h1.size()), bq(h2.size()), cq(h3.size());
  for (size_t k = 0; k < h1.size(); k++) a[k] = cplx(h1[k], 0.0);
  // freq_xlating_fft_filter_ccc hands fft_filter_ccc gr_complex taps: the rotated taps are binary32 pairs
  for (size_t k = 0; k < h2.size(); k++) {
    const cplx z = (double)h2[k] * octant((long)k);
    bq[k] = cplx((double)(float)z.real(), (double)(float)z.imag());
  }
  // the resampler filters BEHIND the mixer: its taps meet the input rotated, in exact arithmetic
  for (size_t k = 0; k < h3.size(); k++) cq[k] = (double)h3[k] * octant((long)k);
  const std::vector<cplx> g = conv(conv(a, bq), cq);
  for (const cplx &z : g) { outv.push_back(z.real()); outv.push_back(z.imag()); }
  return UWSPR_OK;
}

// the kernel's tap image: [32][J] float2, jj ascending = tap index descending
int frontend_tap_image(int mode, std::vector<float> &img, int *J_out, int *dcols_out) {
  std::vector<double> g;
  int D = 0;
  const int rc = frontend_design(mode, 0, g, &D);
  if (rc) return rc;
  const int NT = (int)g.size() / 2;
  int J = (NT + K0_DEC - 1) / K0_DEC;
  J = (J + 7) / 8 * 8;
  img.assign((size_t)2 * K0_DEC * J, 0.0f);
  for (int p = 0; p < K0_DEC; p++)
    for (int jj = 0; jj < J; jj++) {
      const int k = K0_DEC * (J - 1 - jj) + p;
      if (k < NT) { img[2 * ((size_t)p * J + jj)] = (float)g[2 * k]; img[2 * ((size_t)p * J + jj) + 1] = (float)g[2 * k + 1]; }
    }
  *J_out = J; *dcols_out = D / K0_DEC;
  return (D % K0_DEC) ? UWSPR_ERR_ARG : UWSPR_OK;
}

int frontend_prepare() {
  // up to 160 KB of dynamic LDS (149 KB in the grc mode)
  return hipFuncSetAttribute(reinterpret_cast<const void *>(k0_frontend), hipFuncAttributeMaxDynamicSharedMemorySize,
                             160 * 1024) == hipSuccess ? 0 : -1;
}

void launch_frontend(uwspr_ctx *c, const float *audio, int B, int nin, float2 *out, int nout) {
  prof_scope ps(c, UWSPR_K_SPECTROGRAM, B);
  dim3 grid((nout + K0_OUT - 1) / K0_OUT, B);
  hipLaunchKernelGGL(k0_frontend, grid, dim3(K0_WG), k0_lds_bytes(c->fe_J), c->stream, audio, nin,
                     (const float2 *)c->d_fe_taps, out, nout, c->fe_J, c->fe_dcols);
}

}  // namespace uwspr
