// K2 -- average spectrum, smoothing, noise percentile, normalisation, peak
// pick and candidate sort: FDR_impl::transform, lib/FDR_impl.cc:257-319.
//
// Tiny (per frame: 348 x band_w adds, then <= 512-element vectors); one
// 256-thread workgroup per frame.  Every reduction keeps the reference's
// left-to-right binary32 order:
//   psavg[j]  = sum over rows i ascending            (cc:257-263)
//   smspec[i] = sum over j=-3..3 ascending           (cc:268-275)
// The qsort + percentile (cc:277-285) is a rank selection (any correct sort
// yields the same value), the peak scan (cc:293-306) is an ordered compaction,
// the bubble sort (cc:311-319) is a stable descending rank by snr.
#include "uwspr_internal.h"

#pragma clang fp contract(off)

namespace uwspr {

// FDR_impl.cc:303 takes `log10(smspec[j])` of a float: g++ resolves it to log10f, and log10f is NOT correctly rounded
// -- its last bit is whatever the host's libm computes.  This is glibc 2.35's (the image's; Ubuntu 22.04) algorithm,
// operation for operation: sysdeps/ieee754/flt-32/e_log10f.c (k = exponent, mantissa scaled into [sqrt(1/2), sqrt 2),
// y*log10_2lo + ivln10*logf(x) + y*log10_2hi in binary32) around sysdeps/ieee754/flt-32/e_logf.c (16-entry table
// of 1/c and log c, third-degree polynomial, all in binary64, rounded once) with its table logf_data.c.  On x86-64
// libm selects a build of logf with or without fused multiply-adds by the CPU; both give the same binary32 result for
// EVERY argument (tests/test_log10_gap.py walks all 2^31 - 2^23 positive ones against the host's libm with both), so
// the plain form below is the one restated.  With it the `snr` field is the reference's to the bit wherever the
// reference runs on that libm; a libm with another log10f (glibc >= 2.40 rounds correctly) differs in the last
// bit on a few per cent of the arguments, as the binary64 route of rounds 1-4 did.
__device__ const double kLogfTab[16][2] = {
    {0x1.661ec79f8f3bep+0, -0x1.57bf7808caadep-2}, {0x1.571ed4aaf883dp+0, -0x1.2bef0a7c06ddbp-2},
    {0x1.49539f0f010bp+0, -0x1.01eae7f513a67p-2},  {0x1.3c995b0b80385p+0, -0x1.b31d8a68224e9p-3},
    {0x1.30d190c8864a5p+0, -0x1.6574f0ac07758p-3}, {0x1.25e227b0b8eap+0, -0x1.1aa2bc79c81p-3},
    {0x1.1bb4a4a1a343fp+0, -0x1.a4e76ce8c0e5ep-4}, {0x1.12358f08ae5bap+0, -0x1.1973c5a611cccp-4},
    {0x1.0953f419900a7p+0, -0x1.252f438e10c1ep-5}, {0x1p+0, 0x0p+0},
    {0x1.e608cfd9a47acp-1, 0x1.aa5aa5df25984p-5},  {0x1.ca4b31f026aap-1, 0x1.c5e53aa362eb4p-4},
    {0x1.b2036576afce6p-1, 0x1.526e57720db08p-3},  {0x1.9c2d163a1aa2dp-1, 0x1.bc2860d22477p-3},
    {0x1.886e6037841edp-1, 0x1.1058bc8a07ee1p-2},  {0x1.767dcf5534862p-1, 0x1.4043057b6ee09p-2}};

__device__ __forceinline__ float logf_glibc235(float x) {
  uint32_t ix = __float_as_uint(x);
  if (ix == 0x3f800000u) return 0.0f;
  if (ix - 0x00800000u >= 0x7f800000u - 0x00800000u) {       // zero, subnormal, negative, inf, nan
    if (ix * 2 == 0) return -__builtin_inff();
    if (ix == 0x7f800000u) return x;
    if ((ix & 0x80000000u) || ix * 2 >= 0xff000000u) return __builtin_nanf("");
    ix = __float_as_uint(x * 0x1p23f);
    ix -= 23u << 23;
  }
  const uint32_t tmp = ix - 0x3f330000u;
  const int i = (int)((tmp >> 19) & 15u);
  const int k = (int32_t)tmp >> 23;
  const uint32_t iz = ix - (tmp & (0x1ffu << 23));
  const double invc = kLogfTab[i][0], logc = kLogfTab[i][1];
  const double z = (double)__uint_as_float(iz);
  const double r = z * invc - 1.0;
  const double y0 = logc + (double)k * 0x1.62e42fefa39efp-1;
  const double r2 = r * r;
  double y = 0x1.5575b0be00b6ap-2 * r + -0x1.ffffef20a4123p-2;
  y = -0x1.00ea348b88334p-2 * r2 + y;
  y = y * r2 + (y0 + r);
  return (float)y;
}

__device__ __forceinline__ float log10f_glibc235(float x) {
  const float two25 = 3.3554432000e+07f, ivln10 = 4.3429449201e-01f, log10_2hi = 3.0102920532e-01f,
              log10_2lo = 7.9034151668e-07f;
  int32_t hx = (int32_t)__float_as_uint(x), k = 0;
  if (hx < 0x00800000) {
    if ((hx & 0x7fffffff) == 0) return -two25 / fabsf(x);
    if (hx < 0) return (x - x) / (x - x);
    k -= 25;
    x *= two25;
    hx = (int32_t)__float_as_uint(x);
  }
  if (hx >= 0x7f800000) return x + x;
  k += (hx >> 23) - 127;
  const int32_t i = (int32_t)(((uint32_t)k & 0x80000000u) >> 31);
  hx = (hx & 0x007fffff) | ((0x7f - i) << 23);
  const float y = (float)(k + i);
  const float z = y * log10_2lo + ivln10 * logf_glibc235(__uint_as_float((uint32_t)hx));
  return z + y * log10_2hi;
}

// diagnostics (tests/test_log10_gap.py): the `snr` expression of cc:303 over an array
__global__ void k_debug_snr_db(const float *__restrict__ x, float *__restrict__ out, long long n) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) out[i] = 10.0f * log10f_glibc235(x[i]);
}

// 1024 threads: the frame's 60 KB tile comes into LDS with 16-byte loads, four per thread (with 256 threads and
// 4-byte loads the copy alone was most of the kernel's 15 us); the later steps are strided loops over <= 512 items
constexpr int K2_THREADS = 1024;
constexpr int K2_MAXV = 512;  // >= band_w and >= finpb

__global__ __launch_bounds__(K2_THREADS) void k2_spectrum(
    const float *__restrict__ ps, fdr_consts f, float *__restrict__ psavg_g,
    float *__restrict__ smraw_g, float *__restrict__ smspec_g, float *__restrict__ noise_g,
    uwspr_candidate *__restrict__ cands, int32_t *__restrict__ npk_g, int stage_lds,
    int32_t *__restrict__ work_count, int32_t *__restrict__ work_list) {
  extern __shared__ __align__(16) float ps_s[];  // [n][band_w] when stage_lds
  __shared__ float psavg[K2_MAXV];
  __shared__ float sm[K2_MAXV];
  __shared__ int flag[K2_MAXV];
  __shared__ float pk_freq[K2_MAXV / 2 + 1];
  __shared__ float pk_snr[K2_MAXV / 2 + 1];
  __shared__ float noise_s;
  __shared__ int npk_s;

  const int b = blockIdx.x, tid = threadIdx.x;
  const float *psb = ps + (size_t)b * f.n * f.band_w;
  if (tid == 0) noise_s = __builtin_nanf("");  // stays NaN only if the frame holds NaNs

  // psavg over the kept columns, rows ascending (cc:257-263).  The column sums
  // are serial in the row index, so the frame's tile is first brought into LDS
  // with all 256 threads (coalesced), then one thread per column adds it up.
  if (stage_lds) {
    const int tot = f.n * f.band_w;
    if ((tot & 3) == 0 && ((size_t)b * tot & 3) == 0) {   // 16-byte aligned frame tile (ps itself is)
      const float4 *src = reinterpret_cast<const float4 *>(psb);
      float4 *dst = reinterpret_cast<float4 *>(ps_s);
      for (int e = tid; e < (tot >> 2); e += K2_THREADS) dst[e] = src[e];
    } else {
      for (int e = tid; e < tot; e += K2_THREADS) ps_s[e] = psb[e];
    }
    __syncthreads();
    for (int col = tid; col < f.band_w; col += K2_THREADS) {
      float acc = 0.0f;
#pragma unroll 12
      for (int i = 0; i < f.n; i++) acc = acc + ps_s[i * f.band_w + col];
      psavg[col] = acc;
      psavg_g[(size_t)b * f.band_w + col] = acc;
    }
  } else {
    for (int col = tid; col < f.band_w; col += K2_THREADS) {
      float acc = 0.0f;
      int i = 0;
      for (; i + 12 <= f.n; i += 12) {
        float v[12];
#pragma unroll
        for (int q = 0; q < 12; q++) v[q] = psb[(size_t)(i + q) * f.band_w + col];
#pragma unroll
        for (int q = 0; q < 12; q++) acc = acc + v[q];
      }
      for (; i < f.n; i++) acc = acc + psb[(size_t)i * f.band_w + col];
      psavg[col] = acc;
      psavg_g[(size_t)b * f.band_w + col] = acc;
    }
  }
  __syncthreads();

  // 7-tap smoothing inside the pass band (cc:265-275)
  for (int i = tid; i < f.finpb; i += K2_THREADS) {
    float acc = 0.0f;
    for (int j = -3; j <= 3; j++) acc = acc + psavg[f.m - f.hpbm + i + j - f.band_lo];
    sm[i] = acc;
    smraw_g[(size_t)b * f.finpb + i] = acc;
  }
  __syncthreads();

  // 30th percentile by rank selection (cc:277-285)
  for (int i = tid; i < f.finpb; i += K2_THREADS) {
    float v = sm[i];
    int rank = 0;
    for (int j = 0; j < f.finpb; j++) {
      float u = sm[j];
      rank += (u < v) || (u == v && j < i);
    }
    if (rank == f.noiseidx) noise_s = v;
  }
  if (tid == 0) npk_s = 0;
  __syncthreads();
  const float noise = noise_s;
  if (tid == 0) noise_g[b] = noise;

  // SNR in linear form and floor (cc:287-291)
  for (int j = tid; j < f.finpb; j += K2_THREADS) {
    float v = (float)((double)ieee_divf(sm[j], noise) - 1.0);
    if (v < f.min_snr) v = f.min_snr_floor;
    sm[j] = v;   // each thread rewrites only its own slots
    smspec_g[(size_t)b * f.finpb + j] = v;
  }
  __syncthreads();

  // strict local maxima (cc:293-306)
  for (int j = tid; j < f.finpb; j += K2_THREADS)
    flag[j] = (j >= 1 && j < f.finpb - 1 && sm[j] > sm[j - 1] && sm[j] > sm[j + 1]) ? 1 : 0;
  __syncthreads();
  for (int j = tid; j < f.finpb; j += K2_THREADS) {
    if (flag[j]) {
      int pos = 0;
      for (int q = 0; q < j; q++) pos += flag[q];
      if (pos < f.maxfreqs) {
        pk_freq[pos] = (float)(j - f.hpbm) * f.df;
        // cc:303: 10*log10(smspec) -- log10f, this libm's (see log10f_glibc235)
        pk_snr[pos] = 10.0f * log10f_glibc235(sm[j]);
        atomicAdd(&npk_s, 1);
      }
    }
  }
  __syncthreads();
  const int npk = npk_s;
  if (tid == 0) npk_g[b] = npk;

  // stable descending order by snr (bubble sort cc:309-319)
  uwspr_candidate *out = cands + (size_t)b * f.maxfreqs;
  for (int k = tid; k < npk; k += K2_THREADS) {
    float s = pk_snr[k];
    int rank = 0;
    for (int q = 0; q < npk; q++) {
      float u = pk_snr[q];
      rank += (u > s) || (u == s && q < k);
    }
    uwspr_candidate c;
    c.freq = pk_freq[k]; c.snr = s; c.drift = 0.0f; c.sync = 0.0f; c.shift = 0;
    c.m_type = UWSPR_LINEAR;
    c.m_nonlinear.V1 = 0.0; c.m_nonlinear.V2 = 0.0; c.m_nonlinear.p1 = 0; c.m_nonlinear.p2 = 0;
    out[rank] = c;
  }
  // work list for the coarse search: one item per (frame, candidate)
  if (tid == 0 && npk > 0) {
    const int base = atomicAdd(work_count, npk);
    for (int k = 0; k < npk; k++) work_list[base + k] = b * f.cand_slots + k;
  }
}

void launch_spectrum(uwspr_ctx *c, int B) {
  prof_scope ps(c, UWSPR_K_SPECTRUM, B);
  const size_t tile = (size_t)c->fc.n * c->fc.band_w * sizeof(float);
  const int stage = tile <= 60 * 1024;
  hipLaunchKernelGGL(k2_spectrum, dim3(B), dim3(K2_THREADS), stage ? tile : 0, c->stream, c->d_ps,
                     c->fc, c->d_psavg, c->d_smraw, c->d_smspec, c->d_noise, c->cur_cands, c->cur_npk,
                     stage, c->d_work, c->d_work + 1);
}

}  // namespace uwspr

// diagnostics (not part of the ABI header): out[i] = 10 * log10f(x[i]) as K2 computes a candidate's snr; device pointers
extern "C" int uwspr_debug_snr_db(uwspr_ctx *c, const float *x_dev, float *out_dev, long long n) {
  if (!c || !x_dev || !out_dev || n < 0) return UWSPR_ERR_ARG;
  if (n == 0) return UWSPR_OK;
  hipLaunchKernelGGL(uwspr::k_debug_snr_db, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, c->stream, x_dev, out_dev, n);
  return hipGetLastError() == hipSuccess ? UWSPR_OK : UWSPR_ERR_HIP;
}
