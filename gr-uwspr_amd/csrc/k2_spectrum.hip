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
        // cc:303: 10*log10(smspec) as a binary32 value
        pk_snr[pos] = 10.0f * (float)log10((double)sm[j]);
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
