// K3 -- coarse (freq, start offset, drift) search and candidate selection.
//
// Reference: FDR_impl::transform hot loop 2, lib/FDR_impl.cc:339-409 with the
// powersum kernel cc:188-210 and the SLM trajectory generator lib/slm.cc:36-121.
// Per candidate: 5 tuned bins x 26 half-symbol offsets x ((2*maxdrift+1) linear
// drifts + 125 straight-line trajectories) = 16380 hypotheses at the defaults,
// each a 162-term sync-vector correlation over the spectrogram.
//
// Mapping (one 1024-thread workgroup per candidate):
//  * the spectrogram window the candidate can touch is staged once into LDS as
//    float4 {sqrt ps[row][c-3], [c-1], [c+1], [c+3]} per (row, centre column c):
//    one ds_read_b128 gather per symbol instead of four gathers + four sqrt;
//  * the bin offsets ifd-ifr of every (ifr, hypothesis, symbol) come from a
//    table built once per context with the reference's exact expressions
//    (cc:353 double / cc:382-385 float), 4 symbols per 32-bit word, read
//    coalesced across lanes;
//  * one lane = one hypothesis, 162 sequential steps, so ss and pow accumulate
//    in the reference's order (cc:207-209) and the metric is bit-identical;
//  * all 16380 metrics stay in LDS; wave 0 then replays the reference's
//    order-dependent running-best rule (strict > for linear cc:360, ratio
//    against the running best for nonlinear cc:392) with ballots: a wave scans
//    64 metrics per step and only serialises on acceptances.
// Roofline: LDS-gather / VALU bound; HBM traffic is the tile once (~60 KB).
#include "uwspr_internal.h"

#pragma clang fp contract(off)

namespace uwspr {

constexpr int K3_THREADS = 1024;

__global__ __launch_bounds__(K3_THREADS) void k3_coarse(
    const float *__restrict__ ps, fdr_consts f, const uint32_t *__restrict__ off_tab,
    uwspr_candidate *__restrict__ cands, const int32_t *__restrict__ npk,
    float *__restrict__ syncgrid, int grid_cap) {
  extern __shared__ __align__(16) unsigned char smem[];
  float4 *tile = reinterpret_cast<float4 *>(smem);                            // [n][nc]
  float *syncbuf = reinterpret_cast<float *>(smem + (size_t)f.n * f.nc * 16); // [ntot]

  const int b = blockIdx.y, j = blockIdx.x, tid = threadIdx.x;
  if (j >= npk[b]) return;  // workgroup-uniform
  uwspr_candidate *cand = cands + (size_t)b * f.maxfreqs + j;
  const float freq0 = cand->freq;
  // cc:341: if0 = freq/df + m (binary32), truncated
  const int if0 = (int)(ieee_divf(freq0, f.df) + (float)f.m);

  // ---- stage the sqrt tile ------------------------------------------------
  const float *psb = ps + (size_t)b * f.n * f.band_w;
  const int c0 = if0 - 2 + f.off_min - f.band_lo;  // band column of centre index 0
  for (int idx = tid; idx < f.n * f.nc; idx += K3_THREADS) {
    int row = idx / f.nc, ci = idx - row * f.nc;
    const float *pr = psb + (size_t)row * f.band_w + c0 + ci;
    tile[idx] = make_float4(ieee_sqrtf(pr[-3]), ieee_sqrtf(pr[-1]), ieee_sqrtf(pr[1]),
                            ieee_sqrtf(pr[3]));
  }
  __syncthreads();

  // ---- one lane per hypothesis -------------------------------------------
  const int hc = f.cell_hyps;
  const int nc2 = 2 * f.nc;
  for (int g = tid; g < f.ntot; g += K3_THREADS) {
    const int cell = g / hc, h = g - cell * hc;
    const int ifr_i = cell / UWSPR_NK0, k0 = cell - ifr_i * UWSPR_NK0;
    const int ifr_idx = if0 - 2 + ifr_i - f.ifr_lo;
    const uint32_t *ot = off_tab + (size_t)ifr_idx * 41 * hc + h;
    int idx = k0 * f.nc + ifr_i - f.off_min;  // tile index of (row k0, offset 0)
    float ss = 0.0f, pw = 0.0f;
#pragma unroll
    for (int k4 = 0; k4 < 41; k4++) {
      const uint32_t w = ot[(size_t)k4 * hc];
#pragma unroll
      for (int kk = 0; kk < 4; kk++) {
        const int k = 4 * k4 + kk;
        if (k < UWSPR_NSYM) {
          const int o = (int)(int8_t)(w >> (8 * kk));
          const float4 P = tile[idx + o];
          idx += nc2;  // kindex = k0 + 2k (cc:197)
          const float cm = (P.y + P.w) - (P.x + P.z);
          ss = pr3_bit(k) ? ss + cm : ss - cm;  // (2*pr3[k]-1)*cm, cc:207
          pw = pw + P.x; pw = pw + P.y; pw = pw + P.z; pw = pw + P.w;  // cc:209
        }
      }
    }
    syncbuf[g] = ieee_divf(ss, pw);  // cc:357,390
  }
  __syncthreads();

  if (syncgrid != nullptr && j < grid_cap) {
    float *gout = syncgrid + ((size_t)b * grid_cap + j) * f.ntot;
    for (int g = tid; g < f.ntot; g += K3_THREADS) gout[g] = syncbuf[g];
  }

  // ---- replay the running-best selection in reference order ---------------
  if (tid < 64) {
    float best = -1e30f;
    int gbest = -1;
    for (int base = 0; base < f.ntot; base += 64) {
      const int g = base + tid;
      const bool in = g < f.ntot;
      const float v = in ? syncbuf[g] : 0.0f;
      const bool lin = (g % hc) < f.nlin;
      int start = 0;
      for (;;) {
        const bool pred = in && tid >= start &&
                          (lin ? (v > best) : (ieee_divf(v, best) > f.threshold));
        const unsigned long long mask = __ballot(pred);
        if (mask == 0ull) break;
        const int first = __ffsll((long long)mask) - 1;
        best = __shfl(v, first);
        gbest = base + first;
        start = first + 1;
      }
    }
    if (tid == 0) {
      cand->sync = best;
      if (gbest >= 0) {
        const int cell = gbest / hc, h = gbest - cell * hc;
        const int ifr_i = cell / UWSPR_NK0, k0 = cell - ifr_i * UWSPR_NK0;
        cand->shift = 128 * k0;                                // cc:361,397
        cand->freq = (float)(if0 - 2 + ifr_i - f.m) * f.df;    // cc:362,398
        if (h < f.nlin) {
          cand->m_type = UWSPR_LINEAR;
          cand->m_nonlinear.V1 = 0.0; cand->m_nonlinear.V2 = 0.0;
          cand->m_nonlinear.p1 = 0; cand->m_nonlinear.p2 = 0;
          cand->m_linear.drift = (float)(h - f.maxdrift);      // cc:366
        } else {
          const int s = h - f.nlin;  // slm.cc:76-116: p2 fastest, then V1, then V2
          cand->m_type = UWSPR_NONLINEAR;
          cand->m_nonlinear.V1 = (double)((s / 5) % 5) - 2.0;
          cand->m_nonlinear.V2 = (double)(s / 25) - 2.0;
          cand->m_nonlinear.p1 = 0;
          cand->m_nonlinear.p2 = 50 + 200 * (s % 5);
        }
      }
    }
  }
}

size_t coarse_lds_bytes(const fdr_consts &f) {
  return (size_t)f.n * f.nc * 16 + (size_t)f.ntot * 4;
}

void launch_coarse(uwspr_ctx *c, int B) {
  const fdr_consts &f = c->fc;
  prof_scope ps(c, UWSPR_K_COARSE, (int64_t)B);
  hipLaunchKernelGGL(k3_coarse, dim3(f.cand_slots, B), dim3(K3_THREADS), coarse_lds_bytes(f),
                     c->stream, c->d_ps, f, c->d_off, c->d_cands, c->d_npk, c->d_syncgrid,
                     c->d_syncgrid ? c->grid_cap : 0);
}

int coarse_configure(const fdr_consts &f) {
  size_t need = coarse_lds_bytes(f);
  if (need > 160 * 1024) return -1;
  hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(k3_coarse),
                                     hipFuncAttributeMaxDynamicSharedMemorySize, (int)need);
  return e == hipSuccess ? 0 : -2;
}

}  // namespace uwspr
