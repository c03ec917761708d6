// K3 -- coarse (freq, start offset, drift) search and candidate selection.
//
// Reference: FDR_impl::transform hot loop 2, lib/FDR_impl.cc:339-409 with the
// powersum kernel cc:188-210 and the SLM trajectory generator lib/slm.cc:36-121.
// Per candidate: 5 tuned bins x 26 half-symbol offsets x ((2*maxdrift+1) linear
// drifts + 125 straight-line trajectories) = 16380 hypotheses at the defaults,
// each a 162-term sync-vector correlation over the spectrogram.
//
// What makes it cheap without changing a single result:
//  * A hypothesis is fully described by its sequence of bin offsets ifd-ifr over
//    the 162 symbols.  Those sequences are built once per context with the
//    reference's exact expressions (cc:353 binary64 / cc:382-385 binary32) and
//    DEDUPLICATED: at the defaults only 38 of the 126 per-cell sequences are
//    distinct (the SLM reach is a few bins, so many trajectories quantise to the
//    same path).  Identical sequences give bit-identical metrics, so each distinct
//    one is evaluated once (130 x 38 = 4940 evaluations per candidate, not 16380)
//    and the selection below reads it back through a hypothesis -> sequence map.
//  * The spectrogram window the candidate can touch is staged once into LDS as
//    float4 {sqrt ps[row][c-3], [c-1], [c+1], [c+3]} per (row, centre column c):
//    one ds_read_b128 gather per symbol instead of four gathers + four sqrt.
//  * One lane = one (cell, sequence), 162 sequential steps, so ss and pow
//    accumulate in the reference's order (cc:207-209): bit-identical metrics.
//  * Wave 0 then replays the reference's ORDER-DEPENDENT running-best rule over
//    all 16380 hypotheses in reference order (strict > for linear cc:360, ratio
//    against the running best for nonlinear cc:392) with ballots: 64 hypotheses
//    per step, serialising only on acceptances.
// One 1024-thread workgroup per candidate; everything between the spectrogram
// tile read and the 48-byte candidate record stays in LDS.
// Roofline: LDS-gather / VALU bound; HBM traffic is the tile once (~60 KB).
#include <algorithm>

#include "uwspr_internal.h"

#pragma clang fp contract(off)

namespace uwspr {

constexpr int K3_THREADS = 1024;

struct k3_lds_layout { size_t tile, uoff, sync, umap, total; };
__host__ __device__ inline k3_lds_layout k3_layout(const fdr_consts &f) {
  k3_lds_layout l;
  l.tile = 0;
  l.uoff = l.tile + (size_t)f.n * f.nc * 16;
  l.sync = l.uoff + (f.uoff_global ? 0 : (size_t)UWSPR_NIFR * f.umax * 41 * 4);
  l.umap = l.sync + (size_t)UWSPR_NIFR * UWSPR_NK0 * f.umax * 4;
  l.total = l.umap + (((size_t)UWSPR_NIFR * f.cell_hyps * 2 + 15) & ~(size_t)15);
  // after the evaluation the tile + offset-table range is reused for the expanded
  // metrics [ntot rounded to 64] + 3 floats per 64-value slice: grow it if needed
  const size_t reuse = ((size_t)((f.ntot + 63) & ~63) + 3 * (size_t)((f.ntot + 63) >> 6)) * 4;
  if (reuse > l.sync) {
    const size_t extra = (reuse - l.sync + 15) & ~(size_t)15;
    l.sync += extra; l.umap += extra; l.total += extra;
  }
  return l;
}

__global__ __launch_bounds__(K3_THREADS) void k3_coarse(
    const float *__restrict__ ps, fdr_consts f, const uint32_t *__restrict__ uoff_tab,
    const uint16_t *__restrict__ umap_tab, uwspr_candidate *__restrict__ cands,
    const int32_t *__restrict__ work, float *__restrict__ syncgrid, int grid_cap) {
  extern __shared__ __align__(16) unsigned char smem[];
  const k3_lds_layout lay = k3_layout(f);
  float4 *tile = reinterpret_cast<float4 *>(smem + lay.tile);        // [n][nc]
  uint32_t *uoff = reinterpret_cast<uint32_t *>(smem + lay.uoff);    // [5][umax][41]
  float *syncbuf = reinterpret_cast<float *>(smem + lay.sync);       // [130][umax]
  uint16_t *umap = reinterpret_cast<uint16_t *>(smem + lay.umap);    // [5][cell_hyps]

  const int tid = threadIdx.x;
  const int nwork = work[0];
#ifdef K3_STAMPS
  long long t_a = clock64(), t_b = 0, t_c = 0, t_d = 0;
#endif
  // persistent workgroups: K2 compacted the (frame, candidate) pairs into a work
  // list, so no workgroup is launched only to find it has no candidate
  for (int wi = blockIdx.x; wi < nwork; wi += gridDim.x) {
  const int item = work[1 + wi];
  const int b = item / f.cand_slots, j = item - b * f.cand_slots;
  uwspr_candidate *cand = cands + (size_t)b * f.maxfreqs + j;
  const float freq0 = cand->freq;
  // cc:341: if0 = freq/df + m (binary32), truncated
  const int if0 = (int)(ieee_divf(freq0, f.df) + (float)f.m);
  const int r0 = if0 - 2 - f.ifr_lo;  // first of the 5 rows of the offset tables

  // ---- stage the sqrt tile and this candidate's offset rows ----------------
  const float *psb = ps + (size_t)b * f.n * f.band_w;
  const int c0 = if0 - 2 + f.off_min - f.band_lo;  // band column of centre index 0
  for (int idx = tid; idx < f.n * f.nc; idx += K3_THREADS) {
    int row = idx / f.nc, ci = idx - row * f.nc;
    const float *pr = psb + (size_t)row * f.band_w + c0 + ci;
    tile[idx] = make_float4(ieee_sqrtf(pr[-3]), ieee_sqrtf(pr[-1]), ieee_sqrtf(pr[1]),
                            ieee_sqrtf(pr[3]));
  }
  const int nuw = UWSPR_NIFR * f.umax * 41;
  // (large cf / maxdrift: the sequences stay in HBM/L2 and only the tile and the metrics use LDS)
  if (!f.uoff_global)
    for (int idx = tid; idx < nuw; idx += K3_THREADS) uoff[idx] = uoff_tab[(size_t)r0 * f.umax * 41 + idx];
  const uint32_t *uo_base = f.uoff_global ? uoff_tab + (size_t)r0 * f.umax * 41 : uoff;
  for (int idx = tid; idx < UWSPR_NIFR * f.cell_hyps; idx += K3_THREADS)
    umap[idx] = umap_tab[(size_t)r0 * f.cell_hyps + idx];
  __syncthreads();
#ifdef K3_STAMPS
  t_b = clock64();
#endif

  // ---- one lane per (cell, distinct offset sequence) -------------------------
  const int nc2 = 2 * f.nc;
  const int neval = UWSPR_NIFR * UWSPR_NK0 * f.umax;
  for (int g = tid; g < neval; g += K3_THREADS) {
    const int cell = g / f.umax, u = g - cell * f.umax;
    const int ifr_i = cell / UWSPR_NK0, k0 = cell - ifr_i * UWSPR_NK0;
    const uint32_t *ot = uo_base + (ifr_i * f.umax + u) * 41;
    int idx = k0 * f.nc + ifr_i - f.off_min;  // tile index of (row k0, offset 0)
    float ss = 0.0f, pw = 0.0f;
    // Software pipeline: the 4 gathers of symbol group k4+1 (and the offset word
    // of group k4+2) are in flight while group k4 is accumulated, so the LDS
    // latency is paid once per 4 symbols instead of once per symbol.
    float4 Pc[4], Pn[4];
    uint32_t w1 = ot[0], w2 = ot[1];
#pragma unroll
    for (int kk = 0; kk < 4; kk++) {
      Pc[kk] = tile[idx + (int)(int8_t)(w1 >> (8 * kk))];
      idx += nc2;  // kindex = k0 + 2k (cc:197)
    }
#pragma unroll
    for (int k4 = 0; k4 < 41; k4++) {
      w1 = w2;
      if (k4 + 2 < 41) w2 = ot[k4 + 2];
      if (k4 + 1 < 41) {
#pragma unroll
        for (int kk = 0; kk < 4; kk++) {
          if (4 * (k4 + 1) + kk < UWSPR_NSYM) {
            Pn[kk] = tile[idx + (int)(int8_t)(w1 >> (8 * kk))];
            idx += nc2;
          }
        }
      }
#pragma unroll
      for (int kk = 0; kk < 4; kk++) {
        const int k = 4 * k4 + kk;
        if (k < UWSPR_NSYM) {
          const float4 P = Pc[kk];
          const float cm = (P.y + P.w) - (P.x + P.z);
          ss = pr3_bit(k) ? ss + cm : ss - cm;  // (2*pr3[k]-1)*cm, cc:207
          pw = pw + P.x; pw = pw + P.y; pw = pw + P.z; pw = pw + P.w;  // cc:209
        }
      }
#pragma unroll
      for (int kk = 0; kk < 4; kk++) Pc[kk] = Pn[kk];
    }
    syncbuf[g] = ieee_divf(ss, pw);  // cc:357,390
  }
  __syncthreads();
#ifdef K3_STAMPS
  t_c = clock64();
#endif

  const int hc = f.cell_hyps;
  if (syncgrid != nullptr && j < grid_cap) {
    float *gout = syncgrid + ((size_t)b * grid_cap + j) * f.ntot;
    for (int g = tid; g < f.ntot; g += K3_THREADS) {
      const int cell = g / hc, h = g - cell * hc;
      gout[g] = syncbuf[cell * f.umax + umap[(cell / UWSPR_NK0) * hc + h]];
    }
  }

  // ---- replay the running-best selection in reference order ---------------
  // The rule (cc:360 strict > for linear, cc:392 ratio against the RUNNING best
  // for nonlinear) is an order-dependent fold over all ntot hypotheses.  It is
  // replayed exactly, but cheaply:
  //  (1) all threads expand the metrics into reference order (`full`, reusing the
  //      tile / offset-table LDS, no longer needed) and reduce every 64-value
  //      slice to {max over linear, max and min over nonlinear} hypotheses;
  //  (2) wave 0 scans: 64 slice summaries at a time are tested against the
  //      current best -- the predicates are monotone in v (v > best; v/best > thr
  //      rises with v for best > 0 and falls for best < 0), so a slice whose
  //      extreme value fails cannot contain an acceptance and is skipped -- and
  //      only slices that may accept are scanned value by value with ballots.
  float *full = reinterpret_cast<float *>(smem);                       // [ntot]
  float *ssum = full + ((f.ntot + 63) & ~63);                          // [nslice][3]
  const int nslice = (f.ntot + 63) >> 6;
  {
    const int lane = tid & 63;
    // thread walks g = tid, tid + 1024, ...: (cell, h) advance by carry, no division in the loop
    const int qstep = K3_THREADS / hc, rstep = K3_THREADS - qstep * hc;
    int cell = tid / hc, h = tid - cell * hc;
    const float ninf = -__builtin_inff(), pinf = __builtin_inff();
    for (int g = tid; g < nslice * 64; g += K3_THREADS) {
      const bool in = g < f.ntot;
      const int ifr_i = cell / UWSPR_NK0;  // constant divisor
      const float v = in ? syncbuf[cell * f.umax + umap[ifr_i * hc + h]] : 0.0f;
      const bool lin = h < f.nlin;
      // NaN never satisfies a predicate: keep it out of the extremes
      const bool use = in && (v == v);
      float mlin = (use && lin) ? v : ninf;
      float mxnl = (use && !lin) ? v : ninf;
      float mnnl = (use && !lin) ? v : pinf;
#pragma unroll
      for (int o = 32; o > 0; o >>= 1) {
        mlin = fmaxf(mlin, __shfl_xor(mlin, o));
        mxnl = fmaxf(mxnl, __shfl_xor(mxnl, o));
        mnnl = fminf(mnnl, __shfl_xor(mnnl, o));
      }
      if (in) full[g] = v;
      const int sl = g >> 6;
      if (lane == 0) { ssum[3 * sl] = mlin; ssum[3 * sl + 1] = mxnl; ssum[3 * sl + 2] = mnnl; }
      cell += qstep; h += rstep;
      if (h >= hc) { h -= hc; cell += 1; }
    }
  }
  __syncthreads();
#ifdef K3_STAMPS
  long long t_c2 = clock64();
#endif
  if (tid < 64) {
    float best = -1e30f;
    int gbest = -1;
    int pos = 0;  // first slice not yet decided
    while (pos < nslice) {
      // find the first slice >= pos that may contain an acceptance
      int hit = -1;
      for (int blk = pos & ~63; blk < nslice && hit < 0; blk += 64) {
        const int sl = blk + tid;
        bool may = false;
        if (sl >= pos && sl < nslice) {
          const float mlin = ssum[3 * sl], mxnl = ssum[3 * sl + 1], mnnl = ssum[3 * sl + 2];
          may = mlin > best;
          if (best > 0.0f) may = may || (ieee_divf(mxnl, best) > f.threshold);
          else if (best < 0.0f) may = may || (ieee_divf(mnnl, best) > f.threshold);
          else may = true;  // best == +-0: scan exactly
        }
        const unsigned long long m = __ballot(may);
        if (m != 0ull) hit = blk + __ffsll((long long)m) - 1;
      }
      if (hit < 0) break;
      // exact scan of slice `hit`
      const int g = hit * 64 + tid;
      const bool in = g < f.ntot;
      const float v = in ? full[g] : 0.0f;
      const bool lin = in && (g % hc) < f.nlin;
      int start = 0;
      for (;;) {
        const bool pred = in && tid >= start &&
                          (lin ? (v > best) : (ieee_divf(v, best) > f.threshold));
        const unsigned long long mask = __ballot(pred);
        if (mask == 0ull) break;
        const int first = __ffsll((long long)mask) - 1;
        best = __shfl(v, first);
        gbest = hit * 64 + first;
        start = first + 1;
      }
      pos = hit + 1;
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
  __syncthreads();  // LDS is reused by the next work item
#ifdef K3_STAMPS
  t_d = clock64();
  if (tid == 0 && blockIdx.x == 7) printf("K3 stamps (cycles): stage %lld eval %lld expand %lld fold %lld\n", t_b - t_a, t_c - t_b, t_c2 - t_c, t_d - t_c2);
#endif
  }
}

size_t coarse_lds_bytes(const fdr_consts &f) { return k3_layout(f).total; }

void launch_coarse(uwspr_ctx *c, int B) {
  const fdr_consts &f = c->fc;
  prof_scope ps(c, UWSPR_K_COARSE, (int64_t)B);
  const long long items_max = (long long)f.cand_slots * B;
  const int grid = (int)std::min<long long>(items_max, c->num_cus);
  hipLaunchKernelGGL(k3_coarse, dim3(grid), dim3(K3_THREADS), coarse_lds_bytes(f), c->stream,
                     c->d_ps, f, c->d_off, c->d_umap, c->cur_cands, c->d_work, c->d_syncgrid,
                     c->d_syncgrid ? c->grid_cap : 0);
}

int coarse_configure(const fdr_consts &f) {
  size_t need = coarse_lds_bytes(f);
  if (need > 160 * 1024) return -1;
  // the attribute belongs to the function, not to a context: always the device maximum, so that a
  // second context with a smaller tile cannot lower the limit under an earlier one
  hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(k3_coarse),
                                     hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  return e == hipSuccess ? 0 : -2;
}

}  // namespace uwspr
