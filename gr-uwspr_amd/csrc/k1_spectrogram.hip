// K1 -- sliding-window power spectrogram.
//
// Reference: FDR_impl::transform hot loop 1, lib/FDR_impl.cc:222-254 (window
// cc:101-105, FFTW plan cc:123-132):  ps[i][j] = |FFT512(x[128i..128i+511]*w)|^2
// with fftshift, i < n (348).  Only the columns the rest of the path ever reads
// (band_lo .. band_lo+band_w-1) are written to HBM.
//
// Mapping: one wavefront per row, 8 points per lane, three register radix-8
// passes (fft512_lane.h) with two exchanges through a per-wave LDS image; the
// 256-entry twiddle table is staged in LDS once per workgroup and each lane
// keeps its 14 pass-B/C twiddles and 8 window taps in VGPRs across rows.
// Loads of x are 512-B coalesced per register slot; the 4x overlap between
// consecutive rows is served by L1/L2 (a workgroup walks 16 adjacent rows).
// Roofline: HBM (8.3 flop/B), algorithmic bytes/frame = 360000 + 348*512*4.
#include "uwspr_internal.h"
#include "fft512_lane.h"

#pragma clang fp contract(off)

namespace uwspr {

constexpr int K1_WAVES = 4;
constexpr int K1_ROWS_PER_WAVE = 4;

__global__ __launch_bounds__(64 * K1_WAVES) void k1_spectrogram(
    const float2 *__restrict__ frames, int fl, int n, const float *__restrict__ window,
    const float2 *__restrict__ twiddle, float *__restrict__ ps, int band_lo, int band_w,
    int32_t *__restrict__ work_count) {
  __shared__ cpx tw_s[256];
  __shared__ cpx xch[K1_WAVES][XCHG_LEN];

  const int tid = threadIdx.x;
  const int L = tid & 63;
  const int wv = tid >> 6;
  const int b = blockIdx.y;
  // K2 (next on the stream) appends to the coarse-search work list: reset its counter
  if (blockIdx.x == 0 && b == 0 && tid == 0) *work_count = 0;

  tw_s[tid] = cpx{twiddle[tid].x, twiddle[tid].y};
  __syncthreads();

  float wreg[8];
#pragma unroll
  for (int r = 0; r < 8; r++) wreg[r] = window[in_sample(L, r)];
  const pass_tw twB = load_pass_tw(tw_s, L & 7, 8);
  const pass_tw twC = load_pass_tw(tw_s, L, 64);
  const cpx w64 = tw_s[64], w192 = tw_s[192];

  const float2 *x = frames + (size_t)b * fl;
  cpx *lds = xch[wv];
  const int row0 = (blockIdx.x * K1_WAVES + wv) * K1_ROWS_PER_WAVE;

  for (int rr = 0; rr < K1_ROWS_PER_WAVE; rr++) {
    const int row = row0 + rr;
    const bool live = row < n;  // wave-uniform
    cpx y[8];
    if (live) {
#pragma unroll
      for (int r = 0; r < 8; r++) {
        float2 s = x[row * 128 + in_sample(L, r)];
        // FDR_impl.cc:230-231: one binary32 multiply per component
        y[r].r = s.x * wreg[r];
        y[r].i = s.y * wreg[r];
      }
      pass_a(y, w64, w192);
#pragma unroll
      for (int r = 0; r < 8; r++) lds[xidx(posA(L, r))] = y[r];
    }
    __syncthreads();
    if (live) {
#pragma unroll
      for (int e = 0; e < 8; e++) y[e] = lds[xidx(posB(L, e))];
      pass_bc(y, twB);
    }
    __syncthreads();
    if (live) {
#pragma unroll
      for (int e = 0; e < 8; e++) lds[xidx(posB(L, e))] = y[e];
    }
    __syncthreads();
    if (live) {
#pragma unroll
      for (int e = 0; e < 8; e++) y[e] = lds[xidx(posC(L, e))];
      pass_bc(y, twC);
      float *out = ps + ((size_t)b * n + row) * band_w;
#pragma unroll
      for (int e = 0; e < 8; e++) {
        int col = out_col(L, e) - band_lo;
        // FDR_impl.cc:252: re*re + im*im, two products, one add, no fusion
        if (col >= 0 && col < band_w) out[col] = y[e].r * y[e].r + y[e].i * y[e].i;
      }
    }
    __syncthreads();
  }
}

void launch_spectrogram(uwspr_ctx *c, const float *frames, int B) {
  const fdr_consts &f = c->fc;
  prof_scope ps(c, UWSPR_K_SPECTROGRAM, B);
  dim3 grid((f.n + K1_WAVES * K1_ROWS_PER_WAVE - 1) / (K1_WAVES * K1_ROWS_PER_WAVE), B);
  hipLaunchKernelGGL(k1_spectrogram, grid, dim3(64 * K1_WAVES), 0, c->stream,
                     (const float2 *)frames, f.fl, f.n, c->d_window,
                     (const float2 *)c->d_twiddle, c->d_ps, f.band_lo, f.band_w, c->d_work);
}

}  // namespace uwspr
