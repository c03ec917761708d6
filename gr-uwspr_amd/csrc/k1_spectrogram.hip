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
// A wavefront walks rpw CONSECUTIVE rows and keeps the raw samples in registers: consecutive rows
// overlap by 384 of 512 samples, and in this lane mapping (lane L holds samples L + 64 q, q = 0..7)
// the next row's blocks q = 0..5 are this row's q = 2..7 -- so a row costs two new 512-B coalesced
// loads per lane instead of eight, requested one row ahead of their use.
// Bound: FP32 VALU issue + LDS exchange, which add on this chip (340 flops + 32 ds_*_b64 per row;
// DESIGN 5.1).  Algorithmic bytes/frame = 360000 in + 348*band_w*4 out: HBM is ~25 % busy.
#include "uwspr_internal.h"
#include "fft512_lane.h"

#include <stdlib.h>

#include <type_traits>

#pragma clang fp contract(off)

namespace uwspr {

constexpr int K1_WAVES = 4;
// rows per wavefront: 29 for large batches (348 = 12 x 29: 256 frames are 3072 wavefronts = one
// round at 3 per SIMD, 2.2 loads per row), 6 for small ones (58 wavefronts per frame: latency)
constexpr int K1_ROWS_LARGE = 29, K1_ROWS_SMALL = 6, K1_LARGE_BATCH = 64;

// the exchange images are per wavefront: the wavefronts of a workgroup need no common barrier
__device__ __forceinline__ void k1_wave_fence() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

__device__ __forceinline__ void k1_landed(float2 &v) { asm volatile("" : "+v"(v.x), "+v"(v.y)); }
// register-slot constants of the three index families (p = lane part ^ slot part, bitwise disjoint)
__device__ __forceinline__ constexpr int k1_slotA(int r) { return r; }
__device__ __forceinline__ constexpr int k1_slotB(int e) { return xidx(8 * e); }
__device__ __forceinline__ constexpr int k1_slotC(int a) { return xidx(64 * a); }

#ifdef UWSPR_K1_STAMPS   // tools/k1_stamps.py: per-wavefront begin/end on the 100 MHz wall clock + hardware id
__device__ unsigned long long k1_stamp_buf[3 * 16384];
#define K1_STAMP(slot)                                                                      \
  if (L == 0) {                                                                             \
    const int wid = blockIdx.x * K1_WAVES + wv;                  \
    if (wid < 16384) k1_stamp_buf[3 * wid + (slot)] =                                       \
        (slot) == 2 ? ((unsigned long long)__builtin_amdgcn_s_getreg(20 | (31 << 11)) << 32) |      \
                      (unsigned)__builtin_amdgcn_s_getreg(4 | (31 << 11))                   \
                    : wall_clock64();                                                       \
  }
#else
#define K1_STAMP(slot)
#endif

// NARROW: the band lies within columns 192..319 (bins 0..63 and 448..511): pass C computes only
// register slots 0 and 7 (fft512_lane.h: pass_c_narrow), 9 % of the kernel's arithmetic less.
template <bool NARROW>
__global__ __launch_bounds__(64 * K1_WAVES, 3) void k1_spectrogram(
    const float2 *__restrict__ frames, int B, int rpw, int fstride, int n, const float *__restrict__ window,
    const float2 *__restrict__ twiddle, float *__restrict__ ps, int band_lo, int band_w,
    int32_t *__restrict__ work_count) {
  __shared__ cpx tw_s[256];
  __shared__ cpx xch[K1_WAVES][XCHG_LEN];

  const int tid = threadIdx.x;
  const int L = tid & 63;
  const int wv = tid >> 6;
  // wave item = (frame, group of rpw consecutive rows), flattened so that every
  // workgroup (and with the round-robin dispatch every XCD) gets the same number of working waves
  const int gpf = (n + rpw - 1) / rpw;
  const int item = blockIdx.x * K1_WAVES + wv;
  const int b = item / gpf;
  // K2 (next on the stream) appends to the coarse-search work list: reset its counter
  if (blockIdx.x == 0 && tid == 0) *work_count = 0;

  K1_STAMP(0) K1_STAMP(2)
  tw_s[tid] = cpx{twiddle[tid].x, twiddle[tid].y};
  __syncthreads();

  float wreg[8];
#pragma unroll
  for (int r = 0; r < 8; r++) wreg[r] = window[in_sample(L, r)];
  const pass_tw twB = load_pass_tw(tw_s, L & 7, 8);
  const pass_tw twC = load_pass_tw(tw_s, L, 64);
  const cpx w64 = tw_s[64], w192 = tw_s[192];

  const float2 *x = frames + (size_t)b * fstride;
  cpx *lds = xch[wv];
  const int row0 = (item - b * gpf) * rpw;
  if (b >= B) { K1_STAMP(1) return; }   // wave-uniform; no workgroup barrier below

  // raw[(q + 2 ph) & 7] = x[128 row + L + 64 q] with ph = row phase: the next row's two new blocks
  // land in the two slots this row's q = 0, 1 leave dead after the window multiply -- no moves
  float2 raw[8];
  // xidx is linear over XOR and lane / slot bits are disjoint: xidx(lane | slot) = xidx(lane) ^ xidx(slot)
  const int laneA = xidx(posA(L, 0)), laneB = xidx(posB(L, 0)), laneC = xidx(posC(L, 0));
#pragma unroll
  for (int q = 0; q < 8; q++) raw[q] = x[row0 * 128 + L + 64 * q];
  const int rend = min(row0 + rpw, n);
  auto do_row = [&](int row, auto PH) {
    constexpr int ph = decltype(PH)::value;
    cpx y[8];
#pragma unroll
    for (int r = 0; r < 8; r++) {
      const float2 sv = raw[(rev3(r) + 2 * ph) & 7];   // in_sample(L, r) = L + 64 rev3(r)
      // FDR_impl.cc:230-231: one binary32 multiply per component
      y[r].r = sv.x * wreg[r];
      y[r].i = sv.y * wreg[r];
    }
    // the next row's two blocks: requested here, in flight during this row's arithmetic and landed
    // (k1_landed) before this row's stores -- left to the compiler the loads sink to the next row's
    // first use and every row waits out a full memory latency; waiting after the stores would also
    // wait for the stores (vmcnt counts them).  The wave's last row re-reads its own row (no branch).
    const float2 *nx = x + (row + 1 < rend ? row + 1 : row) * 128 + L;
    raw[(2 * ph) & 7] = nx[64 * 6];
    raw[(2 * ph + 1) & 7] = nx[64 * 7];
    __builtin_amdgcn_sched_barrier(0);
    pass_a(y, w64, w192);
#pragma unroll
    for (int r = 0; r < 8; r++) lds[laneA ^ k1_slotA(r)] = y[r];
    k1_wave_fence();
#pragma unroll
    for (int e = 0; e < 8; e++) y[e] = lds[laneB ^ k1_slotB(e)];
    pass_bc(y, twB);
    k1_wave_fence();
#pragma unroll
    for (int e = 0; e < 8; e++) lds[laneB ^ k1_slotB(e)] = y[e];
    k1_wave_fence();
#pragma unroll
    for (int e = 0; e < 8; e++) y[e] = lds[laneC ^ k1_slotC(e)];
    if (NARROW) pass_c_narrow(y, twC); else pass_bc(y, twC);
    k1_landed(raw[(2 * ph) & 7]); k1_landed(raw[(2 * ph + 1) & 7]);
    float *out = ps + ((size_t)b * n + row) * band_w;
#pragma unroll
    for (int e = 0; e < 8; e++) {
      if (NARROW && e != 0 && e != 7) continue;
      int col = out_col(L, e) - band_lo;
      // FDR_impl.cc:252: re*re + im*im, two products, one add, no fusion
      if (col >= 0 && col < band_w) out[col] = y[e].r * y[e].r + y[e].i * y[e].i;
    }
    k1_wave_fence();
  };
  for (int row = row0; row < rend; row += 4) {   // wave-uniform bounds
    do_row(row, std::integral_constant<int, 0>{});
    if (row + 1 < rend) do_row(row + 1, std::integral_constant<int, 1>{});
    if (row + 2 < rend) do_row(row + 2, std::integral_constant<int, 2>{});
    if (row + 3 < rend) do_row(row + 3, std::integral_constant<int, 3>{});
  }
  K1_STAMP(1)
}

#ifdef UWSPR_K1_STAMPS
extern "C" int uwspr_debug_k1_stamps(unsigned long long *out, int nwaves) {
  return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(k1_stamp_buf), sizeof(unsigned long long) * 3 * nwaves);
}
#endif

void launch_spectrogram(uwspr_ctx *c, const float *frames, int B) {
  const fdr_consts &f = c->fc;
  prof_scope ps(c, UWSPR_K_SPECTROGRAM, B);
  int rpw = B >= K1_LARGE_BATCH ? K1_ROWS_LARGE : K1_ROWS_SMALL;
  if (c->opt[UWSPR_OPT_K1_ROWS] > 0) rpw = c->opt[UWSPR_OPT_K1_ROWS];   // (option "k1_rows": A/B only)
  const int items = B * ((f.n + rpw - 1) / rpw);
  dim3 grid((items + K1_WAVES - 1) / K1_WAVES);
  const bool narrow = f.band_lo >= 192 && f.band_lo + f.band_w <= 320;
  auto go = [&](auto kern) {
    hipLaunchKernelGGL(kern, grid, dim3(64 * K1_WAVES), 0, c->stream,
                       (const float2 *)frames, B, rpw, c->fstride, f.n, c->d_window,
                       (const float2 *)c->d_twiddle, c->d_ps, f.band_lo, f.band_w, c->d_work);
  };
  if (narrow) go(k1_spectrogram<true>); else go(k1_spectrogram<false>);
}

}  // namespace uwspr
