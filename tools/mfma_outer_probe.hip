// mfma_outer_probe.hip -- round 4: the matrix pipe as the exact binary32 MULTIPLIER of the tone correlations.
//
// v_mfma_f32_32x32x1_2b_f32 with C = 0 computes, for each of its two blocks, the 32 x 32 outer product
// D[i][j] = fma(A[i], B[j], 0) = fl(A[i] * B[j]) -- 2 048 correctly rounded products per instruction, 32 per lane, in
// 64 matrix-pipe cycles: the rate of v_mul_f32 at its best (2 cycles per wave-instruction and SIMD).  With
//   block 0:  A = x.re of 32 symbols,  B = [ c_0..c_15 | -s_0..-s_15 ]   (16 (tone, hypothesis) columns of one slot)
//   block 1:  A = x.im of 32 symbols,  B = [ s_0..s_15 |  c_0..c_15 ]
// lane j < 16 holds re*c (register r) and im*s (register r + 16) of 16 symbols for ITS column, lane j >= 16 holds
// -(re*s) and im*c: the reference's  inp = (inp + re*c) + im*s,  quad = (quad - re*s) + im*c  (cc:206-207) are two
// v_add_f32 per accumulator, 32 per MFMA, where the all-VALU form spends 64 instructions on the same 512 terms.
//
// This probe answers, before a kernel is built on it:
//   (1) the lane / register layout of the 2-block result (printed, and checked against the formula used below);
//   (2) is the product bit for bit v_mul_f32's -- random operands, denormal operands and products, zeros of both
//       signs, infinities, NaNs (their payloads are reported separately);
//   (3) the rate of [1 MFMA + 32 v_add_f32 on the previous MFMA's result] per wavefront against [64 v_mul/v_add] at
//       1..4 wavefronts per SIMD (the second is the loop body of the VALU kernels for the same 512 terms), and -- to
//       tell "the f32 MFMA occupies the vector ALU" from "an MFMA blocks its wavefront" -- the same 32 adds beside a
//       v_mfma_f32_32x32x16_bf16 (the dedicated matrix core) and alone.
// Diagnostic only; nothing here is on the product path.
//   hipcc -O3 -fno-slp-vectorize -ffp-contract=off --offload-arch=gfx950 tools/mfma_outer_probe.hip -o /tmp/mfma_outer_probe && /tmp/mfma_outer_probe
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>
#include <vector>
#pragma clang fp contract(off)
typedef float f32x32 __attribute__((ext_vector_type(32)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef int i32x4 __attribute__((ext_vector_type(4)));

__global__ void products(const float *a, const float *b, float *d) {
  const int l = threadIdx.x;
  f32x32 z = {};
  f32x32 r = __builtin_amdgcn_mfma_f32_32x32x1f32(a[l], b[l], z, 0, 0, 0);
#pragma unroll
  for (int i = 0; i < 32; i++) d[l * 32 + i] = r[i];
}

// element (block, row i, column j) of the result: lane and register
static inline void where(int blk, int i, int j, int *lane, int *reg) {
  *lane = j + 32 * ((i % 8) / 4);
  *reg = 16 * blk + 4 * (i / 8) + (i % 4);
}

// MODE 0: all-VALU body for 512 (symbol, column, sample) terms: 64 instructions (8 per term and lane / 64 lanes)
// MODE 1: one MFMA (products of the NEXT step, double-buffered) + 32 v_add_f32
template <int MODE>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, 8))) void rate(float *out, const float *xa, const float *xb, int iters, long long *cyc) {
  extern __shared__ float pad[];
  const int l = threadIdx.x & 63;
  float acc[16];
#pragma unroll
  for (int q = 0; q < 16; q++) acc[q] = 0.0f;
  float a = xa[l], b = xb[l];
  const long long t0 = __builtin_readcyclecounter();
  if (MODE == 0) {
    // a lane owns 8 terms per step: (inp, quad) of 4 hypotheses: 16 mul + 16 add ... x2 steps = 64 instructions
    float c[4], s[4];
#pragma unroll
    for (int q = 0; q < 4; q++) { c[q] = b + q; s[q] = b - q; }
    for (int it = 0; it < iters; it++) {
#pragma unroll
      for (int u = 0; u < 2; u++) {
        const float xr = a + u, xi = a - u;
#pragma unroll
        for (int q = 0; q < 4; q++) {
          acc[q] = (acc[q] + xr * c[q]) + xi * s[q];
          acc[4 + q] = (acc[4 + q] - xr * s[q]) + xi * c[q];
        }
      }
      asm volatile("" : "+v"(a));
    }
  } else if (MODE == 1) {
    // The MFMA as inline asm: with the builtin the compiler gives both product sets the same registers and waits
    // (s_nop 15) for each MFMA right behind it.  Here the NEXT step's MFMA is issued before this step's adds ("+v" on
    // the set being added keeps its adds behind that issue), so a result is read >= 32 adds after its MFMA was issued
    // -- far beyond the 18 wait states the ISA asks for between an XDL write and a VALU read.
    f32x32 p, pn;
    asm volatile("v_mfma_f32_32x32x1_2b_f32 %0, %1, %2, 0" : "=&v"(p) : "v"(a), "v"(b));
    for (int it = 0; it < iters; it += 2) {
      asm volatile("v_mfma_f32_32x32x1_2b_f32 %0, %2, %3, 0" : "=&v"(pn), "+v"(p) : "v"(a), "v"(b));
#pragma unroll
      for (int q = 0; q < 16; q++) acc[q] = (acc[q] + p[q]) + p[q + 16];
      asm volatile("v_mfma_f32_32x32x1_2b_f32 %0, %2, %3, 0" : "=&v"(p), "+v"(pn) : "v"(a), "v"(b));
#pragma unroll
      for (int q = 0; q < 16; q++) acc[q] = (acc[q] + pn[q]) + pn[q + 16];
    }
  }
  if (MODE == 2 || MODE == 3) {   // 32 adds per iteration beside one bf16 MFMA (2) or alone (3)
    f32x16 p = {}, pn = {};
    i32x4 ab = {0x3f803f80, 0x3f803f80, 0x3f803f80, 0x3f803f80};
    for (int it = 0; it < iters; it += 2) {
      if (MODE == 2) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %2, %2, 0" : "=&v"(pn), "+v"(p) : "v"(ab));
      else asm volatile("" : "+v"(pn), "+v"(p));
#pragma unroll
      for (int q = 0; q < 16; q++) acc[q] = (acc[q] + p[q]) + p[15 - q];
      if (MODE == 2) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %2, %2, 0" : "=&v"(p), "+v"(pn) : "v"(ab));
      else asm volatile("" : "+v"(pn), "+v"(p));
#pragma unroll
      for (int q = 0; q < 16; q++) acc[q] = (acc[q] + pn[q]) + pn[15 - q];
    }
  }
  const long long t1 = __builtin_readcyclecounter();
  float r = 0;
#pragma unroll
  for (int q = 0; q < 16; q++) r += acc[q];
  out[blockIdx.x * 256 + l] = r + pad[0] * 0.0f;
  if (l == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int MODE>
static void run_rate(int waves_per_simd, int iters) {
  // four wavefronts per workgroup (one per SIMD); LDS per workgroup sized so that exactly waves_per_simd of them fit a CU
  const int blocks = 256 * waves_per_simd;
  const size_t lds = (size_t)(160 * 1024 / waves_per_simd) - 1024;
  float *out, *xa, *xb; long long *cyc;
  hipMalloc(&out, (size_t)blocks * 256 * 4); hipMalloc(&xa, 256); hipMalloc(&xb, 256); hipMalloc(&cyc, (size_t)blocks * 8);
  std::vector<float> h(64);
  for (int i = 0; i < 64; i++) h[i] = 1.0f + 1e-3f * i;
  hipMemcpy(xa, h.data(), 256, hipMemcpyHostToDevice); hipMemcpy(xb, h.data(), 256, hipMemcpyHostToDevice);
  hipFuncSetAttribute((const void *)rate<MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  rate<MODE><<<blocks, 256, lds>>>(out, xa, xb, 64, cyc); hipDeviceSynchronize();
  hipEventRecord(e0);
  rate<MODE><<<blocks, 256, lds>>>(out, xa, xb, iters, cyc);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  std::vector<long long> hc(blocks);
  hipMemcpy(hc.data(), cyc, (size_t)blocks * 8, hipMemcpyDeviceToHost);
  double mean = 0; for (long long v : hc) mean += (double)v; mean /= blocks;
  // one iteration = 512 (symbol, column, sample) terms of one wavefront
  static const char *name[4] = {"64 v_mul/v_add (all VALU)", "1 f32 MFMA 32x32x1_2b + 32 v_add", "1 bf16 MFMA 32x32x16 + 32 v_add", "32 v_add alone"};
  printf("  %-34s %d wavefronts/SIMD: %7.3f ms = %6.1f ns per iteration and SIMD   %6.1f cycles per iteration and wavefront, %6.1f per SIMD (wavefront life / kernel time: %.0f MHz)\n",
         name[MODE], waves_per_simd, ms, ms * 1e6 / ((double)iters * waves_per_simd), mean / iters, mean / iters / waves_per_simd,
         mean / (ms * 1e-3) * 1e-6);
  hipFree(out); hipFree(xa); hipFree(xb); hipFree(cyc);
}

static uint32_t bits(float f) { uint32_t u; memcpy(&u, &f, 4); return u; }
static float fbits(uint32_t u) { float f; memcpy(&f, &u, 4); return f; }

int main() {
  float *da, *db, *dd; hipMalloc(&da, 256); hipMalloc(&db, 256); hipMalloc(&dd, 64 * 32 * 4);
  std::vector<float> a(64), b(64), d(64 * 32);
  // (1) layout: A = 1 + row (+64 for block 1), B = 1: D = row tag;  A = 1, B = 1 + column (+64 for block 1)
  printf("(1) layout of v_mfma_f32_32x32x1_2b_f32: lane l supplies A[block l/32][row l%%32] and B[block l/32][column l%%32]\n");
  int bad = 0;
  for (int pass = 0; pass < 2; pass++) {
    for (int l = 0; l < 64; l++) { a[l] = pass ? 1.0f : 1.0f + l; b[l] = pass ? 1.0f + l : 1.0f; }
    hipMemcpy(da, a.data(), 256, hipMemcpyHostToDevice); hipMemcpy(db, b.data(), 256, hipMemcpyHostToDevice);
    products<<<1, 64>>>(da, db, dd); hipMemcpy(d.data(), dd, d.size() * 4, hipMemcpyDeviceToHost);
    for (int blk = 0; blk < 2; blk++)
      for (int i = 0; i < 32; i++)
        for (int j = 0; j < 32; j++) {
          int lane, reg; where(blk, i, j, &lane, &reg);
          const float want = pass ? 1.0f + 32 * blk + j : 1.0f + 32 * blk + i;
          if (d[lane * 32 + reg] != want) bad++;
        }
  }
  printf("    element (block, i, j) sits in lane j + 32*((i%%8)/4), register 16*block + 4*(i/8) + i%%4: %s (%d elements off)\n",
         bad ? "NO" : "confirmed", bad);

  // (2) exactness
  printf("(2) D = fl(A*B) bit for bit?\n");
  uint64_t rng = 0x9E3779B97F4A7C15ull;
  auto next = [&]() { rng ^= rng << 13; rng ^= rng >> 7; rng ^= rng << 17; return (uint32_t)(rng >> 16); };
  long long tried = 0, diff = 0, nan_payload = 0, zero_sign = 0, denorm_tried = 0, denorm_diff = 0;
  for (int round = 0; round < 400; round++) {
    for (int l = 0; l < 64; l++) {
      const int kind = round % 8;
      uint32_t ua = next(), ub = next();
      if (kind == 0) { ua = (ua & 0x807FFFFFu) | (uint32_t)(100 + next() % 56) << 23; ub = (ub & 0x807FFFFFu) | (uint32_t)(100 + next() % 56) << 23; }  // ordinary
      if (kind == 1) { ua = (ua & 0x807FFFFFu) | (uint32_t)(120 + next() % 8) << 23; ub = (ub & 0x807FFFFFu) | (uint32_t)(120 + next() % 8) << 23; }    // |x| ~ 1
      if (kind == 2) { ua = (ua & 0x807FFFFFu) | (uint32_t)(1 + next() % 70) << 23; ub = (ub & 0x807FFFFFu) | (uint32_t)(50 + next() % 60) << 23; }    // products near / below the normal range
      if (kind == 3) { ua &= 0x807FFFFFu; ub = (ub & 0x807FFFFFu) | (uint32_t)(127 + next() % 30) << 23; }                                            // denormal A
      if (kind == 4) { if (l % 7 == 0) ua = 0; if (l % 7 == 1) ua = 0x80000000u; if (l % 5 == 0) ub = 0x80000000u; if (l % 5 == 1) ub = 0; }        // zeros
      if (kind == 5) { if (l % 9 == 0) ua = 0x7F800000u; if (l % 9 == 1) ua = 0xFF800000u; if (l % 11 == 0) ub = 0x7FC01234u; if (l % 11 == 1) ub = 0; } // inf, NaN, inf*0
      if (kind == 6) { ua = (ua & 0x807FFFFFu) | (uint32_t)(200 + next() % 54) << 23; ub = (ub & 0x807FFFFFu) | (uint32_t)(150 + next() % 60) << 23; } // overflow
      a[l] = fbits(ua); b[l] = fbits(ub);
    }
    hipMemcpy(da, a.data(), 256, hipMemcpyHostToDevice); hipMemcpy(db, b.data(), 256, hipMemcpyHostToDevice);
    products<<<1, 64>>>(da, db, dd); hipMemcpy(d.data(), dd, d.size() * 4, hipMemcpyDeviceToHost);
    for (int blk = 0; blk < 2; blk++)
      for (int i = 0; i < 32; i++)
        for (int j = 0; j < 32; j++) {
          int lane, reg; where(blk, i, j, &lane, &reg);
          volatile float x = a[32 * blk + i], y = b[32 * blk + j];
          volatile float w = x * y;   // the host's binary32 product (SSE: one rounding, denormals kept) = v_mul_f32 with denormals on
          const uint32_t got = bits(d[lane * 32 + reg]), want = bits(w);
          tried++;
          const bool dn = (want & 0x7F800000u) == 0 && (want & 0x007FFFFFu) != 0;
          const bool dn_in = ((bits(x) & 0x7F800000u) == 0 && (bits(x) & 0x7FFFFFu)) || ((bits(y) & 0x7F800000u) == 0 && (bits(y) & 0x7FFFFFu));
          if (dn || dn_in) denorm_tried++;
          if (got == want) continue;
          if (w != w && d[lane * 32 + reg] != d[lane * 32 + reg]) { nan_payload++; continue; }
          if ((got | want) == 0x80000000u) { zero_sign++; continue; }     // +0 against -0: C = +0 makes (-0) + (+0) = +0
          if (dn || dn_in) denorm_diff++;
          if (diff < 8) printf("    DIFFERENT: %08x * %08x -> mfma %08x  v_mul %08x\n", bits(x), bits(y), got, want);
          diff++;
        }
  }
  printf("    %lld products: %lld different (%lld of them with a denormal operand or result, %lld such cases tried);\n"
         "    apart from those: %lld NaNs with another payload / sign, %lld zeros with the other sign (fma(a, b, +0): harmless,\n"
         "    an accumulator that starts at +0 never becomes -0)\n", tried, diff, denorm_diff, denorm_tried, nan_payload, zero_sign);

  // (3) rate
  printf("(3) cycles for the 512 terms of one step (32 symbols x 16 columns x {inp, quad}):\n");
  for (int w = 1; w <= 4; w++) { run_rate<0>(w, 4000); run_rate<1>(w, 4000); run_rate<2>(w, 4000); run_rate<3>(w, 4000); }
  return 0;
}
