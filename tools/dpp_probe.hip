// dpp_probe.hip -- issue rate of v_mul_f32 with a DPP row_newbcast source (phasor broadcast from a
// lane of each 16-lane row) against the plain VGPR form.  Diagnostic only.
#include <hip/hip_runtime.h>
#include <stdio.h>
#pragma clang fp contract(off)

template <int N>
__device__ __forceinline__ float mul_bc(float ph, float x) {
  float d;
  asm("v_mul_f32_dpp %0, %1, %2 row_newbcast:%3 row_mask:0xf bank_mask:0xf" : "=v"(d) : "v"(ph), "v"(x), "n"(N));
  return d;
}

template <int H, bool DPP, int NW>
__global__ __launch_bounds__(64 * NW) void probe(float *out, int iters, float seed) {
  float inp[H], quad[H], ph[H];
#pragma unroll
  for (int q = 0; q < H; q++) { inp[q] = 0.001f * threadIdx.x; quad[q] = 0; ph[q] = seed + 1e-3f * (threadIdx.x & 15) + q; }
  float x0 = 0.5f + threadIdx.x, y0 = 0.25f;
  for (int it = 0; it < iters; it++) {
#pragma unroll
    for (int q = 0; q < H; q++) {
#define STEP(J)                                                                   \
      if (DPP) {                                                                  \
        inp[q] = (inp[q] + mul_bc<2 * J>(ph[q], x0)) + mul_bc<2 * J + 1>(ph[q], y0);   \
        quad[q] = (quad[q] - mul_bc<2 * J + 1>(ph[q], x0)) + mul_bc<2 * J>(ph[q], y0); \
      } else {                                                                    \
        inp[q] = (inp[q] + ph[q] * x0) + (ph[q] + 1.0f) * y0;                      \
        quad[q] = (quad[q] - (ph[q] + 1.0f) * x0) + ph[q] * y0;                    \
      }
      STEP(0) STEP(1) STEP(2) STEP(3) STEP(4) STEP(5) STEP(6) STEP(7)
    }
    x0 += 1.0f; y0 -= 1.0f;
  }
  float r = 0;
#pragma unroll
  for (int q = 0; q < H; q++) r += inp[q] + quad[q];
  out[blockIdx.x * 64 * NW + threadIdx.x] = r;
}

template <int H, bool DPP, int NW>
void run(int wgs_per_cu, int iters) {
  int blocks = 256 * wgs_per_cu;
  float *out; (void)hipMalloc(&out, (size_t)blocks * 64 * NW * 4);
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  probe<H, DPP, NW><<<blocks, 64 * NW>>>(out, 8, 0.5f);
  (void)hipDeviceSynchronize();
  (void)hipEventRecord(e0);
  probe<H, DPP, NW><<<blocks, 64 * NW>>>(out, iters, 0.5f);
  (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
  float ms; (void)hipEventElapsedTime(&ms, e0, e1);
  double ops = (double)blocks * 64 * NW * iters * 8 * 8 * H;
  printf("H=%d dpp=%d NW=%d waves/SIMD=%.1f  %.3f ms  %.1f Tops/s  wave-instr/cycle/SIMD@2.4GHz=%.3f\n",
         H, (int)DPP, NW, NW * wgs_per_cu / 4.0, ms, ops / ms / 1e9, ops / 64 / (ms * 1e-3) / 1024 / 2.4e9);
  (void)hipFree(out);
}

int main() {
  const int it = 8192;
  run<3, true, 8>(1, it);  run<3, true, 12>(1, it); run<3, true, 16>(1, it); run<3, true, 16>(2, it);
  run<3, false, 16>(1, it);
  return 0;
}
