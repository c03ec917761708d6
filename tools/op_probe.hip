// op_probe.hip -- does a VALU op with an SGPR source issue slower than an all-VGPR one on gfx950?
// No memory traffic in the loop.  Diagnostic only.
#include <hip/hip_runtime.h>
#include <stdio.h>
#pragma clang fp contract(off)

template <int H, bool SGPR, int NW>
__global__ __launch_bounds__(64 * NW) void probe(float *out, int iters, float c0, float s0, float c1, float s1) {
  float inp[H], quad[H];
#pragma unroll
  for (int q = 0; q < H; q++) { inp[q] = 0.001f * threadIdx.x; quad[q] = 0; }
  float x0 = 0.5f + threadIdx.x, y0 = 0.25f, x1 = 0.125f, y1 = 1.0f + threadIdx.x;
  float pc0 = c0, ps0 = s0, pc1 = c1, ps1 = s1;
  if (!SGPR) {  // make them per-lane values (VGPRs)
    pc0 += 1e-6f * threadIdx.x; ps0 += 1e-6f * threadIdx.x; pc1 -= 1e-6f * threadIdx.x; ps1 -= 1e-6f * threadIdx.x;
  }
  for (int it = 0; it < iters; it++) {
#pragma unroll
    for (int k = 0; k < 8; k++) {
#pragma unroll
      for (int q = 0; q < H; q++) {
        inp[q] = (inp[q] + x0 * pc0) + y0 * ps0;
        quad[q] = (quad[q] - x0 * ps0) + y0 * pc0;
        inp[q] = (inp[q] + x1 * pc1) + y1 * ps1;
        quad[q] = (quad[q] - x1 * ps1) + y1 * pc1;
      }
    }
    x0 += 1.0f; y1 -= 1.0f;   // keep the loop from being folded
  }
  float r = 0;
#pragma unroll
  for (int q = 0; q < H; q++) r += inp[q] + quad[q];
  out[blockIdx.x * 64 * NW + threadIdx.x] = r;
}

template <int H, bool SGPR, int NW>
void run(int wgs_per_cu, int iters) {
  int blocks = 256 * wgs_per_cu;
  float *out; (void)hipMalloc(&out, (size_t)blocks * 64 * NW * 4);
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  probe<H, SGPR, NW><<<blocks, 64 * NW>>>(out, 8, 0.999f, 0.01f, 0.998f, 0.02f);
  (void)hipDeviceSynchronize();
  (void)hipEventRecord(e0);
  probe<H, SGPR, NW><<<blocks, 64 * NW>>>(out, iters, 0.999f, 0.01f, 0.998f, 0.02f);
  (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
  float ms; (void)hipEventElapsedTime(&ms, e0, e1);
  double ops = (double)blocks * 64 * NW * iters * 8 * 16 * H;
  printf("H=%d sgpr=%d NW=%d WG/CU=%d waves/SIMD=%.1f  %.3f ms  %.1f Tops/s  wave-instr/cycle/SIMD@2.4GHz=%.3f\n",
         H, (int)SGPR, NW, wgs_per_cu, NW * wgs_per_cu / 4.0, ms, ops / ms / 1e9, ops / 64 / (ms * 1e-3) / 1024 / 2.4e9);
  (void)hipFree(out);
}

int main() {
  const int it = 4096;
  run<4, true, 8>(1, it);  run<4, false, 8>(1, it);
  run<4, true, 12>(1, it); run<4, false, 12>(1, it);
  run<4, true, 16>(1, it); run<4, false, 16>(1, it);
  run<4, true, 12>(2, it); run<4, false, 12>(2, it);
  run<4, true, 16>(2, it); run<4, false, 16>(2, it);
  run<4, true, 4>(1, it);  run<4, false, 4>(1, it);
  return 0;
}
