// tab_probe.hip -- issue rate of the "tabled" tone-correlation walk: lane = symbol, wave = tone,
// phasors wave-uniform in SGPRs (s_load from a table), samples one ds_read_b128 per two steps,
// H hypotheses per lane (8 binary32 ops per hypothesis and sample, no FMA).  Diagnostic only.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
#pragma clang fp contract(off)

typedef float f8 __attribute__((ext_vector_type(8)));
typedef float f16v __attribute__((ext_vector_type(16)));
#define CONSTAS __attribute__((address_space(4)))

template <int H, int NW, int STEPS>   // STEPS per s_load: 4 (x8) or 8 (x16)
__global__ __launch_bounds__(64 * NW) void probe(const float *__restrict__ tabg, float *out, int iters) {
  __shared__ __align__(16) float smp[(NW + 3) / 4][2 * 64 * 36];
  const int lane = threadIdx.x & 63;
  const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  for (int k = 0; k < 72; k++) smp[wv >> 2][lane * 72 + k] = 0.001f * k + lane;
  __syncthreads();
  float inp[H], quad[H];
#pragma unroll
  for (int q = 0; q < H; q++) { inp[q] = 0; quad[q] = 0; }
  const CONSTAS float *tab = (const CONSTAS float *)(tabg) + (size_t)(wv & 3) * 512;
  for (int it = 0; it < iters; it++) {
#pragma unroll
    for (int c = 0; c < 16 / STEPS; c++) {
      // H phasor runs of STEPS steps each
      float ph[H][2 * STEPS];
#pragma unroll
      for (int q = 0; q < H; q++) {
        if (STEPS == 4) {
          f8 v = *(const CONSTAS f8 *)(tab + q * 2048 + ((it & 15) * 16 + c * STEPS) * 2);
#pragma unroll
          for (int e = 0; e < 8; e++) ph[q][e] = v[e];
        } else {
          f16v v = *(const CONSTAS f16v *)(tab + q * 2048 + ((it & 15) * 16 + c * STEPS) * 2);
#pragma unroll
          for (int e = 0; e < 16; e++) ph[q][e] = v[e];
        }
      }
#pragma unroll
      for (int k = 0; k < STEPS; k += 2) {
        const float4 x = *reinterpret_cast<const float4 *>(&smp[wv >> 2][(it & 1) * 64 * 36 + lane * 36 + 2 * (c * STEPS + k)]);
#pragma unroll
        for (int q = 0; q < H; q++) {
          const float c0 = ph[q][2 * k], s0 = ph[q][2 * k + 1], c1 = ph[q][2 * k + 2], s1 = ph[q][2 * k + 3];
          inp[q] = (inp[q] + x.x * c0) + x.y * s0;
          quad[q] = (quad[q] - x.x * s0) + x.y * c0;
          inp[q] = (inp[q] + x.z * c1) + x.w * s1;
          quad[q] = (quad[q] - x.z * s1) + x.w * c1;
        }
      }
    }
  }
  float r = 0;
#pragma unroll
  for (int q = 0; q < H; q++) r += inp[q] + quad[q];
  out[blockIdx.x * 64 * NW + threadIdx.x] = r;
}

template <int H, int NW, int STEPS>
void run(int wgs_per_cu, int iters, const float *tab) {
  int blocks = 256 * wgs_per_cu;
  float *out; hipMalloc(&out, (size_t)blocks * 64 * NW * 4);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  probe<H, NW, STEPS><<<blocks, 64 * NW>>>(tab, out, 8);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  probe<H, NW, STEPS><<<blocks, 64 * NW>>>(tab, out, iters);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  double ops = (double)blocks * 64 * NW * iters * 16 * 8 * H;
  printf("H=%d NW=%d steps/s_load=%d WG/CU=%d waves/SIMD=%.1f  %.3f ms  %.1f Tops/s  wave-instr/cycle/SIMD@2.4GHz=%.3f\n",
         H, NW, STEPS, wgs_per_cu, NW * wgs_per_cu / 4.0, ms, ops / ms / 1e9, ops / 64 / (ms * 1e-3) / 1024 / 2.4e9);
  hipFree(out);
}

int main() {
  float *tab; hipMalloc(&tab, 64 * 2048 * 4);
  std::vector<float> h(64 * 2048);
  for (size_t i = 0; i < h.size(); i++) h[i] = 0.5f + 1e-4f * (i % 977);
  hipMemcpy(tab, h.data(), h.size() * 4, hipMemcpyHostToDevice);
  const int it = 2048;
  run<4, 8, 4>(1, it, tab);  run<4, 12, 4>(1, it, tab); run<4, 16, 4>(1, it, tab);
  run<4, 8, 4>(2, it, tab);  run<4, 12, 4>(2, it, tab); run<4, 16, 4>(2, it, tab);
  run<4, 4, 4>(1, it, tab);
  run<4, 8, 8>(1, it, tab);  run<4, 12, 8>(1, it, tab); run<4, 16, 8>(1, it, tab);
  run<4, 12, 8>(2, it, tab);
  run<2, 8, 4>(1, it, tab);  run<2, 12, 4>(1, it, tab); run<2, 16, 4>(1, it, tab); run<2, 12, 4>(2, it, tab);
  run<6, 8, 4>(1, it, tab);  run<6, 12, 4>(1, it, tab); run<6, 12, 4>(2, it, tab);
  run<1, 12, 8>(1, it, tab); run<1, 12, 8>(2, it, tab);
  return 0;
}
