// tab_probe2.hip -- as tab_probe, with the next chunk's phasor s_loads and sample ds_reads issued
// BEFORE the current chunk's arithmetic (double-buffered SGPR/VGPR sets).  Diagnostic only.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
#pragma clang fp contract(off)

typedef float f8 __attribute__((ext_vector_type(8)));
#define CONSTAS __attribute__((address_space(4)))

template <int H>
struct chunk_t { f8 ph[H]; float4 x[2]; };

template <int H>
__device__ __forceinline__ void fetch(chunk_t<H> &d, const CONSTAS float *tab, const float *row, int step) {
#pragma unroll
  for (int q = 0; q < H; q++) d.ph[q] = *(const CONSTAS f8 *)(tab + q * 2048 + step * 2);
  d.x[0] = *reinterpret_cast<const float4 *>(row + 2 * (step & 15));
  d.x[1] = *reinterpret_cast<const float4 *>(row + 2 * (step & 15) + 4);
}
template <int H>
__device__ __forceinline__ void mac(const chunk_t<H> &d, float (&inp)[H], float (&quad)[H]) {
#pragma unroll
  for (int k = 0; k < 4; k += 2) {
    const float4 x = d.x[k >> 1];
#pragma unroll
    for (int q = 0; q < H; q++) {
      const float c0 = d.ph[q][2 * k], s0 = d.ph[q][2 * k + 1], c1 = d.ph[q][2 * k + 2], s1 = d.ph[q][2 * k + 3];
      inp[q] = (inp[q] + x.x * c0) + x.y * s0;
      quad[q] = (quad[q] - x.x * s0) + x.y * c0;
      inp[q] = (inp[q] + x.z * c1) + x.w * s1;
      quad[q] = (quad[q] - x.z * s1) + x.w * c1;
    }
  }
}

template <int H, int NW>
__global__ __launch_bounds__(64 * NW) void probe(const float *__restrict__ tabg, float *out, int iters) {
  __shared__ __align__(16) float smp[(NW + 3) / 4][2 * 64 * 36];
  const int lane = threadIdx.x & 63;
  const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  for (int k = 0; k < 72; k++) smp[wv >> 2][lane * 72 + k] = 0.001f * k + lane;
  __syncthreads();
  float inp[H], quad[H];
#pragma unroll
  for (int q = 0; q < H; q++) { inp[q] = 0; quad[q] = 0; }
  const CONSTAS float *tab = (const CONSTAS float *)(tabg) + (size_t)(wv & 3) * 512 + (size_t)(blockIdx.x & 7) * 8192;
  chunk_t<H> A, B;
  fetch<H>(A, tab, &smp[wv >> 2][lane * 36], 0);
  for (int it = 0; it < iters; it++) {   // 16 steps per iteration = 4 chunks of 4
    const float *row = &smp[wv >> 2][(it & 1) * 64 * 36 + lane * 36];
    const int s0 = (it & 15) * 16;
    fetch<H>(B, tab, row, s0 + 4);  mac<H>(A, inp, quad);
    fetch<H>(A, tab, row, s0 + 8);  mac<H>(B, inp, quad);
    fetch<H>(B, tab, row, s0 + 12); mac<H>(A, inp, quad);
    fetch<H>(A, tab, row, (s0 + 16) & 255); mac<H>(B, inp, quad);
  }
  float r = 0;
#pragma unroll
  for (int q = 0; q < H; q++) r += inp[q] + quad[q];
  out[blockIdx.x * 64 * NW + threadIdx.x] = r;
}

template <int H, int NW>
void run(int wgs_per_cu, int iters, const float *tab) {
  int blocks = 256 * wgs_per_cu;
  float *out; (void)hipMalloc(&out, (size_t)blocks * 64 * NW * 4);
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  probe<H, NW><<<blocks, 64 * NW>>>(tab, out, 8);
  (void)hipDeviceSynchronize();
  (void)hipEventRecord(e0);
  probe<H, NW><<<blocks, 64 * NW>>>(tab, out, iters);
  (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
  float ms; (void)hipEventElapsedTime(&ms, e0, e1);
  double ops = (double)blocks * 64 * NW * iters * 16 * 8 * H;
  printf("prefetch H=%d NW=%d WG/CU=%d waves/SIMD=%.1f  %.3f ms  %.1f Tops/s  wave-instr/cycle/SIMD@2.4GHz=%.3f\n",
         H, NW, wgs_per_cu, NW * wgs_per_cu / 4.0, ms, ops / ms / 1e9, ops / 64 / (ms * 1e-3) / 1024 / 2.4e9);
  (void)hipFree(out);
}

int main() {
  float *tab; (void)hipMalloc(&tab, 64 * 8192 * 4);
  std::vector<float> h(64 * 8192);
  for (size_t i = 0; i < h.size(); i++) h[i] = 0.5f + 1e-4f * (i % 977);
  (void)hipMemcpy(tab, h.data(), h.size() * 4, hipMemcpyHostToDevice);
  const int it = 2048;
  run<4, 8>(1, it, tab);  run<4, 12>(1, it, tab); run<4, 16>(1, it, tab);
  run<4, 8>(2, it, tab);  run<4, 12>(2, it, tab); run<4, 16>(2, it, tab);
  run<2, 8>(1, it, tab);  run<2, 12>(1, it, tab); run<2, 16>(1, it, tab); run<2, 12>(2, it, tab);
  run<3, 12>(1, it, tab); run<3, 12>(2, it, tab);
  run<5, 12>(1, it, tab); run<5, 12>(2, it, tab);
  run<1, 12>(1, it, tab); run<1, 12>(2, it, tab); run<1, 16>(2, it, tab);
  return 0;
}
