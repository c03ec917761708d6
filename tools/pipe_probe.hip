// pipe_probe.hip -- the tabled walk (s_load phasor run + ds_read samples + 64 multiply-adds per
// row-item) in three shapes: (0) loads, wait, arithmetic; (1) software-pipelined by row-item:
// wait-all, request the next row-item, arithmetic; (2) shape 0 with half of the wavefronts
// delayed by half an item (stagger).  Diagnostic only.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
#pragma clang fp contract(off)
typedef float f16v __attribute__((ext_vector_type(16)));
typedef float v4f __attribute__((ext_vector_type(4)));
#define CONSTAS __attribute__((address_space(4)))
#define LDSAS __attribute__((address_space(3)))

__device__ __forceinline__ void mac(float &inp, float &quad, const v4f (&x)[4], const f16v &ph) {
#pragma unroll
  for (int j = 0; j < 4; j++) {
    inp = (inp + x[j].x * ph[4 * j]) + x[j].y * ph[4 * j + 1];
    quad = (quad - x[j].x * ph[4 * j + 1]) + x[j].y * ph[4 * j];
    inp = (inp + x[j].z * ph[4 * j + 2]) + x[j].w * ph[4 * j + 3];
    quad = (quad - x[j].z * ph[4 * j + 3]) + x[j].w * ph[4 * j + 2];
  }
}

template <int SHAPE, int NW>
__global__ __launch_bounds__(64 * NW) void probe(const float *__restrict__ tabg, float *out, int iters) {
  __shared__ __align__(16) float smp[2 * 3 * 64 * 36];
  const int lane = threadIdx.x & 63;
  const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  for (int k = threadIdx.x; k < 2 * 3 * 64 * 36; k += 64 * NW) smp[k] = 0.001f * (k % 977);
  __syncthreads();
  float inp[3] = {0, 0, 0}, quad[3] = {0, 0, 0};
  const CONSTAS float *tab = (const CONSTAS float *)(tabg) + (size_t)(wv & 15) * 512 + (size_t)(blockIdx.x & 7) * 8192;
  const LDSAS float *rowp = (const LDSAS float *)smp + lane * 36;
  auto xrd = [&](v4f (&x)[4], int it, int r) {
#pragma unroll
    for (int j = 0; j < 4; j++) x[j] = *(const LDSAS v4f *)(rowp + (r * 64) * 36 + 4 * j + 16 * (it & 1) + (it & 2 ? 3 * 64 * 36 : 0));
  };
  if (SHAPE == 2 && (wv & 8)) __builtin_amdgcn_s_sleep(40);
  if (SHAPE == 0 || SHAPE == 2) {
    for (int it = 0; it < iters; it++) {   // one half: 3 row-items
      const f16v ph = *(const CONSTAS f16v *)(tab + ((it * 16) & 511));
      v4f x[3][4];
#pragma unroll
      for (int r = 0; r < 3; r++) xrd(x[r], it, r);
#pragma unroll
      for (int r = 0; r < 3; r++) mac(inp[r], quad[r], x[r], ph);
    }
  } else {
    f16v runA = *(const CONSTAS f16v *)tab, runB = runA;
    v4f xb[2][4];
    xrd(xb[0], 0, 0);
    for (int it = 0; it < iters; it += 2) {
#define ROWITEM(T, RUN, RUNNEXT, ITN, RN, LOADRUN)                                              \
      {                                                                                        \
        int tok = 0;                                                                           \
        asm volatile("; row-item ready" : "+s"(tok) : "v"(xb[(T) & 1][0]), "v"(xb[(T) & 1][1]), "v"(xb[(T) & 1][2]), "v"(xb[(T) & 1][3]), "s"(RUN)); \
        xrd(xb[((T) + 1) & 1], (ITN) + tok, RN);                                               \
        if (LOADRUN) RUNNEXT = *(const CONSTAS f16v *)(tab + tok + ((((ITN) + 0) * 16) & 511)); \
        __builtin_amdgcn_sched_barrier(0);                                                     \
        mac(inp[(T) % 3], quad[(T) % 3], xb[(T) & 1], RUN);                                    \
      }
      ROWITEM(0, runA, runB, it, 1, false)
      ROWITEM(1, runA, runB, it, 2, false)
      ROWITEM(2, runA, runB, it + 1, 0, true)
      ROWITEM(3, runB, runA, it + 1, 1, false)
      ROWITEM(4, runB, runA, it + 1, 2, false)
      ROWITEM(5, runB, runA, it + 2, 0, true)
    }
  }
  out[blockIdx.x * 64 * NW + threadIdx.x] = inp[0] + inp[1] + inp[2] + quad[0] + quad[1] + quad[2];
}

template <int SHAPE, int NW>
void run(int iters, const float *tab) {
  int blocks = 256;
  float *out; (void)hipMalloc(&out, (size_t)blocks * 64 * NW * 4);
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  probe<SHAPE, NW><<<blocks, 64 * NW>>>(tab, out, 8);
  (void)hipDeviceSynchronize();
  (void)hipEventRecord(e0);
  probe<SHAPE, NW><<<blocks, 64 * NW>>>(tab, out, iters);
  (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
  float ms; (void)hipEventElapsedTime(&ms, e0, e1);
  double ops = (double)blocks * 64 * NW * iters * 64 * 3;
  printf("shape=%d NW=%2d  %.3f ms  %.1f Tops/s  wave-instr/cycle/SIMD@2.4GHz=%.3f\n",
         SHAPE, NW, ms, ops / ms / 1e9, ops / 64 / (ms * 1e-3) / 1024 / 2.4e9);
  (void)hipFree(out);
}

int main() {
  float *tab; (void)hipMalloc(&tab, 64 * 8192 * 4);
  std::vector<float> h(64 * 8192);
  for (size_t i = 0; i < h.size(); i++) h[i] = 0.5f + 1e-4f * (i % 977);
  (void)hipMemcpy(tab, h.data(), h.size() * 4, hipMemcpyHostToDevice);
  const int it = 16384;
  run<0, 16>(it, tab); run<1, 16>(it, tab); run<2, 16>(it, tab);
  run<0, 8>(it, tab); run<1, 8>(it, tab);
  return 0;
}
