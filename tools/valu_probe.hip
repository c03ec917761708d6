// valu_probe.hip -- what binary32 mul/add issue rate does gfx950 sustain for the
// K4 instruction mix (no FMA), as a function of waves per SIMD?  Diagnostic only.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
#pragma clang fp contract(off)

template <int T, bool LDS>
__global__ __launch_bounds__(256) void probe(float *out, int iters, float a, float b) {
  __shared__ float lds[4][64 * 34];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  float c[T], s[T], inp[T], quad[T], cd[T], sd[T];
  for (int j = 0; j < T; j++) { c[j] = 1.0f; s[j] = 0.0f; inp[j] = 0; quad[j] = 0; cd[j] = a + j * 1e-3f; sd[j] = b; }
  for (int k = 0; k < 32; k++) lds[wv][lane * 34 + k] = a * k + lane;
  __syncthreads();
  float xx = a, xy = b;
  for (int it = 0; it < iters; it++) {
#pragma unroll
    for (int k = 0; k < 16; k++) {
      if (LDS) { float2 x = *reinterpret_cast<const float2 *>(&lds[wv][lane * 34 + 2 * k]); xx = x.x; xy = x.y; }
#pragma unroll
      for (int j = 0; j < T; j++) {
        inp[j] = (inp[j] + xx * c[j]) + xy * s[j];
        quad[j] = (quad[j] - xx * s[j]) + xy * c[j];
        const float nc = c[j] * cd[j] - s[j] * sd[j];
        const float ns = c[j] * sd[j] + s[j] * cd[j];
        c[j] = nc; s[j] = ns;
      }
    }
  }
  float r = 0;
  for (int j = 0; j < T; j++) r += inp[j] + quad[j];
  out[blockIdx.x * 256 + threadIdx.x] = r;
}

template <int T, bool LDS>
void run(int wgs_per_cu, int iters) {
  int blocks = 256 * wgs_per_cu;
  float *out; hipMalloc(&out, blocks * 256 * 4);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  probe<T, LDS><<<blocks, 256>>>(out, 8, 0.999f, 0.01f);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  probe<T, LDS><<<blocks, 256>>>(out, iters, 0.999f, 0.01f);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  double ops = (double)blocks * 256 * iters * 16 * 14 * T;
  printf("T=%d lds=%d waves/SIMD=%d  %.3f ms  %.1f Tops/s (lane-ops)  wave-instr/cycle/SIMD@2.4GHz=%.3f\n", T, (int)LDS, wgs_per_cu,
         ms, ops / ms / 1e9, ops / 64 / (ms * 1e-3) / 1024 / 2.4e9);
  hipFree(out);
}

int main() {
  for (int w : {1, 2, 3, 4, 6, 8}) run<1, false>(w, 4096);
  for (int w : {1, 2, 4, 8}) run<4, false>(w, 1024);
  for (int w : {1, 2, 3, 4, 6, 8}) run<1, true>(w, 4096);
  for (int w : {1, 2, 4}) run<4, true>(w, 1024);
  return 0;
}
