// ring_probe.hip -- issue rate of the lag-group walk: T tone phasors per lane, NL lags, one
// ds_read_b128 per lag and two samples; VALU per sample = 6T + 8 T NL.  How does the rate depend
// on LDS bytes per arithmetic instruction and on waves per SIMD?  Diagnostic only.
#include <hip/hip_runtime.h>
#include <stdio.h>
#pragma clang fp contract(off)

template <int NL, int T, bool LDS>
__global__ __launch_bounds__(128) void probe(float *out, int iters, float a, float b) {
  __shared__ __align__(16) float lds[2][32 * 132];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int pr = lane / (4 / T);
  for (int k = lane; k < 32 * 132; k += 64) lds[wv][k] = a * k + lane;
  __syncthreads();
  float c[T], s[T], cd[T], sd[T];
  float inp[T][NL], quad[T][NL];
  for (int j = 0; j < T; j++) { c[j] = 1.0f; s[j] = 0.0f; cd[j] = a + 1e-3f * j; sd[j] = b;
    for (int l = 0; l < NL; l++) { inp[j][l] = 0; quad[j][l] = 0; } }
  const float *row = &lds[wv][pr * 132];
  float4 v[NL];
  for (int l = 0; l < NL; l++) v[l] = make_float4(a + l, b - l, a - l, b + l);
  for (int it = 0; it < iters; it++) {
#pragma unroll
    for (int k = 0; k < 16; k += 2) {
      if (LDS) {
#pragma unroll
        for (int l = 0; l < NL; l++) v[l] = *reinterpret_cast<const float4 *>(&row[2 * (k + 8 * l)]);
      }
#pragma unroll
      for (int half = 0; half < 2; half++) {
#pragma unroll
        for (int j = 0; j < T; j++) {
#pragma unroll
          for (int l = 0; l < NL; l++) {
            const float xx = half ? v[l].z : v[l].x, xy = half ? v[l].w : v[l].y;
            inp[j][l] = (inp[j][l] + xx * c[j]) + xy * s[j];
            quad[j][l] = (quad[j][l] - xx * s[j]) + xy * c[j];
          }
          const float nc = c[j] * cd[j] - s[j] * sd[j], ns = c[j] * sd[j] + s[j] * cd[j];
          c[j] = nc; s[j] = ns;
        }
      }
    }
  }
  float r = 0;
  for (int j = 0; j < T; j++) for (int l = 0; l < NL; l++) r += inp[j][l] + quad[j][l];
  out[blockIdx.x * 128 + threadIdx.x] = r;
}

template <int NL, int T, bool LDS>
void run(int waves_per_simd, int iters) {
  int blocks = 256 * 2 * waves_per_simd;
  float *out; hipMalloc(&out, blocks * 128 * 4);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  probe<NL, T, LDS><<<blocks, 128>>>(out, 4, 0.999f, 0.01f);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  probe<NL, T, LDS><<<blocks, 128>>>(out, iters, 0.999f, 0.01f);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  double instr = (double)blocks * 2 * iters * 16 * (6 * T + 8 * T * NL);
  printf("NL=%d T=%d lds=%d waves/SIMD=%d  %.3f ms  VALU wave-instr/cycle/SIMD@2.4GHz=%.3f\n", NL, T, (int)LDS,
         waves_per_simd, ms, instr / (ms * 1e-3) / 1024 / 2.4e9);
  hipFree(out);
}

int main() {
  for (int w : {2, 3, 4}) {
    run<6, 1, false>(w, 512); run<6, 1, true>(w, 512);
    run<6, 2, false>(w, 256); run<6, 2, true>(w, 256);
    run<3, 4, false>(w, 256); run<3, 4, true>(w, 256);
    run<6, 4, true>(w, 128);
  }
  return 0;
}
