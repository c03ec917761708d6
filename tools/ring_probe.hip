// ring_probe.hip -- issue rate of the lag-group walk (1 phasor recurrence + NL multiply-accumulate
// pairs per sample, NL LDS reads per sample) as a function of NL, waves per SIMD and of where the
// samples come from.  Diagnostic only.
#include <hip/hip_runtime.h>
#include <stdio.h>
#pragma clang fp contract(off)

template <int NL, int MODE>   // MODE 0: NL ds_read_b64 per step; 1: registers (laundered, no LDS); 2: one b64 read per step
__global__ __launch_bounds__(128) void probe(float *out, int iters, float a, float b) {
  __shared__ float lds[2][16 * 200];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int pr = lane >> 2;
  for (int k = lane; k < 16 * 200; k += 64) lds[wv][k] = a * k + lane;
  __syncthreads();
  float c = 1.0f, s = 0.0f, cd = a, sd = b;
  float inp[NL], quad[NL];
  for (int l = 0; l < NL; l++) { inp[l] = 0; quad[l] = 0; }
  const float *row = &lds[wv][pr * 200];
  float2 xr[NL];
  for (int l = 0; l < NL; l++) xr[l] = make_float2(a + l, b - l);
  for (int it = 0; it < iters; it++) {
#pragma unroll
    for (int k = 0; k < 16; k++) {
      float2 x[NL];
#pragma unroll
      for (int l = 0; l < NL; l++) {
        if (MODE == 0) x[l] = *reinterpret_cast<const float2 *>(&row[2 * (k + 8 * l)]);
        else if (MODE == 2) x[l] = (l == 0) ? *reinterpret_cast<const float2 *>(&row[2 * k]) : xr[l];
        else x[l] = xr[l];
        if (MODE != 0) { asm volatile("" : "+v"(x[l].x), "+v"(x[l].y)); }
      }
#pragma unroll
      for (int l = 0; l < NL; l++) {
        inp[l] = (inp[l] + x[l].x * c) + x[l].y * s;
        quad[l] = (quad[l] - x[l].x * s) + x[l].y * c;
      }
      const float nc = c * cd - s * sd, ns = c * sd + s * cd;
      c = nc; s = ns;
    }
  }
  float r = 0;
  for (int l = 0; l < NL; l++) r += inp[l] + quad[l];
  out[blockIdx.x * 128 + threadIdx.x] = r;
}

template <int NL, int MODE>
void run(int waves_per_simd, int iters) {
  int blocks = 256 * 2 * waves_per_simd;   // 2 waves per block, 4 SIMDs per CU
  float *out; hipMalloc(&out, blocks * 128 * 4);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  probe<NL, MODE><<<blocks, 128>>>(out, 4, 0.999f, 0.01f);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  probe<NL, MODE><<<blocks, 128>>>(out, iters, 0.999f, 0.01f);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  double instr = (double)blocks * 2 * iters * 16 * (6 + 8 * NL);
  printf("NL=%d mode=%d waves/SIMD=%d  %.3f ms  VALU wave-instr/cycle/SIMD@2.4GHz=%.3f\n", NL, MODE, waves_per_simd, ms,
         instr / (ms * 1e-3) / 1024 / 2.4e9);
  hipFree(out);
}

int main() {
  for (int w : {2, 4, 8}) {
    run<1, 0>(w, 2048); run<2, 0>(w, 2048); run<6, 0>(w, 512); run<6, 1>(w, 512); run<6, 2>(w, 512);
  }
  return 0;
}
