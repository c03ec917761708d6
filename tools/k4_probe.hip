// k4_probe.hip -- which part of the K4 (T=1) structure costs what?  Diagnostic only.
// Replicates the flat kernel's loop: per 16-sample chunk {fence, 4 LDS writes, fence,
// 4 global loads for the next chunk, 16 x (ds_read_b64 + 14 VALU)} with pieces switchable.
#include <hip/hip_runtime.h>
#include <stdio.h>
#pragma clang fp contract(off)

__device__ __forceinline__ void wfence() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

template <bool LOADS, bool STAGE, bool SINCOS>
__global__ __launch_bounds__(256) void probe(const float2 *__restrict__ frames, int fl, float *out, float a, float b) {
  __shared__ float lds_all[4][16 * 34];
  const int lane = threadIdx.x & 63;
  const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  float *lds = lds_all[wv];
  const long long g0 = ((long long)blockIdx.x * 4 + wv) * 16;
  const int pr = lane >> 2, kk = lane & 15, segq = lane >> 4;
  const int h = (int)(g0 / 162), i0 = (int)(g0 % 162);
  const int frame = h % 256;
  const float2 *src[4];
  for (int t = 0; t < 4; t++) {
    int sym = (i0 + 4 * t + segq) % 162;
    src[t] = frames + ((long long)frame * fl + 368 + 256 * sym + kk);
  }
  float cd, sd;
  if (SINCOS) { double sn, cs; sincos(0.0245 * (double)(a + lane * 1e-3f), &sn, &cs); cd = (float)cs; sd = (float)sn; }
  else { cd = a; sd = b; }
  float2 stage[4];
  for (int t = 0; t < 4; t++) stage[t] = LOADS ? src[t][0] : make_float2(a, b);
  float c = 1, s = 0, inp = 0, quad = 0;
  if (!STAGE) { for (int k = 0; k < 34; k++) lds[pr * 34 + (k % 34)] = a * k; }
  for (int ch = 0; ch < 16; ch++) {
    if (STAGE) {
      wfence();
      for (int t = 0; t < 4; t++) *reinterpret_cast<float2 *>(&lds[(4 * t + segq) * 34 + 2 * kk]) = stage[t];
      wfence();
      if (ch < 15) for (int t = 0; t < 4; t++) stage[t] = LOADS ? src[t][16 * (ch + 1)] : make_float2(a + ch, b);
    }
#pragma unroll
    for (int k = 0; k < 16; k++) {
      const float2 x = *reinterpret_cast<const float2 *>(&lds[pr * 34 + 2 * k]);
      inp = (inp + x.x * c) + x.y * s;
      quad = (quad - x.x * s) + x.y * c;
      const float nc = c * cd - s * sd, ns = c * sd + s * cd;
      c = nc; s = ns;
    }
  }
  out[g0 * 4 + lane] = inp * inp + quad * quad;
}

template <bool LOADS, bool STAGE, bool SINCOS>
void run(const char *name, const float2 *frames, float *out, int waves) {
  int blocks = waves / 4;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  probe<LOADS, STAGE, SINCOS><<<blocks, 256>>>(frames, 45000, out, 0.999f, 0.01f);
  hipDeviceSynchronize();
  float best = 1e9;
  for (int r = 0; r < 5; r++) {
    hipEventRecord(e0);
    probe<LOADS, STAGE, SINCOS><<<blocks, 256>>>(frames, 45000, out, 0.999f, 0.01f);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); if (ms < best) best = ms;
  }
  double instr = (double)waves * 256 * 14;
  printf("%-28s waves=%7d  %8.1f us   %.3f wave-instr/cycle/SIMD@2.2GHz\n", name, waves, best * 1e3,
         instr / (best * 1e-3) / 1024 / 2.2e9);
}

int main() {
  float2 *frames; float *out;
  hipMalloc(&frames, (size_t)256 * 45000 * 8); hipMemset(frames, 0, (size_t)256 * 45000 * 8);
  hipMalloc(&out, (size_t)1 << 28);
  for (int waves : {12960, 12960 * 16}) {
    run<false, false, false>("compute+ldsread only", frames, out, waves);
    run<false, true, false>("+ staging (no loads)", frames, out, waves);
    run<true, true, false>("+ global loads", frames, out, waves);
    run<true, true, true>("+ sincos prologue", frames, out, waves);
  }
  return 0;
}
