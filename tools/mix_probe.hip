// mix_probe.hip -- the tabled walk's instruction mix with its memory parts switched on and off:
// SL = phasors by s_load_dwordx16 per 8 steps (else loop-invariant SGPRs), DS = samples by
// ds_read_b128 (else loop-invariant VGPRs), R = rows per lane sharing a run.  Diagnostic only.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
#pragma clang fp contract(off)
typedef float f16v __attribute__((ext_vector_type(16)));
typedef float v4f __attribute__((ext_vector_type(4)));
#define CONSTAS __attribute__((address_space(4)))

template <int R, bool SL, bool DS, int NW, bool VG>
__global__ __launch_bounds__(64 * NW) void probe(const float *__restrict__ tabg, float *out, int iters) {
  __shared__ __align__(16) float smp[3 * 64 * 36 + 64];
  const int lane = threadIdx.x & 63;
  const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  for (int k = threadIdx.x; k < 3 * 64 * 36; k += 64 * NW) smp[k] = 0.001f * k;
  __syncthreads();
  float inp[R], quad[R];
#pragma unroll
  for (int r = 0; r < R; r++) { inp[r] = 0; quad[r] = 0; }
  const CONSTAS float *tab = (const CONSTAS float *)(tabg) + (size_t)(wv & 15) * 512 + (size_t)(blockIdx.x & 7) * 8192;
  f16v ph = *(const CONSTAS f16v *)tab;
  v4f xv[R][4];
#pragma unroll
  for (int r = 0; r < R; r++)
#pragma unroll
    for (int j = 0; j < 4; j++) xv[r][j] = *(const v4f *)&smp[(r * 64 + lane) * 36 + 4 * j];
  for (int it = 0; it < iters; it++) {   // one item = 8 steps of one hypothesis for R rows
    if (SL) ph = *(const CONSTAS f16v *)(tab + ((it * 16) & 511));
    if (DS) {
#pragma unroll
      for (int r = 0; r < R; r++)
#pragma unroll
        for (int j = 0; j < 4; j++) xv[r][j] = *(const v4f *)&smp[(r * 64 + lane) * 36 + 4 * j + 16 * (it & 1)];
    }
    float pv[16];
#pragma unroll
    for (int e = 0; e < 16; e++) { pv[e] = ph[e]; if (VG) asm volatile("v_mov_b32 %0, %1" : "=v"(pv[e]) : "s"(ph[e])); }
#pragma unroll
    for (int r = 0; r < R; r++)
#pragma unroll
      for (int j = 0; j < 4; j++) {
        const v4f x = xv[r][j];
        if (VG) {
          inp[r] = (inp[r] + x.x * pv[4 * j]) + x.y * pv[4 * j + 1];
          quad[r] = (quad[r] - x.x * pv[4 * j + 1]) + x.y * pv[4 * j];
          inp[r] = (inp[r] + x.z * pv[4 * j + 2]) + x.w * pv[4 * j + 3];
          quad[r] = (quad[r] - x.z * pv[4 * j + 3]) + x.w * pv[4 * j + 2];
          continue;
        }
        inp[r] = (inp[r] + x.x * ph[4 * j]) + x.y * ph[4 * j + 1];
        quad[r] = (quad[r] - x.x * ph[4 * j + 1]) + x.y * ph[4 * j];
        inp[r] = (inp[r] + x.z * ph[4 * j + 2]) + x.w * ph[4 * j + 3];
        quad[r] = (quad[r] - x.z * ph[4 * j + 3]) + x.w * ph[4 * j + 2];
      }
    if (!DS) xv[0][0].x += 1.0f;
  }
  float r0 = 0;
#pragma unroll
  for (int r = 0; r < R; r++) r0 += inp[r] + quad[r];
  out[blockIdx.x * 64 * NW + threadIdx.x] = r0;
}

template <int R, bool SL, bool DS, int NW, bool VG>
void run(int iters, const float *tab) {
  int blocks = 256;
  float *out; (void)hipMalloc(&out, (size_t)blocks * 64 * NW * 4);
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  probe<R, SL, DS, NW, VG><<<blocks, 64 * NW>>>(tab, out, 8);
  (void)hipDeviceSynchronize();
  (void)hipEventRecord(e0);
  probe<R, SL, DS, NW, VG><<<blocks, 64 * NW>>>(tab, out, iters);
  (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
  float ms; (void)hipEventElapsedTime(&ms, e0, e1);
  double ops = (double)blocks * 64 * NW * iters * 64 * R;
  printf("vgpr_phasors=%d rows=%d s_load=%d ds_read=%d NW=%2d  %.3f ms  %.1f Tops/s  wave-instr/cycle/SIMD@2.4GHz=%.3f\n",
         (int)VG, R, (int)SL, (int)DS, NW, ms, ops / ms / 1e9, ops / 64 / (ms * 1e-3) / 1024 / 2.4e9);
  (void)hipFree(out);
}

int main() {
  float *tab; (void)hipMalloc(&tab, 64 * 8192 * 4);
  std::vector<float> h(64 * 8192);
  for (size_t i = 0; i < h.size(); i++) h[i] = 0.5f + 1e-4f * (i % 977);
  (void)hipMemcpy(tab, h.data(), h.size() * 4, hipMemcpyHostToDevice);
  const int it = 16384;
  run<3, true, false, 16, false>(it, tab); run<3, true, false, 16, true>(it, tab);
  run<3, true, true, 16, false>(it, tab); run<3, true, true, 16, true>(it, tab);
  run<3, true, false, 8, false>(it, tab); run<3, true, false, 8, true>(it, tab);
  run<6, true, false, 16, false>(it / 2, tab); run<6, true, false, 16, true>(it / 2, tab);
  return 0;
}
