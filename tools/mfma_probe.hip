// mfma_probe.hip -- can the matrix pipe serve as an exact binary32 multiplier beside
// the VALU?  v_mfma_f32_4x4x1_16b_f32 with C = 0 computes, per 4-lane block,
// D[i][j] = fma(A[i], B[j], 0) = fl(A[i]*B[j]).  Checks (1) the lane layout and bit
// equality with v_mul_f32, (2) the cost of 4 such MFMAs + 22 VALU adds per step
// against 38 VALU ops per step (the k4_group<4> mix).  Diagnostic only.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <string.h>
#include <vector>
#pragma clang fp contract(off)
typedef float f4 __attribute__((ext_vector_type(4)));

__global__ void layout(const float *a, const float *b, float *d) {
  const int l = threadIdx.x;
  f4 z = {0, 0, 0, 0};
  f4 r = __builtin_amdgcn_mfma_f32_4x4x1f32(a[l], b[l], z, 0, 0, 0);
  for (int i = 0; i < 4; i++) d[l * 4 + i] = r[i];
}

template <int MODE>
__global__ __launch_bounds__(256) void rate(float *out, int iters, float a, float b) {
  const int l = threadIdx.x;
  float c = 1.0f + l * 1e-3f, s = 0.1f, cd = a, sd = b;
  float inp[4] = {0, 0, 0, 0}, quad[4] = {0, 0, 0, 0};
  float xx = a * 0.5f + l, xy = b + l;
  f4 z = {0, 0, 0, 0};
  for (int it = 0; it < iters; it++) {
#pragma unroll 4
    for (int k = 0; k < 16; k++) {
      if (MODE == 0) {  // VALU only: 4 lags x (4 mul + 4 add) + 6 phasor
#pragma unroll
        for (int q = 0; q < 4; q++) {
          inp[q] = (inp[q] + (xx + q) * c) + xy * s;
          quad[q] = (quad[q] - (xx + q) * s) + xy * c;
        }
      } else {          // products on the matrix pipe, adds on the VALU
        f4 p1 = __builtin_amdgcn_mfma_f32_4x4x1f32(xx, c, z, 0, 0, 0);
        f4 p2 = __builtin_amdgcn_mfma_f32_4x4x1f32(xy, s, z, 0, 0, 0);
        f4 p3 = __builtin_amdgcn_mfma_f32_4x4x1f32(xx, s, z, 0, 0, 0);
        f4 p4 = __builtin_amdgcn_mfma_f32_4x4x1f32(xy, c, z, 0, 0, 0);
#pragma unroll
        for (int q = 0; q < 4; q++) {
          inp[q] = (inp[q] + p1[q]) + p2[q];
          quad[q] = (quad[q] - p3[q]) + p4[q];
        }
      }
      const float nc = c * cd - s * sd, ns = c * sd + s * cd;
      c = nc; s = ns;
      xx += 1e-3f;
    }
  }
  float r = 0;
  for (int q = 0; q < 4; q++) r += inp[q] + quad[q];
  out[blockIdx.x * 256 + l] = r;
}

template <int MODE>
void run(int wgs_per_cu, int iters) {
  int blocks = 256 * wgs_per_cu;
  float *out; hipMalloc(&out, blocks * 256 * 4);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  rate<MODE><<<blocks, 256>>>(out, 8, 0.999f, 0.01f); hipDeviceSynchronize();
  hipEventRecord(e0);
  rate<MODE><<<blocks, 256>>>(out, iters, 0.999f, 0.01f);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  double steps = (double)blocks * 4 * iters * 16;  // wave-steps
  printf("mode=%d waves/SIMD=%d  %.3f ms  %.1f cycles per wave-step per SIMD @2.4GHz\n", MODE, wgs_per_cu, ms,
         ms * 1e-3 * 2.4e9 * 1024 / steps);
  hipFree(out);
}

int main() {
  std::vector<float> a(64), b(64), d(256);
  for (int i = 0; i < 64; i++) { a[i] = 1.0f + 0.37f * i + 1e-5f * i * i; b[i] = -2.0f + 0.113f * i; }
  float *da, *db, *dd; hipMalloc(&da, 256); hipMalloc(&db, 256); hipMalloc(&dd, 1024);
  hipMemcpy(da, a.data(), 256, hipMemcpyHostToDevice); hipMemcpy(db, b.data(), 256, hipMemcpyHostToDevice);
  layout<<<1, 64>>>(da, db, dd); hipMemcpy(d.data(), dd, 1024, hipMemcpyDeviceToHost);
  int ok = 0, tried = 0;
  // hypothesis: lane l = (block l/4, j = l%4); D reg i = A[block][i] * B[block][j]
  for (int l = 0; l < 64; l++) for (int i = 0; i < 4; i++) {
    float want = a[(l / 4) * 4 + i] * b[l];
    tried++; ok += memcmp(&want, &d[l * 4 + i], 4) == 0;
  }
  printf("layout D[lane][i] == A[4*(lane/4)+i] * B[lane], bitwise: %d / %d\n", ok, tried);
  for (int w : {1, 2, 4}) { run<0>(w, 512); run<1>(w, 512); }
  return 0;
}
