// fetch_calib.hip -- calibrates rocprofv3 FETCH_SIZE for 8-byte-per-lane coalesced
// streaming reads (the access width K4 uses) on a known byte count.
#include <hip/hip_runtime.h>
#include <stdio.h>
__global__ void read8(const float2 *p, size_t n, float *out) {
  float acc = 0;
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    float2 v = p[i]; acc += v.x + v.y;
  }
  if (acc == 123.456f) out[0] = acc;
}
__global__ void read16(const float4 *p, size_t n, float *out) {
  float acc = 0;
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    float4 v = p[i]; acc += v.x + v.y + v.z + v.w;
  }
  if (acc == 123.456f) out[0] = acc;
}
int main() {
  const size_t bytes = 2048ull << 20;  // 2 GiB, far beyond the 256 MiB Infinity Cache
  void *buf; float *out;
  hipMalloc(&buf, bytes); hipMalloc(&out, 4);
  hipMemset(buf, 0, bytes);
  hipDeviceSynchronize();
  read8<<<4096, 256>>>((const float2 *)buf, bytes / 8, out);
  hipDeviceSynchronize();
  read16<<<4096, 256>>>((const float4 *)buf, bytes / 16, out);
  hipDeviceSynchronize();
  printf("read %zu bytes with 8-byte and 16-byte loads\n", bytes);
  return 0;
}
