// smem_probe.hip -- scalar-load (s_load_dwordx16) rate per CU on gfx950: streaming through a table
// that misses the scalar cache vs re-reading lines that hit.  Diagnostic only.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f16v __attribute__((ext_vector_type(16)));
#define CONSTAS __attribute__((address_space(4)))

// every wave reads `n` consecutive 64-B lines starting at its own offset (stride_w lines apart),
// wrapping inside `span` lines; `dep`: wait for each load before the next
template <int NW, bool DEP>
__global__ __launch_bounds__(64 * NW) void probe(const float *tab, float *out, int n, int span, int stride_w, int share) {
  const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int w = share ? (wv / share) : wv;
  const CONSTAS float *t = (const CONSTAS float *)tab + (size_t)blockIdx.x * span * 16;
  float acc = 0.0f;
  int line = (w * stride_w) % span;
  for (int i = 0; i < n; i += 4) {
    f16v a = *(const CONSTAS f16v *)(t + (size_t)line * 16); line = (line + 1) % span;
    if (DEP) acc += a[0];
    f16v b = *(const CONSTAS f16v *)(t + (size_t)line * 16); line = (line + 1) % span;
    if (DEP) acc += b[1];
    f16v c = *(const CONSTAS f16v *)(t + (size_t)line * 16); line = (line + 1) % span;
    if (DEP) acc += c[2];
    f16v d = *(const CONSTAS f16v *)(t + (size_t)line * 16); line = (line + 1) % span;
    acc += a[3] + b[4] + c[5] + d[6];
  }
  if ((threadIdx.x & 63) == 0) out[blockIdx.x * NW + wv] = acc;
}

template <int NW, bool DEP>
void run(const float *tab, float *out, int n, int span, int stride_w, int share, const char *what) {
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  probe<NW, DEP><<<256, 64 * NW>>>(tab, out, 64, span, stride_w, share);
  (void)hipDeviceSynchronize();
  (void)hipEventRecord(e0);
  probe<NW, DEP><<<256, 64 * NW>>>(tab, out, n, span, stride_w, share);
  (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
  float ms; (void)hipEventElapsedTime(&ms, e0, e1);
  printf("%-34s NW=%2d dep=%d span=%5d lines: %.1f ns per load per wave, %.2f loads/us per CU, %.2f GB/s per CU\n", what, NW,
         (int)DEP, span, ms * 1e6 / n, (double)n * NW / (ms * 1e3), (double)n * NW * 64 / (ms * 1e6));
}

int main() {
  const int SPAN = 8192;   // lines per CU region = 512 KB
  float *tab, *out;
  (void)hipMalloc(&tab, (size_t)256 * SPAN * 64); (void)hipMemset(tab, 0, (size_t)256 * SPAN * 64);
  (void)hipMalloc(&out, 256 * 16 * 4);
  const int n = 8192;
  run<16, true>(tab, out, n, 8, 0, 0, "hit, all waves same lines");
  run<16, false>(tab, out, n, 8, 0, 0, "hit, all waves same lines");
  run<1, true>(tab, out, n, 8, 0, 0, "hit, one wave");
  run<1, true>(tab, out, n, SPAN, 0, 0, "stream (miss), one wave");
  run<1, false>(tab, out, n, SPAN, 0, 0, "stream (miss), one wave");
  run<4, true>(tab, out, n, SPAN, 512, 0, "stream, 4 waves own streams");
  run<16, true>(tab, out, n, SPAN, 512, 0, "stream, 16 waves own streams");
  run<16, false>(tab, out, n, SPAN, 512, 0, "stream, 16 waves own streams");
  run<16, true>(tab, out, n, SPAN, 512, 4, "stream, 4 streams x 4 waves share");
  run<16, false>(tab, out, n, SPAN, 512, 4, "stream, 4 streams x 4 waves share");
  run<16, true>(tab, out, n, 256, 16, 0, "16 KB working set, 16 waves");
  run<16, true>(tab, out, n, 128, 8, 0, "8 KB working set, 16 waves");
  run<16, true>(tab, out, n, 512, 32, 0, "32 KB working set, 16 waves");
  return 0;
}
