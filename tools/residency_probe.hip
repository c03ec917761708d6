// residency_probe.hip -- how many workgroups / wavefronts does a CU of this box really hold at once?
// Diagnostic only.  Every wavefront spins for a fixed time and records its hardware placement (HW_ID, XCC_ID)
// and its start / end on the 100 MHz real-time counter; the host then counts, per CU and per SIMD, the
// largest number alive at the same time.  Build: hipcc --offload-arch=gfx950 -O3 tools/residency_probe.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <algorithm>
#include <map>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %d (%s) at line %d\n", (int)e_, hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

struct stamp { unsigned long long r0, r1; unsigned hw, xcc; };
extern __shared__ float dyn_lds[];

template <int NV, bool BAR>
__global__ void spin(stamp *st, int ticks, float *out) {
  const unsigned long long r0 = __builtin_amdgcn_s_memrealtime();
  float acc[NV];
#pragma unroll
  for (int i = 0; i < NV; i++) acc[i] = (float)(threadIdx.x + i);
  if (BAR) { dyn_lds[threadIdx.x] = acc[0]; __syncthreads(); acc[0] += dyn_lds[(threadIdx.x + 1) % blockDim.x]; }
  while (__builtin_amdgcn_s_memrealtime() - r0 < (unsigned long long)ticks) {
#pragma unroll
    for (int i = 0; i < NV; i++) acc[i] = acc[i] * 0.999f + 1e-3f;
  }
  float r = 0;
#pragma unroll
  for (int i = 0; i < NV; i++) r += acc[i];
  out[(size_t)blockIdx.x * blockDim.x + threadIdx.x] = r;
  if ((threadIdx.x & 63) == 0) {
    stamp s; s.r0 = r0; s.r1 = __builtin_amdgcn_s_memrealtime();
    s.hw = __builtin_amdgcn_s_getreg((31 << 11) | 4); s.xcc = __builtin_amdgcn_s_getreg((31 << 11) | 20);
    st[(size_t)blockIdx.x * (blockDim.x / 64) + (threadIdx.x >> 6)] = s;
  }
}

typedef void (*kern_t)(stamp *, int, float *);

static void census(const char *name, kern_t k, int threads, size_t lds, int wgs_per_cu) {
  const int blocks = 256 * wgs_per_cu, wpb = threads / 64;
  if (lds > 64 * 1024) CK(hipFuncSetAttribute((const void *)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  stamp *st; float *out;
  CK(hipMalloc(&st, (size_t)blocks * wpb * sizeof(stamp))); CK(hipMalloc(&out, (size_t)blocks * threads * 4));
  int occ = 0; CK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, (const void *)k, threads, lds));
  hipLaunchKernelGGL(k, dim3(blocks), dim3(threads), lds, 0, st, 20000 /* 200 us */, out);
  CK(hipDeviceSynchronize());
  std::vector<stamp> h((size_t)blocks * wpb);
  CK(hipMemcpy(h.data(), st, h.size() * sizeof(stamp), hipMemcpyDeviceToHost));
  std::map<unsigned, std::vector<std::pair<unsigned long long, int>>> cu, simd;
  unsigned long long t0 = ~0ull, t1 = 0;
  for (auto &s : h) {
    const unsigned c = ((s.xcc & 15) << 12) | (((s.hw >> 13) & 7) << 9) | (((s.hw >> 12) & 1) << 8) | (((s.hw >> 8) & 15) << 4);
    cu[c].push_back({s.r0, 1}); cu[c].push_back({s.r1, -1});
    simd[c | ((s.hw >> 4) & 3)].push_back({s.r0, 1}); simd[c | ((s.hw >> 4) & 3)].push_back({s.r1, -1});
    t0 = std::min(t0, s.r0); t1 = std::max(t1, s.r1);
  }
  auto peak = [](std::map<unsigned, std::vector<std::pair<unsigned long long, int>>> &m, int &mx, double &avg) {
    mx = 0; avg = 0;
    for (auto &kv : m) {
      auto &e = kv.second; std::sort(e.begin(), e.end());
      int cur = 0, p = 0;
      for (auto &q : e) { cur += q.second; p = std::max(p, cur); }
      mx = std::max(mx, p); avg += p;
    }
    avg /= std::max<size_t>(1, m.size());
  };
  int mc, ms; double ac, as; peak(cu, mc, ac); peak(simd, ms, as);
  printf("%-10s threads %4d lds %6zu  launched %d WG/CU  occ-api %2d | CUs %zu  waves/CU max %2d avg %5.2f (= %.2f WGs)  waves/SIMD max %d avg %.2f  span %.0f us\n",
         name, threads, lds, wgs_per_cu, occ, cu.size(), mc, ac, ac / wpb, ms, as, (t1 - t0) / 100.0);
  CK(hipFree(st)); CK(hipFree(out));
}

int main() {
  hipDeviceProp_t pr; CK(hipGetDeviceProperties(&pr, 0));
  printf("# %s CUs %d\n", pr.name, pr.multiProcessorCount);
  for (int threads : {64, 128, 256, 512, 1024}) {
    const int per = 2048 / threads;   // enough to fill 32 waves per CU
    census("v8", spin<8, false>, threads, 0, per);
    census("v8+bar", spin<8, true>, threads, 4096, per);
    census("v8+lds20k", spin<8, true>, threads, 20480, std::min(per, 8));
  }
  census("v64", spin<64, false>, 256, 0, 8);
  census("v64+bar", spin<64, true>, 256, 4096, 8);
  census("v100", spin<100, false>, 256, 0, 4);
  census("v100+bar", spin<100, true>, 256, 4096, 4);
  census("v100+bar", spin<100, true>, 256, 40960, 4);
  census("v100+bar", spin<100, true>, 256, 40960, 3);
  return 0;
}
