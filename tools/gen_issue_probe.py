#!/usr/bin/env python3
"""Generates tools/issue_probe.hip: VALU issue-rate probes with EXACT instruction order (inline asm).

Question (round-3 review, item 1): what keeps the K4 correlation loops at 0.33-0.55 of the
1-instruction-per-2-cycles VALU rate?  The compiler's schedule of k4_fpack / k4_ring / k4_lag0 is a chain
in which nearly every instruction depends on the one before it (v_mul t; v_add acc, t, acc; v_mul t; ...).
Each body below is 64 VALU instructions (32 v_mul_f32 + 32 v_add_f32: the multiply-add mix of two sample steps
x four hypotheses, cc:206-207) in a chosen ORDER, optionally with the loop's LDS reads, run at a forced
number of wavefronts per SIMD.  Reported: cycles per VALU instruction per SIMD (s_memtime), the in-kernel
clock (s_memtime / s_memrealtime), and the wall-clock rate.
"""
import sys

ACC = list(range(10, 26))      # 16 accumulators: inp/quad x 8 (v10..v25)
TMP = list(range(30, 62))      # temporaries v30..v61
XS = [2, 3, 4, 5]              # sample values (xx, xy of two steps)
PH = list(range(64, 80))       # phasor values: 4 hypotheses x (c, s) x 2 steps = 16 (v64..v79)


def body_serial():
    """The compiler's order: mul t; add acc,t; mul t; add acc,t -- every instruction depends on its predecessor,
    the same temporary is reused (as in the -O3 listing of k4_fpack)."""
    out = []
    for h in range(4):
        for half in range(2):
            c, s = PH[4 * h + 2 * half], PH[4 * h + 2 * half + 1]
            xx, xy = XS[2 * half], XS[2 * half + 1]
            ai, aq = ACC[2 * h], ACC[2 * h + 1]
            t = TMP[0]
            out += ["v_mul_f32 v%d, v%d, v%d" % (t, xx, c), "v_add_f32 v%d, v%d, v%d" % (ai, t, ai),
                    "v_mul_f32 v%d, v%d, v%d" % (t, xy, s), "v_add_f32 v%d, v%d, v%d" % (ai, t, ai),
                    "v_mul_f32 v%d, v%d, v%d" % (t, xx, s), "v_sub_f32 v%d, v%d, v%d" % (aq, aq, t),
                    "v_mul_f32 v%d, v%d, v%d" % (t, xy, c), "v_add_f32 v%d, v%d, v%d" % (aq, t, aq)]
    return out


def body_interleaved(width):
    """`width` independent accumulator chains advance in lock step: all their multiplies, then all their adds.
    width = 8: distance 8 between dependent instructions."""
    out = []
    chains = []   # (acc, xx-ish, phasor, op)
    for half in range(2):
        for h in range(4):
            c, s = PH[4 * h + 2 * half], PH[4 * h + 2 * half + 1]
            xx, xy = XS[2 * half], XS[2 * half + 1]
            ai, aq = ACC[2 * h], ACC[2 * h + 1]
            chains.append((half, [(ai, xx, c, "add"), (ai, xy, s, "add")]))
            chains.append((half, [(aq, xx, s, "sub"), (aq, xy, c, "add")]))
    for half in range(2):
        ch = [c for hf, c in chains if hf == half]       # 8 chains of 2 (mul, add) pairs
        for g in range(0, 8, width):
            grp = ch[g:g + width]
            for step in range(2):
                for i, c in enumerate(grp):
                    acc, x, p, op = c[step]
                    out.append("v_mul_f32 v%d, v%d, v%d" % (TMP[i], x, p))
                for i, c in enumerate(grp):
                    acc, x, p, op = c[step]
                    if op == "sub":
                        out.append("v_sub_f32 v%d, v%d, v%d" % (acc, acc, TMP[i]))
                    else:
                        out.append("v_add_f32 v%d, v%d, v%d" % (acc, TMP[i], acc))
    return out


def body_indep():
    out = []
    for i in range(64):
        r = 10 + (i % 16)
        out.append(("v_mul_f32 v%d, v%d, v2" if i % 2 == 0 else "v_add_f32 v%d, v%d, v3") % (r, r))
    return out


def body_dep1():
    return ["v_add_f32 v10, v10, v2" if i % 2 else "v_mul_f32 v10, v10, v3" for i in range(64)]


def with_lds(instrs, nreads, sgpr_ph=False, wait="late"):
    """insert `nreads` ds_read_b128 (one own-row sample read + broadcast phasor reads) spread over the body;
    results go to v80.. (not consumed by the arithmetic: issue cost only) and are waited for at the end ("late")
    or right before the next group as the compiler does ("tight")."""
    out = []
    n = len(instrs)
    at = {int(i * n / nreads): i for i in range(nreads)} if nreads else {}
    for i, ins in enumerate(instrs):
        if i in at:
            k = at[i]
            addr = "v6" if k == 0 else "v7"      # v6: per-lane row address, v7: broadcast address
            out.append("ds_read_b128 v[%d:%d], %s offset:%d" % (80 + 4 * k, 83 + 4 * k, addr, 16 * k))
            if wait == "tight" and k > 0:
                out.append("s_waitcnt lgkmcnt(1)")
        out.append(ins)
    if nreads:
        out.append("s_waitcnt lgkmcnt(0)")
    return out


VARIANTS = [
    ("indep16", body_indep(), 0, "late"),
    ("dep1", body_dep1(), 0, "late"),
    ("serial", body_serial(), 0, "late"),
    ("inter2", body_interleaved(2), 0, "late"),
    ("inter4", body_interleaved(4), 0, "late"),
    ("inter8", body_interleaved(8), 0, "late"),
    ("serial_lds5", body_serial(), 5, "tight"),
    ("inter8_lds5", body_interleaved(8), 5, "late"),
    ("inter8_lds1", body_interleaved(8), 1, "late"),
    ("serial_lds1", body_serial(), 1, "late"),
]

HEAD = r'''// GENERATED by tools/gen_issue_probe.py -- do not edit.  Diagnostic only (nothing here is on the product path).
// hipcc --offload-arch=gfx950 -O3 tools/issue_probe.hip -o /tmp/issue_probe && /tmp/issue_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <algorithm>
#include <vector>

#define CLOB "v2","v3","v4","v5","v6","v7","v10","v11","v12","v13","v14","v15","v16","v17","v18","v19","v20","v21","v22","v23","v24","v25", \
  "v30","v31","v32","v33","v34","v35","v36","v37","v38","v39","v40","v41","v42","v43","v44","v45", \
  "v64","v65","v66","v67","v68","v69","v70","v71","v72","v73","v74","v75","v76","v77","v78","v79", \
  "v80","v81","v82","v83","v84","v85","v86","v87","v88","v89","v90","v91","v92","v93","v94","v95","v96","v97","v98","v99"

struct stamp { unsigned long long c0, c1, r0, r1; unsigned hw, xcc; };
extern __shared__ float dyn_lds[];
typedef void (*kern_t)(float *, stamp *, int, float, float);
struct variant { const char *name; kern_t k; int nvalu; int nlds; };
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %d at line %d\\n", (int)e_, __LINE__); exit(1); } } while (0)
'''

KERNEL = r'''
__global__ __launch_bounds__(256) void probe_%(name)s(float *out, stamp *st, int iters, float a, float b) {
  const int lane = threadIdx.x & 63;
  dyn_lds[threadIdx.x * 36 %% 8192] = a;   // touch LDS so the allocation is real
  __syncthreads();
  // seed registers (values stay finite: multipliers ~1, addends ~0)
  asm volatile(
    "v_mov_b32 v2, %%0\n v_mov_b32 v3, %%1\n v_mov_b32 v4, %%0\n v_mov_b32 v5, %%1\n"
    "v_mov_b32 v6, %%2\n v_mov_b32 v7, 0\n"
    "v_mov_b32 v10, 0\n v_mov_b32 v11, 0\n v_mov_b32 v12, 0\n v_mov_b32 v13, 0\n v_mov_b32 v14, 0\n v_mov_b32 v15, 0\n v_mov_b32 v16, 0\n v_mov_b32 v17, 0\n"
    "v_mov_b32 v18, 0\n v_mov_b32 v19, 0\n v_mov_b32 v20, 0\n v_mov_b32 v21, 0\n v_mov_b32 v22, 0\n v_mov_b32 v23, 0\n v_mov_b32 v24, 0\n v_mov_b32 v25, 0\n"
    "v_mov_b32 v64, %%1\n v_mov_b32 v65, %%1\n v_mov_b32 v66, %%1\n v_mov_b32 v67, %%1\n v_mov_b32 v68, %%1\n v_mov_b32 v69, %%1\n v_mov_b32 v70, %%1\n v_mov_b32 v71, %%1\n"
    "v_mov_b32 v72, %%1\n v_mov_b32 v73, %%1\n v_mov_b32 v74, %%1\n v_mov_b32 v75, %%1\n v_mov_b32 v76, %%1\n v_mov_b32 v77, %%1\n v_mov_b32 v78, %%1\n v_mov_b32 v79, %%1\n"
    :: "v"(a), "v"(b), "v"(lane * 144) : CLOB);
  const unsigned long long c0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
  for (int it = 0; it < iters; it++) {
    asm volatile(
%(body)s
      ::: CLOB);
  }
  const unsigned long long c1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
  float r;
  asm volatile("v_add_f32 %%0, v10, v11\n v_add_f32 %%0, %%0, v12\n v_add_f32 %%0, %%0, v13\n v_add_f32 %%0, %%0, v14\n v_add_f32 %%0, %%0, v15\n"
               "v_add_f32 %%0, %%0, v16\n v_add_f32 %%0, %%0, v17\n v_add_f32 %%0, %%0, v18\n v_add_f32 %%0, %%0, v19\n v_add_f32 %%0, %%0, v20\n"
               "v_add_f32 %%0, %%0, v21\n v_add_f32 %%0, %%0, v22\n v_add_f32 %%0, %%0, v23\n v_add_f32 %%0, %%0, v24\n v_add_f32 %%0, %%0, v25\n"
               : "=v"(r) :: CLOB);
  out[blockIdx.x * 256 + threadIdx.x] = r;
  if (lane == 0) {
    stamp s; s.c0 = c0; s.c1 = c1; s.r0 = r0; s.r1 = r1;
    s.hw = __builtin_amdgcn_s_getreg((31 << 11) | 4); s.xcc = __builtin_amdgcn_s_getreg((31 << 11) | 20);
    st[blockIdx.x * 4 + (threadIdx.x >> 6)] = s;
  }
}
'''

MAIN = r'''
static void run(const variant &v, int wps, int iters) {
  // wps workgroups of 4 wavefronts per CU, forced: the dynamic LDS allocation lets exactly wps fit
  const int blocks = 256 * wps;
  size_t lds = (160 * 1024 / wps) & ~1023;
  CK(hipFuncSetAttribute((const void *)v.k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));   // (every time: the attribute sticks to the function, and the LDS a launch reserves follows it)
  float *out; stamp *st;
  CK(hipMalloc(&out, (size_t)blocks * 256 * 4)); CK(hipMalloc(&st, (size_t)blocks * 4 * sizeof(stamp)));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  hipLaunchKernelGGL(v.k, dim3(blocks), dim3(256), lds, 0, out, st, 64, 0.999f, 1e-3f);
  CK(hipDeviceSynchronize());
  CK(hipEventRecord(e0));
  hipLaunchKernelGGL(v.k, dim3(blocks), dim3(256), lds, 0, out, st, iters, 0.999f, 1e-3f);
  CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
  if (hipGetLastError() != hipSuccess) { printf("%s: launch failed\n", v.name); exit(1); }
  float ms; CK(hipEventElapsedTime(&ms, e0, e1));
  std::vector<stamp> h((size_t)blocks * 4);
  CK(hipMemcpy(h.data(), st, h.size() * sizeof(stamp), hipMemcpyDeviceToHost));
  std::vector<double> cyc, clk;
  unsigned long long rmin = ~0ull, rmax = 0;
  for (auto &s : h) {
    cyc.push_back((double)(s.c1 - s.c0));
    clk.push_back((double)(s.c1 - s.c0) / (double)(s.r1 - s.r0) * 100.0);   // MHz (s_memrealtime: 100 MHz)
    rmin = std::min(rmin, s.r0); rmax = std::max(rmax, s.r1);
  }
  // residency: the largest number of wavefronts alive at once on one SIMD (hw id bits: simd 5:4, cu 11:8, sh 12, se 15:13; xcc 3:0)
  int maxres = 0, nsimd = 0; double avgres = 0;
  {
    std::vector<std::vector<std::pair<unsigned long long, int>>> ev(1 << 16);
    for (auto &s : h) {
      const unsigned key = ((s.xcc & 15) << 12) | (((s.hw >> 13) & 7) << 9) | (((s.hw >> 12) & 1) << 8) | (((s.hw >> 8) & 15) << 4) | ((s.hw >> 4) & 3);
      ev[key].push_back({s.r0, +1}); ev[key].push_back({s.r1, -1});
    }
    for (auto &e : ev) {
      if (e.empty()) continue;
      std::sort(e.begin(), e.end());
      int cur = 0, mx = 0;
      for (auto &q : e) { cur += q.second; mx = std::max(mx, cur); }
      maxres = std::max(maxres, mx); avgres += mx; nsimd++;
    }
    avgres /= std::max(1, nsimd);
  }
  std::sort(cyc.begin(), cyc.end()); std::sort(clk.begin(), clk.end());
  const double med = cyc[cyc.size() / 2], mx = cyc.back();
  const double n = (double)iters * v.nvalu;
  // cycles per VALU instruction per SIMD = wave lifetime / (instructions x wavefronts per SIMD)
  int occ = 0;
  CK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, (const void *)v.k, 256, lds));
  printf("%-14s occ-api %d  SIMDs seen %d  resident/SIMD max %d avg %.2f | waves/SIMD=%d  cyc/instr/SIMD med %.2f max %.2f   per wave %.2f   clock %.0f MHz   wall %.3f ms (span %.3f)  instr/cyc/SIMD@wall,2.4GHz %.3f\n",
         v.name, occ, nsimd, maxres, avgres, wps, med / n / wps, mx / n / wps, med / n, clk[clk.size() / 2], ms, (rmax - rmin) / 1e5,
         n * wps / (ms * 1e-3 * 2.4e9));
  CK(hipFree(out)); CK(hipFree(st));
}

int main(int argc, char **argv) {
  const int iters = argc > 1 ? atoi(argv[1]) : 20000;
  hipDeviceProp_t pr; CK(hipGetDeviceProperties(&pr, 0));
  printf("# %s  CUs %d  maxThreads/CU %d  LDS/CU %zu  LDS/block %zu  regs/block %d  clock %d kHz\\n", pr.name, pr.multiProcessorCount,
         pr.maxThreadsPerMultiProcessor, pr.maxSharedMemoryPerMultiProcessor, pr.sharedMemPerBlock, pr.regsPerBlock, pr.clockRate);
  for (const variant &v : variants)
    for (int w : {1, 2, 3, 4, 5, 6, 8}) run(v, w, iters / w);
  return 0;
}
'''


def main():
    out = [HEAD]
    names = []
    for name, instrs, nlds, wait in VARIANTS:
        b = with_lds(instrs, nlds, wait=wait)
        body = "\n".join('      "%s\\n"' % i for i in b)
        out.append(KERNEL % dict(name=name, body=body))
        names.append((name, 64, nlds))
    out.append("static const variant variants[] = {\n" +
               "\n".join('  {"%s", probe_%s, %d, %d},' % (n, n, nv, nl) for n, nv, nl in names) + "\n};\n")
    out.append(MAIN)
    path = sys.argv[1] if len(sys.argv) > 1 else "tools/issue_probe.hip"
    open(path, "w").write("".join(out))


if __name__ == "__main__":
    main()
