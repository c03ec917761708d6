// fft_lane_emul.cc -- CPU emulation of K1's 64-lane 512-point transform (gr-uwspr_amd/csrc/fft512_lane.h).
// Runs the three register passes and the two exchanges through the swizzled LDS image exactly as the
// kernel does, lane by lane, and compares every bin BIT FOR BIT with the textbook iterative radix-2
// decimation-in-time FFT on bit-reversed input using the same binary32 twiddle table (the arithmetic
// contract stated in the header).  Also checks that the exchange indices are a permutation free of the
// bank conflicts the header claims, and that the narrow pass C reproduces slots 0 and 7.
// Test infrastructure: built and run by tests/test_host_blocks.py with g++ -ffp-contract=off.
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <set>
#include <vector>

#include "fft512_lane.h"

using namespace uwspr;

static int rev9(int x) { int r = 0; for (int b = 0; b < 9; b++) r |= ((x >> b) & 1) << (8 - b); return r; }

int main() {
  std::vector<cpx> tw(256);
  for (int k = 0; k < 256; k++) {
    const double a = 2.0 * M_PI * (double)k / 512.0;
    tw[k] = cpx{(float)cos(a), (float)(-sin(a))};
  }
  tw[0] = cpx{1.0f, 0.0f}; tw[128] = cpx{0.0f, -1.0f};
  uint32_t seed = 12345u;
  auto rnd = [&]() { seed = seed * 1664525u + 1013904223u; return (float)((int)(seed >> 8) - (1 << 23)) / (float)(1 << 20); };
  int bad = 0;
  for (int trial = 0; trial < 4; trial++) {
    std::vector<cpx> x(512);
    for (auto &v : x) { v.r = rnd(); v.i = rnd(); }
    // ---- reference: iterative radix-2 DIT on bit-reversed input ---------------------------------
    std::vector<cpx> ref(512);
    for (int p = 0; p < 512; p++) ref[p] = x[rev9(p)];
    for (int h = 1; h < 512; h <<= 1)
      for (int b0 = 0; b0 < 512; b0 += 2 * h)
        for (int j = 0; j < h; j++) {
          const cpx w = tw[j * (256 / h)];
          cpx &u = ref[b0 + j], &v = ref[b0 + j + h];
          const float tr = w.r * v.r - w.i * v.i, ti = w.r * v.i + w.i * v.r;
          const float ur = u.r, ui = u.i;
          u.r = ur + tr; u.i = ui + ti; v.r = ur - tr; v.i = ui - ti;
        }
    // ---- 64 emulated lanes --------------------------------------------------------------------------
    for (int narrow = 0; narrow < 2; narrow++) {
      std::vector<cpx> lds(XCHG_LEN);
      cpx y[64][8];
      for (int L = 0; L < 64; L++) {
        for (int r = 0; r < 8; r++) y[L][r] = x[in_sample(L, r)];
        pass_a(y[L], tw[64], tw[192]);
      }
      for (int L = 0; L < 64; L++) for (int r = 0; r < 8; r++) lds[xidx(posA(L, r))] = y[L][r];
      for (int L = 0; L < 64; L++) {
        for (int e = 0; e < 8; e++) y[L][e] = lds[xidx(posB(L, e))];
        const pass_tw t = load_pass_tw(tw.data(), L & 7, 8);
        pass_bc(y[L], t);
      }
      for (int L = 0; L < 64; L++) for (int e = 0; e < 8; e++) lds[xidx(posB(L, e))] = y[L][e];
      for (int L = 0; L < 64; L++) {
        for (int e = 0; e < 8; e++) y[L][e] = lds[xidx(posC(L, e))];
        const pass_tw t = load_pass_tw(tw.data(), L, 64);
        if (narrow) pass_c_narrow(y[L], t); else pass_bc(y[L], t);
        for (int a = 0; a < 8; a++) {
          if (narrow && a != 0 && a != 7) continue;
          const int k = 64 * a + L;
          if (memcmp(&y[L][a], &ref[k], sizeof(cpx)) != 0) {
            if (bad < 5) printf("mismatch trial %d narrow %d bin %d: %a %a vs %a %a\n", trial, narrow, k,
                                y[L][a].r, y[L][a].i, ref[k].r, ref[k].i);
            bad++;
          }
          if (out_col(L, a) != (k ^ 256)) bad++;
        }
      }
    }
  }
  // ---- the exchange image: a permutation of 0..511, conflict-free in all four access patterns -------
  std::set<int> seen;
  for (int p = 0; p < 512; p++) seen.insert(xidx(p));
  if ((int)seen.size() != 512 || *seen.rbegin() >= XCHG_LEN) { printf("xidx is not a permutation\n"); bad++; }
  auto conflict_free = [&](auto pos, int group, int banks_elems, const char *what) {
    for (int r = 0; r < 8; r++)
      for (int g0 = 0; g0 < 64; g0 += group) {
        std::set<int> b;
        for (int L = g0; L < g0 + group; L++) b.insert(xidx(pos(L, r)) % banks_elems);
        if ((int)b.size() != group) { printf("bank conflict: %s slot %d lanes %d..\n", what, r, g0); bad++; }
      }
  };
  // 8-byte elements: ds_write_b64 in groups of 16 lanes over 32 banks, ds_read_b64 32 lanes over 64 banks
  conflict_free([](int L, int r) { return posA(L, r); }, 16, 16, "write A");
  conflict_free([](int L, int r) { return posB(L, r); }, 32, 32, "read B");
  conflict_free([](int L, int r) { return posB(L, r); }, 16, 16, "write B");
  conflict_free([](int L, int r) { return posC(L, r); }, 32, 32, "read C");
  printf(bad ? "FAIL %d\n" : "OK\n", bad);
  return bad ? 1 : 0;
}
