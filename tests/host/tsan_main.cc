// tsan_main.cc -- the persistent host pool behind uwspr_decode_batch under ThreadSanitizer (CPU build; the
// GPU side has no sanitizers on this pool): many small jobs back to back from two client threads, with
// different thread budgets, results compared with the single-threaded decode.  Built and run by
// tests/test_host_tail.py.
#include <stdio.h>
#include <string.h>

#include <thread>
#include <vector>

#include "../../include/uwspr_hip.h"

int main() {
  const int n = 48;
  std::vector<uwspr_demod_out> recs(n);
  unsigned seed = 777;
  auto rnd = [&]() { seed = seed * 1664525u + 1013904223u; return seed >> 8; };
  for (int r = 0; r < n; r++) {
    uwspr_demod_out &d = recs[r];
    memset(&d, 0, sizeof(d));
    uint8_t data[11] = {0}, enc[176];
    for (int i = 0; i < 6; i++) data[i] = (uint8_t)rnd();
    data[6] = (uint8_t)(rnd() & 0xC0);
    uwspr_fano_encode(enc, data, 11);
    // interleave: the decoder de-interleaves, so build the transmitted order by inverting it on a ramp
    uint8_t ramp[162], order[162];
    for (int i = 0; i < 162; i++) ramp[i] = (uint8_t)i;
    uwspr_deinterleave(ramp);                      // ramp[p] = source position of destination p
    for (int p = 0; p < 162; p++) order[ramp[p]] = (uint8_t)p;
    d.worth_a_try = (r % 7 == 3) ? 0 : 1;
    for (int k = 0; k < UWSPR_NJIG; k++) {
      d.jig_sync[k] = 0.5f; d.jig_rms[k] = 50.0f;
      for (int j = 0; j < 162; j++) {
        const int v = 128 + (enc[order[j]] ? 40 : -40) + (int)(rnd() % 41) - 20;
        d.symbols[k][j] = (uint8_t)(v < 0 ? 0 : v > 255 ? 255 : v);
      }
    }
    if (r % 5 == 0) for (int j = 0; j < 162; j++) d.symbols[0][j] = 128;   // first try: gate passes, nothing to decode
  }
  std::vector<int8_t> ref(7 * n);
  std::vector<uint8_t> refok(n);
  std::vector<int32_t> refidt(n);
  const int good = uwspr_decode_batch(recs.data(), n, 1, ref.data(), refidt.data(), refok.data());
  int bad = 0;
  auto client = [&](int budget) {
    std::vector<int8_t> m(7 * n);
    std::vector<uint8_t> ok(n);
    std::vector<int32_t> idt(n);
    for (int it = 0; it < 60; it++) {
      const int g = uwspr_decode_batch(recs.data(), n, budget, m.data(), idt.data(), ok.data());
      if (g != good || memcmp(m.data(), ref.data(), m.size()) || memcmp(ok.data(), refok.data(), n) ||
          memcmp(idt.data(), refidt.data(), n * sizeof(int32_t)))
        __atomic_fetch_add(&bad, 1, __ATOMIC_RELAXED);
    }
  };
  std::thread a(client, 0), b(client, 3);
  client(2);
  a.join(); b.join();
  printf("tsan ok: %d of %d records decode, %d mismatching jobs, pool of %d\n", good, n, bad, uwspr_host_threads());
  return (bad == 0 && good >= n / 2) ? 0 : 2;
}
