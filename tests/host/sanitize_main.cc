// sanitize_main.cc -- the host tail (Fano, de-interleave, unpack, .c2 reader,
// gate/retry loop) under AddressSanitizer + UBSan on the CPU build (GPU ASan is
// not available on this pool).  Built and run by tests/test_host_tail.py.
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "../../include/uwspr_hip.h"

int main(int argc, char **argv) {
  unsigned seed = 12345;
  auto rnd = [&]() { seed = seed * 1664525u + 1013904223u; return seed >> 8; };
  int decoded = 0, timeouts = 0;
  for (int t = 0; t < 40; t++) {
    uint8_t data[11] = {0}, enc[176], soft[162], out[11];
    for (int i = 0; i < 6; i++) data[i] = (uint8_t)rnd();
    data[6] = (uint8_t)(rnd() & 0xC0);
    uwspr_fano_encode(enc, data, 11);
    const int noise = 10 + 12 * (t % 8);
    for (int i = 0; i < 162; i++) {
      int v = 128 + (enc[i] ? 45 : -45) + (int)(rnd() % (2 * noise + 1)) - noise;
      soft[i] = (uint8_t)(v < 0 ? 0 : v > 255 ? 255 : v);
    }
    uint32_t metric, cycles, maxnp;
    int rc = uwspr_fano_decode(soft, out, &metric, &cycles, &maxnp, 60, 10000);
    if (rc == 0 && memcmp(out, data, 7) == 0) decoded++; else timeouts++;
    uint8_t tmp[162];
    memcpy(tmp, soft, 162);
    uwspr_deinterleave(tmp);
    char txt[32];
    int8_t m7[7];
    for (int i = 0; i < 7; i++) m7[i] = (int8_t)rnd();
    uwspr_unpack_message(m7, txt, sizeof(txt));
  }
  // pure noise must time out without touching anything out of bounds
  for (int t = 0; t < 3; t++) {
    uint8_t soft[162], out[11];
    for (int i = 0; i < 162; i++) soft[i] = (uint8_t)rnd();
    uint32_t metric, cycles, maxnp;
    if (uwspr_fano_decode(soft, out, &metric, &cycles, &maxnp, 60, 10000) != 0) timeouts++;
  }
  uwspr_demod_out d;
  memset(&d, 0, sizeof(d));
  int8_t msg[7]; int32_t idt;
  if (uwspr_decode_candidate(&d, msg, &idt)) return 3;   // worth_a_try == 0
  d.worth_a_try = 1;
  for (int k = 0; k < UWSPR_NJIG; k++) { d.jig_sync[k] = 0.5f; d.jig_rms[k] = 50.0f; memset(d.symbols[k], 128, 162); }
  uwspr_decode_candidate(&d, msg, &idt);                  // 17 time-outs / trivial decode
  if (argc > 1) {
    static float iq[90000];
    double f; int32_t ty;
    if (uwspr_c2_read(argv[1], iq, &f, &ty) != 0) return 4;
    if (uwspr_c2_read("/nonexistent.c2", iq, &f, &ty) == 0) return 5;
  }
  printf("sanitized ok: decoded %d, not decoded %d\n", decoded, timeouts);
  return decoded >= 20 ? 0 : 2;
}
