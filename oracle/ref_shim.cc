/*
 * ref_shim.cc -- TEST INFRASTRUCTURE ONLY.
 *
 * extern "C" doorways into the pieces of the real gr-uwspr reference that
 * compile from their own source files with plain g++ (no GNU Radio, pmt,
 * Boost, FFTW or VOLK needed): lib/slm.cc, lib/Fano.cc (+ lib/tab.c,
 * lib/metric_tables.c) and lib/helpers.cc.  oracle/Makefile compiles those
 * files WHERE THEY LIE under /root/reference and links them with this shim
 * into oracle/_ref/libuwspr_ref.so.  Nothing of the reference is copied into
 * this repository: the .so is git-ignored (it stays out of history).  It is NOT
 * gpurun-ignored: like the product's own built libraries it travels with the
 * tree to the GPU box, as the build rules of this project prescribe for
 * oracle/_ref, so the live pin tests (tests/test_oracle_pins.py,
 * tests/test_host_tail.py) run wherever the file is; where it is absent they
 * skip and the committed fixtures under tests/golden/ that were generated
 * through it (fano_ref.npz, slm_table_ref.npy) carry the same pins.  No
 * reference SOURCE travels or is committed.
 *
 * lib/FDR_impl.cc and lib/sync_and_demodulate_impl.cc are NOT buildable here
 * (they need gnuradio/pmt/boost/fftw3/volk headers and libraries that this
 * image lacks), so they are not part of _ref; see oracle/uwspr_oracle.c.
 */
#include <string.h>
#include <stdlib.h>

#include "slm.h"      /* /root/reference/lib */
#include "Fano.h"
#include "helpers.h"

using namespace gr::uwspr;

extern "C" {

/* lib/slm.cc:36 */
float ref_slm_frequency_drift(double V1, double V2, int p1, int p2, float cf,
                              float t) {
  SLM slm;
  mode_nonlinear m;
  m.V1 = V1; m.V2 = V2; m.p1 = p1; m.p2 = p2;
  return slm.slmFrequencyDrift(m, cf, t);
}

/* lib/slm.cc:76,118: run the generator to exhaustion; returns the count */
int ref_slm_generate_all(double *V1, double *V2, int *p1, int *p2, int cap) {
  SLM slm;
  mode_nonlinear m;
  int n = 0;
  slm.slmGeneratorInit();
  while (slm.slmGenerator(&m)) {
    if (n < cap) { V1[n] = m.V1; V2[n] = m.V2; p1[n] = m.p1; p2[n] = m.p2; }
    n++;
  }
  return n;
}

/* lib/Fano.cc:36-45 */
void ref_fano_mettab(int *out /* [2][256] */) {
  Fano f;
  memcpy(out, f.mettab, sizeof(int) * 2 * 256);
}

/* lib/Fano.cc:81 */
int ref_fano_encode(unsigned char *symbols, unsigned char *data,
                    unsigned int nbytes) {
  Fano f;
  return f.encode(symbols, data, nbytes);
}

/* lib/Fano.cc:110, with the block's own metric table */
int ref_fano_decode(unsigned int *metric, unsigned int *cycles,
                    unsigned int *maxnp, unsigned char *data,
                    unsigned char *symbols, unsigned int nbits, int delta,
                    unsigned int maxcycles) {
  Fano f;
  return f.fano(metric, cycles, maxnp, data, symbols, nbits, f.mettab, delta,
                maxcycles);
}

/* lib/helpers.cc unpk_ (called at WSPR_unpacker_impl.cc:129) with an empty
 * hash table; out must hold 23 bytes. */
int ref_unpk(const signed char *message7, char *call_loc_pow) {
  helpers h;
  char msg[12];
  memset(msg, 0, sizeof(msg));
  memcpy(msg, message7, 7);
  char *hashtab = (char *)calloc(32768 * 13, 1);
  char callsign[13];
  memset(callsign, 0, sizeof(callsign));
  memset(call_loc_pow, 0, 23);
  int r = h.unpk_(msg, hashtab, call_loc_pow, callsign);
  free(hashtab);
  return r;
}

}  // extern "C"
