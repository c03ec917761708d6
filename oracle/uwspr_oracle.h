/*
 * uwspr_oracle.h -- TEST INFRASTRUCTURE ONLY.
 *
 * Plain-C CPU restatement of the gr-uwspr coarse (FDR) + fine
 * (sync_and_demodulate) hot path.  It exists to CHECK the HIP path; it is
 * never linked into, imported by, or called from the product
 * (gr-uwspr_amd/, include/).  Only tests/, __graft_entry__.smoke() and the
 * cpu_baseline leg of bench.py may use it.
 *
 * Every function cites the reference file:line it restates (paths relative to
 * the upstream gr-uwspr tree).  Arithmetic follows SURVEY.md Appendix A:
 * "float" = IEEE binary32 with no FMA contraction (build with
 * -ffp-contract=off), "double" = binary64, int casts truncate.
 *
 * Pinning status (see DESIGN.md "Oracle"):
 *   - SLM (slm.cc)                 pinned bit-exact against the real reference
 *                                  object built in oracle/_ref, and lib/slm_qa.cc
 *   - Fano / deinterleave / unpack checked through oracle/_ref (real reference)
 *   - FDR_impl.cc / sync_and_demodulate_impl.cc need GNU Radio, pmt, Boost,
 *     FFTW3f and VOLK, none of which exist in this image => unbuildable here.
 *     The restatement of those two files is pinned by the known answers the
 *     survey recorded from the real reference:
 *     VE3EMB.c2 -> candidate fields, sync to 9 digits, decoded blob/message.
 *   - FFTW3f itself is a third-party, un-vendored, unpinned dependency:
 *     spectrogram values are "parity unpinned" at bit level; the oracle FFT is
 *     a radix-2 DIT checked against a float64 DFT to 1e-5.
 */
#ifndef UWSPR_ORACLE_H
#define UWSPR_ORACLE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* lib/candidate_t.h:27-50 -- same 48-byte layout */
typedef struct { float drift; } orc_mode_linear;
typedef struct { double V1, V2; int32_t p1, p2; } orc_mode_nonlinear;
enum { ORC_LINEAR = 0, ORC_NONLINEAR = 1 };
typedef struct {
  float freq;
  float snr;
  float drift;
  float sync;
  int32_t shift;
  int32_t m_type;
  union {
    orc_mode_linear m_linear;
    orc_mode_nonlinear m_nonlinear;
  };
} orc_candidate;

#define ORC_NSYM 162
#define ORC_NSLM 125
#define ORC_NK0 26
#define ORC_NIFR 5
#define ORC_NJIG 17

/* FDR block state: lib/FDR_impl.cc:48-151 (constructor) */
typedef struct {
  int fs, fl, spb, maxdrift, maxfreqs, halfbandwidth, cf;
  float threshold;
  int size, m, hpbm, n, finpb, noiseidx;
  float df, min_snr;
  float *w;    /* [size] half-sine window, FDR_impl.cc:101-105 */
  float *tw;   /* [size/2][2] FFT twiddles (oracle FFT spec, see .c) */
} orc_fdr;

/* returns 0, or <0 on the parameter errors the reference exit()s on
 * (FDR_impl.cc:85-90) or would read out of bounds on. */
int orc_fdr_init(orc_fdr *f, int fs, int fl, int spb, int maxdrift,
                 int maxfreqs, int halfbandwidth, int cf, int threshold);
void orc_fdr_free(orc_fdr *f);

/* number of hypotheses per (ifr,k0) cell: (2*maxdrift+1) linear + 125 SLM */
int orc_fdr_cell_hyps(const orc_fdr *f);

/* FDR_impl.cc:222-254.  iq = interleaved (I,Q) float pairs [fl]; ps [n][size] */
void orc_fdr_spectrogram(const orc_fdr *f, const float *iq, float *ps);
/* FDR_impl.cc:257-291.  psavg[size]; smraw[finpb] (before normalisation);
 * smspec[finpb] (after); *noise.  Any output pointer may be NULL. */
void orc_fdr_stats(const orc_fdr *f, const float *ps, float *psavg,
                   float *smraw, float *smspec, float *noise);
/* FDR_impl.cc:293-319: local maxima + stable bubble sort. returns npk */
int orc_fdr_peaks(const orc_fdr *f, const float *smspec, orc_candidate *cands);
/* FDR_impl.cc:339-409 for ONE candidate (freq,snr set by peaks).
 * syncgrid (optional) [5][26][cell_hyps] receives every hypothesis metric. */
void orc_fdr_search(const orc_fdr *f, const float *ps, orc_candidate *cand,
                    float *syncgrid);
/* whole handler FDR_impl.cc:214-456; cands[maxfreqs]; returns npk */
int orc_fdr_transform(const orc_fdr *f, const float *iq, orc_candidate *cands);

/* lib/slm.cc:36-73 */
float orc_slm_frequency_drift(double V1, double V2, int p1, int p2, float cf,
                              float t);
/* lib/slm.cc:76-116: idx-th generated instance (0..124); returns 0 past the end */
int orc_slm_generate(int idx, double *V1, double *V2, int *p1, int *p2);

/* lib/sync_and_demodulate_impl.cc:126-256, argument for argument
 * (carrierfrequency is a member there; passed as cf here) */
void orc_sync_and_demodulate(const orc_candidate *cand, int cf,
                             const float *id, const float *qd, long np,
                             unsigned char *symbols, float *f1, int ifmin,
                             int ifmax, float fstep, int *shift1, int lagmin,
                             int lagmax, int lagstep, float *drift1,
                             int symfac, float *sync, int mode);

/* lib/sync_and_demodulate_impl.cc:265-282 */
void orc_deinterleave(unsigned char *sym);

/* Per-candidate refinement schedule S0..S5 of
 * sync_and_demodulate_impl.cc:403-482 WITHOUT the Fano call: all 17 jiggered
 * mode-2 vectors are produced (what the reference computes when Fano never
 * succeeds); the caller replays the rms/sync gates and Fano in order. */
typedef struct {
  float f1;          /* after S4 (or S2 when the gate fails) */
  float drift1;
  float sync1;       /* sync after S4 / S2, before the mode-2 calls */
  int32_t shift1;
  int32_t worth_a_try;
  float jig_sync[ORC_NJIG];                 /* *sync of each mode-2 call */
  float jig_rms[ORC_NJIG];                  /* cc:469-474 */
  int32_t jig_shift[ORC_NJIG];
  unsigned char symbols[ORC_NJIG][ORC_NSYM];/* before deinterleave */
} orc_demod_out;

void orc_demod_candidate(const orc_candidate *cand, int cf, const float *id,
                         const float *qd, long np, orc_demod_out *out);

/* sync_and_demodulate_impl.cc:469-474 */
float orc_symbols_rms(const unsigned char *symbols);

/* Test support for FDR_impl.cc:303 (`10*log10(smspec)`: the binary32 overload, i.e. the host libm's log10f, which is
 * not correctly rounded).  PINNED PLATFORM: glibc 2.35 / x86-64.  Since round 6 the oracle's orc_fdr_peaks calls
 * orc_log10f_glibc235 (use_fma 0), not the host's log10f: a host with another libm (glibc >= 2.40 rounds log10f
 * correctly) cannot change the oracle's `snr`.  orc_log10f_glibc235 restates glibc 2.35's algorithm -- what the HIP kernel computes (k2_spectrum.hip: log10f_glibc235) -- with the multiply-adds of its
 * logf plain (use_fma 0) or fused (1: the build libm selects on CPUs with FMA); orc_log10f_walk counts the binary32
 * patterns lo_bits <= b < hi_bits (step `stride`) on which this host's log10f differs from it (*first_bad = the first);
 * orc_snr_db is cc:303 over an array, as orc_fdr_peaks has it (the restatement). */
float orc_log10f_glibc235(float x, int use_fma);
long orc_log10f_walk(uint32_t lo_bits, uint32_t hi_bits, uint32_t stride, int use_fma, uint32_t *first_bad);
void orc_snr_db(const float *x, float *out, long n);

#ifdef __cplusplus
}
#endif
#endif
