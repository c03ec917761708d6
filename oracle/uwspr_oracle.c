/*
 * uwspr_oracle.c -- TEST INFRASTRUCTURE ONLY (see uwspr_oracle.h).
 *
 * CPU restatement of the gr-uwspr hot path, written from the reference's
 * behaviour (file:line cited per function, paths relative to the upstream
 * tree).  Build with:  gcc -O2 -std=gnu11 -ffp-contract=off
 *
 * Type discipline: every expression below spells out the float/double/int
 * promotions the reference's C++ performs as compiled by g++ >= 6 (where
 * <math.h> exposes the float overloads of sqrt/log10, so sqrt(float) and
 * log10(float) evaluate in binary32).
 */
#include "uwspr_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>

/* lib/pr3.h:5-13 -- the 162 WSPR sync bits, written here as the 21 bytes
 * of their packed form and expanded at first use.  (LSB first.) */
static const unsigned char pr3_packed[21] = {
  0x03, 0x71, 0xa4, 0x07, 0xa4, 0x40, 0xb3, 0x58, 0x58, 0x95, 0x34,
  0x56, 0x04, 0xc9, 0xcd, 0xe2, 0xa0, 0x0c, 0x58, 0x63, 0x00};
static unsigned char pr3[ORC_NSYM];
static int pr3_ready = 0;
static void pr3_init(void) {
  if (pr3_ready) return;
  for (int i = 0; i < ORC_NSYM; i++)
    pr3[i] = (pr3_packed[i >> 3] >> (i & 7)) & 1;
  pr3_ready = 1;
}
const unsigned char *orc_pr3(void) { pr3_init(); return pr3; }

/* ------------------------------------------------------------------ SLM */

/* lib/slm.cc:36-73.  V double, p int, cf/t float; Sign is an int (bool*2-1)
 * stored in float; numerator/denominator double; result rounded to float. */
float orc_slm_frequency_drift(double V1, double V2, int p1, int p2, float cf,
                              float t) {
  const float c = 1500.0f;
  double q1 = V1 * (double)t + (double)p1;
  double q2 = V2 * (double)t + (double)p2;
  float Sign = (float)(((q1 * V1 + q2 * V2) > 0) * 2 - 1);
  double numerator = fabs(V1 * q1 + V2 * q2);
  double denominator = sqrt(pow(q1, 2) + pow(q2, 2));
  if (denominator == 0) return 0.0f;
  return (float)((double)(-Sign) * numerator / denominator * (double)cf /
                 (double)c);
}

/* lib/slm.cc:76-116: p2 fastest, then V1, then V2; p1 always 0. */
int orc_slm_generate(int idx, double *V1, double *V2, int *p1, int *p2) {
  const int nV1 = 5, nV2 = 5, np2 = 5;
  if (idx < 0 || idx >= nV1 * nV2 * np2) return 0;
  int ip2 = idx % np2;
  int iV1 = (idx / np2) % nV1;
  int iV2 = idx / (np2 * nV1);
  *V1 = iV1 * 1.0 + -2.0;
  *V2 = iV2 * 1.0 + -2.0;
  *p1 = 0;
  *p2 = ip2 * 200 + 50;
  return 1;
}

/* ------------------------------------------------------------------ FDR */

int orc_fdr_cell_hyps(const orc_fdr *f) { return 2 * f->maxdrift + 1 + ORC_NSLM; }

/* ifd for a linear hypothesis, FDR_impl.cc:353 (double expression, trunc) */
static int ifd_linear(const orc_fdr *f, int ifr, int k, int drift) {
  return (int)((double)ifr + ((double)(float)k - 81.0) / 81.0 *
                                 (double)(float)drift / (2.0 * (double)f->df));
}
/* ifd for an SLM hypothesis, FDR_impl.cc:382-385 (float expression, trunc) */
static int ifd_slm(const orc_fdr *f, int ifr, int k, double V1, double V2,
                   int p1, int p2) {
  float t = (float)(k * 111 / 162);
  return (int)((float)ifr +
               orc_slm_frequency_drift(V1, V2, p1, p2, (float)f->cf, t) / f->df);
}

/* FDR_impl.cc:48-151 */
int orc_fdr_init(orc_fdr *f, int fs, int fl, int spb, int maxdrift,
                 int maxfreqs, int halfbandwidth, int cf, int threshold) {
  pr3_init();
  memset(f, 0, sizeof(*f));
  f->fs = fs; f->fl = fl; f->spb = spb; f->maxdrift = maxdrift;
  f->maxfreqs = maxfreqs; f->halfbandwidth = halfbandwidth; f->cf = cf;
  f->threshold = (float)threshold;
  f->size = 2 * spb;
  int maxfreq = (int)((float)fs / 2.0);
  if (halfbandwidth > maxfreq) return -1;           /* reference exit(-1)s */
  if (f->size != 512 || maxdrift < 0 || maxfreqs < 1) return -3; /* oracle FFT is 512-point */
  f->df = (float)fs / (float)f->size;
  f->m = f->size / 2;
  f->hpbm = (int)ceil((float)(int)(float)halfbandwidth / f->df);
  f->n = (int)(floor(((float)fl / (float)spb) * 2.0) - 3);
  f->finpb = 2 * f->hpbm;
  f->noiseidx = (int)floor(0.3 * (float)f->finpb);
  f->min_snr = (float)pow(10.0, -7.0 / 10.0);
  if (f->n < 2 * (ORC_NSYM - 1) + ORC_NK0) return -3;
  if ((f->n - 1) * (spb / 2) + f->size > fl) return -3;
  f->w = (float *)malloc(sizeof(float) * f->size);
  for (int i = 0; i < f->size; i++)
    f->w[i] = (float)sin((M_PI / (f->size - 1)) * i);
  /* Oracle FFT twiddles: W[k] = exp(-2*pi*i*k/512), k<256, from double
   * cos/sin rounded to float; k=0 and k=128 are exactly (1,0) and (0,-1). */
  f->tw = (float *)malloc(sizeof(float) * f->size);
  for (int k = 0; k < f->size / 2; k++) {
    double ang = 2.0 * M_PI * (double)k / 512.0;
    f->tw[2 * k] = (float)cos(ang);
    f->tw[2 * k + 1] = (float)(-sin(ang));
  }
  f->tw[0] = 1.0f; f->tw[1] = 0.0f;
  f->tw[2 * 128] = 0.0f; f->tw[2 * 128 + 1] = -1.0f;
  /* Column budget: the reference reads out of bounds when
   * the pass band plus search reach leaves [0,size); reject those. */
  if (f->m - f->hpbm - 3 < 0 || f->m + f->hpbm - 1 + 3 > f->size - 1) {
    orc_fdr_free(f);
    return -2;
  }
  for (int ifr = f->m - f->hpbm + 1 - 2; ifr <= f->m + f->hpbm - 2 + 2; ifr++) {
    for (int k = 0; k < ORC_NSYM; k++) {
      for (int d = -maxdrift; d <= maxdrift; d++) {
        int ifd = ifd_linear(f, ifr, k, d);
        if (ifd - 3 < 0 || ifd + 3 > f->size - 1) { orc_fdr_free(f); return -2; }
      }
      for (int s = 0; s < ORC_NSLM; s++) {
        double V1, V2; int p1, p2;
        orc_slm_generate(s, &V1, &V2, &p1, &p2);
        int ifd = ifd_slm(f, ifr, k, V1, V2, p1, p2);
        if (ifd - 3 < 0 || ifd + 3 > f->size - 1) { orc_fdr_free(f); return -2; }
      }
    }
  }
  return 0;
}

void orc_fdr_free(orc_fdr *f) {
  free(f->w); free(f->tw);
  f->w = NULL; f->tw = NULL;
}

/* Oracle FFT specification: in-place iterative radix-2 decimation-in-time,
 * bit-reversed input order, 9 stages; butterfly
 *     t = W * x[b+j+h]  (tr = wr*xr - wi*xi ; ti = wr*xi + wi*xr)
 *     x[b+j] = u + t ; x[b+j+h] = u - t
 * all in binary32 without contraction.  The reference calls FFTW3f
 * (FDR_impl.cc:123-132,244), which is not vendored; any correct DFT agrees
 * with it to fp32 rounding only. */
static void fft512(const float *tw, float *re, float *im) {
  for (int p = 0; p < 512; p++) {
    int q = 0;
    for (int b = 0; b < 9; b++) q |= ((p >> b) & 1) << (8 - b);
    if (q > p) {
      float t = re[p]; re[p] = re[q]; re[q] = t;
      t = im[p]; im[p] = im[q]; im[q] = t;
    }
  }
  for (int h = 1; h < 512; h <<= 1) {
    int step = 256 / h;
    for (int b = 0; b < 512; b += 2 * h) {
      for (int j = 0; j < h; j++) {
        float wr = tw[2 * j * step], wi = tw[2 * j * step + 1];
        float xr = re[b + j + h], xi = im[b + j + h];
        float tr = wr * xr - wi * xi;
        float ti = wr * xi + wi * xr;
        float ur = re[b + j], ui = im[b + j];
        re[b + j] = ur + tr; im[b + j] = ui + ti;
        re[b + j + h] = ur - tr; im[b + j + h] = ui - ti;
      }
    }
  }
}

/* FDR_impl.cc:222-254 */
void orc_fdr_spectrogram(const orc_fdr *f, const float *iq, float *ps) {
  float re[512], im[512];
  int size = f->size, spb = f->spb;
  for (int i = 0; i < f->n; i++) {
    for (int j = 0; j < size; j++) {
      int k = i * (spb / 2) + j;
      /* cc:230-231: complex<double> * float w, stored to float == one
       * binary32 multiply (24x24-bit product is exact in double) */
      re[j] = (float)((double)iq[2 * k] * (double)f->w[j]);
      im[j] = (float)((double)iq[2 * k + 1] * (double)f->w[j]);
    }
    fft512(f->tw, re, im);
    for (int j = 0; j < size; j++) {
      int k = j + spb;               /* cc:247-248 fftshift */
      if (k > size - 1) k = k - size;
      ps[i * size + j] = re[k] * re[k] + im[k] * im[k];
    }
  }
}

/* FDR_impl.cc:168-173 */
static int floatcomp(const void *a, const void *b) {
  if (*(const float *)a < *(const float *)b) return -1;
  return *(const float *)a > *(const float *)b;
}

/* FDR_impl.cc:257-291 */
void orc_fdr_stats(const orc_fdr *f, const float *ps, float *psavg_out,
                   float *smraw, float *smspec_out, float *noise_out) {
  int size = f->size, n = f->n, finpb = f->finpb;
  float *psavg = (float *)malloc(sizeof(float) * size);
  float *smspec = (float *)malloc(sizeof(float) * finpb);
  float *tmpsort = (float *)malloc(sizeof(float) * finpb);
  for (int j = 0; j < size; j++) {
    psavg[j] = 0;
    for (int i = 0; i < n; i++) psavg[j] = psavg[j] + ps[i * size + j];
  }
  for (int i = 0; i < finpb; i++) {
    smspec[i] = 0.0f;
    for (int j = -3; j <= 3; j++) {
      int k = f->m - f->hpbm + i + j;
      smspec[i] = smspec[i] + psavg[k];
    }
  }
  if (smraw) memcpy(smraw, smspec, sizeof(float) * finpb);
  memcpy(tmpsort, smspec, sizeof(float) * finpb);
  qsort(tmpsort, finpb, sizeof(float), floatcomp);
  float noise_level = tmpsort[f->noiseidx];
  for (int j = 0; j < finpb; j++) {
    smspec[j] = (float)((double)(smspec[j] / noise_level) - 1.0);
    if (smspec[j] < f->min_snr) smspec[j] = (float)(0.1 * (double)f->min_snr);
  }
  if (psavg_out) memcpy(psavg_out, psavg, sizeof(float) * size);
  if (smspec_out) memcpy(smspec_out, smspec, sizeof(float) * finpb);
  if (noise_out) *noise_out = noise_level;
  free(psavg); free(smspec); free(tmpsort);
}

/* FDR_impl.cc:293-319 */
int orc_fdr_peaks(const orc_fdr *f, const float *smspec, orc_candidate *cands) {
  int npk = 0;
  for (int j = 1; j < f->finpb - 1; j++) {
    if ((smspec[j] > smspec[j - 1]) && (smspec[j] > smspec[j + 1]) &&
        (npk < f->maxfreqs)) {
      memset(&cands[npk], 0, sizeof(orc_candidate));
      cands[npk].freq = (float)(j - f->hpbm) * f->df;
      /* cc:303: log10(float) resolves to the binary32 overload (g++ >= 6), and libm's log10f is not correctly
       * rounded: its last bit belongs to the C library.  The pinned platform is glibc 2.35 / x86-64 (this image, where
       * oracle/_ref is built and run); the oracle calls the RESTATEMENT of that log10f, not the host's, so that a host
       * with another libm changes nothing here (tests/test_log10_gap.py states whether this host's log10f is that one) */
      cands[npk].snr = (float)10 * orc_log10f_glibc235(smspec[j], 0);
      npk++;
    }
  }
  for (int pass = 1; pass <= npk - 1; pass++) {
    for (int k = 0; k < npk - pass; k++) {
      if (cands[k].snr < cands[k + 1].snr) {
        orc_candidate tmp = cands[k];
        cands[k] = cands[k + 1];
        cands[k + 1] = tmp;
      }
    }
  }
  return npk;
}

/* FDR_impl.cc:188-210 */
static void powersum(const orc_fdr *f, const float *ps, int k0, int k, int ifd,
                     float *ss, float *pw) {
  float p[4];
  int kindex = k0 + 2 * k;
  const float *row = ps + (size_t)kindex * f->size;
  p[0] = sqrtf(row[ifd - 3]);
  p[1] = sqrtf(row[ifd - 1]);
  p[2] = sqrtf(row[ifd + 1]);
  p[3] = sqrtf(row[ifd + 3]);
  *ss = *ss + (float)(2 * pr3[k] - 1) * ((p[1] + p[3]) - (p[0] + p[2]));
  *pw = *pw + p[0] + p[1] + p[2] + p[3];
}

/* FDR_impl.cc:339-409, one candidate */
void orc_fdr_search(const orc_fdr *f, const float *ps, orc_candidate *cand,
                    float *syncgrid) {
  int hc = orc_fdr_cell_hyps(f);
  cand->sync = (float)-1e30;
  int if0 = (int)(cand->freq / f->df + (float)f->m);
  for (int ifr = if0 - 2; ifr <= if0 + 2; ifr++) {
    for (int k0 = 0; k0 < ORC_NK0; k0++) {
      int h = 0;
      for (int drift = -f->maxdrift; drift <= f->maxdrift; drift++, h++) {
        float ss = 0.0f, pw = 0.0f;
        for (int k = 0; k < ORC_NSYM; k++)
          powersum(f, ps, k0, k, ifd_linear(f, ifr, k, drift), &ss, &pw);
        float sync = ss / pw;
        if (syncgrid)
          syncgrid[((size_t)(ifr - (if0 - 2)) * ORC_NK0 + k0) * hc + h] = sync;
        if (sync > cand->sync) {
          cand->shift = 128 * k0;
          cand->freq = (float)(ifr - f->m) * f->df;
          cand->sync = sync;
          cand->m_type = ORC_LINEAR;
          cand->m_linear.drift = (float)drift;
        }
      }
      for (int s = 0; s < ORC_NSLM; s++, h++) {
        double V1, V2; int p1, p2;
        orc_slm_generate(s, &V1, &V2, &p1, &p2);
        float ss = 0.0f, pw = 0.0f;
        for (int k = 0; k < ORC_NSYM; k++)
          powersum(f, ps, k0, k, ifd_slm(f, ifr, k, V1, V2, p1, p2), &ss, &pw);
        float sync = ss / pw;
        if (syncgrid)
          syncgrid[((size_t)(ifr - (if0 - 2)) * ORC_NK0 + k0) * hc + h] = sync;
        if (sync / cand->sync > f->threshold) {
          cand->shift = 128 * k0;
          cand->freq = (float)(ifr - f->m) * f->df;
          cand->sync = sync;
          cand->m_type = ORC_NONLINEAR;
          cand->m_nonlinear.V1 = V1;
          cand->m_nonlinear.V2 = V2;
          cand->m_nonlinear.p1 = p1;
          cand->m_nonlinear.p2 = p2;
        }
      }
    }
  }
}

/* FDR_impl.cc:214-456 */
int orc_fdr_transform(const orc_fdr *f, const float *iq, orc_candidate *cands) {
  float *ps = (float *)malloc(sizeof(float) * (size_t)f->n * f->size);
  float *smspec = (float *)malloc(sizeof(float) * f->finpb);
  orc_fdr_spectrogram(f, iq, ps);
  orc_fdr_stats(f, ps, NULL, NULL, smspec, NULL);
  int npk = orc_fdr_peaks(f, smspec, cands);
  for (int j = 0; j < npk; j++) orc_fdr_search(f, ps, &cands[j], NULL);
  free(ps); free(smspec);
  return npk;
}

/* ----------------------------------------------------- sync_and_demodulate */

/* sync_and_demodulate_impl.cc:126-256 */
void orc_sync_and_demodulate(const orc_candidate *cand, int cf,
                             const float *id, const float *qd, long np,
                             unsigned char *symbols, float *f1, int ifmin,
                             int ifmax, float fstep, int *shift1, int lagmin,
                             int lagmax, int lagstep, float *drift1,
                             int symfac, float *sync, int mode) {
  pr3_init();
  float fplast = -10000.0f;
  const float dt = (float)(1.0 / 375.0), df = (float)(375.0 / 256.0);
  float delta[4] = {(float)(-(double)df * 1.5), (float)(-(double)df * 0.5),
                    (float)((double)df * 0.5), (float)((double)df * 1.5)};
  int i, j, k, lag, n;
  float inp[4], quad[4];
  float p[4];
  float cmet, totp, syncmax, fac;
  float c[4][256], s[4][256];
  float cdphi, sdphi;
  float ss;
  float f0 = 0.0f, fp = 0.0f, fbest = 0.0f, fsum = 0.0f, f2sum = 0.0f;
  float fsymb[ORC_NSYM];
  int best_shift = 0, ifreq;
  /* cc:158,177: `t` is read uninitialised in the nonlinear branch (its
   * assignment is dead code); every observed build behaves as t = 0. */
  float t = 0.0f;
  memset(fsymb, 0, sizeof(fsymb));
  syncmax = (float)-1e30;
  if (mode == 0) { ifmin = 0; ifmax = 0; fstep = 0.0f; f0 = *f1; }
  if (mode == 1) { lagmin = *shift1; lagmax = *shift1; f0 = *f1; }
  if (mode == 2) { lagmin = *shift1; lagmax = *shift1; ifmin = 0; ifmax = 0; f0 = *f1; }
  for (ifreq = ifmin; ifreq <= ifmax; ifreq++) {
    f0 = *f1 + (float)ifreq * fstep;
    for (lag = lagmin; lag <= lagmax; lag = lag + lagstep) {
      ss = 0.0f; totp = 0.0f;
      for (i = 0; i < ORC_NSYM; i++) {
        if (cand->m_type == ORC_LINEAR) {
          fp = (float)((double)f0 + ((double)*drift1 / 2.0) *
                                        ((double)(float)i - 81.0) / 81.0);
        } else {
          fp = f0 + orc_slm_frequency_drift(cand->m_nonlinear.V1,
                                            cand->m_nonlinear.V2,
                                            cand->m_nonlinear.p1,
                                            cand->m_nonlinear.p2, (float)cf, t);
        }
        if (i == 0 || (fp != fplast)) {
          for (j = 0; j < 4; j++) {
            double ang = 2 * M_PI * (double)dt * (double)(fp + delta[j]);
            cdphi = (float)cos(ang);
            sdphi = (float)sin(ang);
            c[j][0] = 1; s[j][0] = 0;
            for (k = 1; k < 256; k++) {
              c[j][k] = c[j][k - 1] * cdphi - s[j][k - 1] * sdphi;
              s[j][k] = c[j][k - 1] * sdphi + s[j][k - 1] * cdphi;
            }
            fplast = fp;
          }
        }
        for (j = 0; j < 4; j++) {
          inp[j] = 0.0f; quad[j] = 0.0f;
          for (k = 0; k < 256; k++) {
            n = lag + i * 256 + k;
            if ((n > 0) && (n < np)) {
              inp[j] = inp[j] + id[n] * c[j][k] + qd[n] * s[j][k];
              quad[j] = quad[j] - id[n] * s[j][k] + qd[n] * c[j][k];
            }
          }
          p[j] = sqrtf(inp[j] * inp[j] + quad[j] * quad[j]);
        }
        totp = totp + p[0] + p[1] + p[2] + p[3];
        cmet = (p[1] + p[3]) - (p[0] + p[2]);
        ss = (pr3[i] == 1) ? ss + cmet : ss - cmet;
        if (mode == 2) {
          if (pr3[i] == 1) fsymb[i] = p[3] - p[1];
          else fsymb[i] = p[2] - p[0];
        }
      }
      ss = ss / totp;
      if (ss > syncmax) {
        syncmax = ss;
        best_shift = lag;
        fbest = f0;
      }
    }
  }
  if (mode <= 1) {
    *sync = syncmax;
    *shift1 = best_shift;
    *f1 = fbest;
    return;
  }
  if (mode == 2) {
    *sync = syncmax;
    for (i = 0; i < ORC_NSYM; i++) {
      fsum = (float)((double)fsum + (double)fsymb[i] / 162.0);
      f2sum = (float)((double)f2sum + (double)(fsymb[i] * fsymb[i]) / 162.0);
    }
    fac = sqrtf(f2sum - fsum * fsum);
    for (i = 0; i < ORC_NSYM; i++) {
      fsymb[i] = (float)symfac * fsymb[i] / fac;
      if (fsymb[i] > 127) fsymb[i] = 127.0f;
      if (fsymb[i] < -128) fsymb[i] = -128.0f;
      float v = fsymb[i] + 128;
      /* (unsigned char)NaN is undefined in C; this build defines it as 0 */
      symbols[i] = (v != v) ? 0 : (unsigned char)v;
    }
  }
}

/* sync_and_demodulate_impl.cc:265-282 */
void orc_deinterleave(unsigned char *sym) {
  unsigned char tmp[ORC_NSYM];
  unsigned char p = 0, i = 0, j;
  while (p < ORC_NSYM) {
    j = (unsigned char)(((i * 0x80200802ULL) & 0x0884422110ULL) * 0x0101010101ULL >> 32);
    if (j < ORC_NSYM) { tmp[p] = sym[j]; p = p + 1; }
    i = i + 1;
  }
  for (i = 0; i < ORC_NSYM; i++) sym[i] = tmp[i];
}

/* sync_and_demodulate_impl.cc:469-474 */
float orc_symbols_rms(const unsigned char *symbols) {
  float sq = 0.0f;
  for (int i = 0; i < ORC_NSYM; i++) {
    float y = (float)((double)(float)symbols[i] - 128.0);
    sq += y * y;
  }
  return (float)sqrt((double)sq / 162.0);
}

/* sync_and_demodulate_impl.cc:403-482 for one candidate, Fano left out */
void orc_demod_candidate(const orc_candidate *cand_in, int cf, const float *id,
                         const float *qd, long np, orc_demod_out *out) {
  orc_candidate cand = *cand_in;
  unsigned char symbols[ORC_NSYM];
  const float minsync1 = 0.10f;
  const int iifac = 8, symfac = 50;
  float f1, fstep, sync1, drift1;
  int shift1, lagmin, lagmax, lagstep, ifmin, ifmax, worth_a_try;
  memset(out, 0, sizeof(*out));
  /* cc:373: the PDU unpack writes m_linear.drift = 0 over the nonlinear union */
  if (cand.m_type == ORC_NONLINEAR) cand.m_linear.drift = 0;
  memset(symbols, 0, sizeof(symbols));
  f1 = cand.freq;
  drift1 = cand.m_linear.drift;
  shift1 = cand.shift;
  sync1 = cand.sync;
  fstep = 0.0f; ifmin = 0; ifmax = 0;
  lagmin = shift1 - 128; lagmax = shift1 + 128; lagstep = 64;
  orc_sync_and_demodulate(&cand, cf, id, qd, np, symbols, &f1, ifmin, ifmax,
                          fstep, &shift1, lagmin, lagmax, lagstep, &drift1,
                          symfac, &sync1, 0);
  fstep = 0.25f; ifmin = -2; ifmax = 2;
  orc_sync_and_demodulate(&cand, cf, id, qd, np, symbols, &f1, ifmin, ifmax,
                          fstep, &shift1, lagmin, lagmax, lagstep, &drift1,
                          symfac, &sync1, 1);
  if (cand.m_type == ORC_LINEAR) {
    fstep = 0.0f; ifmin = 0; ifmax = 0;
    float driftp, driftm, syncp, syncm;
    driftp = (float)((double)drift1 + 0.5);
    orc_sync_and_demodulate(&cand, cf, id, qd, np, symbols, &f1, ifmin, ifmax,
                            fstep, &shift1, lagmin, lagmax, lagstep, &driftp,
                            symfac, &syncp, 1);
    driftm = (float)((double)drift1 - 0.5);
    orc_sync_and_demodulate(&cand, cf, id, qd, np, symbols, &f1, ifmin, ifmax,
                            fstep, &shift1, lagmin, lagmax, lagstep, &driftm,
                            symfac, &syncm, 1);
    if (syncp > sync1) { drift1 = driftp; sync1 = syncp; }
    else if (syncm > sync1) { drift1 = driftm; sync1 = syncm; }
  }
  if (sync1 > minsync1) {
    lagmin = shift1 - 32; lagmax = shift1 + 32; lagstep = 16;
    orc_sync_and_demodulate(&cand, cf, id, qd, np, symbols, &f1, ifmin, ifmax,
                            fstep, &shift1, lagmin, lagmax, lagstep, &drift1,
                            symfac, &sync1, 0);
    fstep = 0.05f; ifmin = -2; ifmax = 2;
    orc_sync_and_demodulate(&cand, cf, id, qd, np, symbols, &f1, ifmin, ifmax,
                            fstep, &shift1, lagmin, lagmax, lagstep, &drift1,
                            symfac, &sync1, 1);
    worth_a_try = 1;
  } else {
    worth_a_try = 0;
  }
  out->f1 = f1; out->drift1 = drift1; out->sync1 = sync1; out->shift1 = shift1;
  out->worth_a_try = worth_a_try;
  for (int idt = 0; worth_a_try && idt <= (128 / iifac); idt++) {
    int ii = (idt + 1) / 2;
    if (idt % 2 == 1) ii = -ii;
    ii = iifac * ii;
    int jiggered_shift = shift1 + ii;
    float s2 = sync1;
    out->jig_shift[idt] = jiggered_shift;
    orc_sync_and_demodulate(&cand, cf, id, qd, np, out->symbols[idt], &f1,
                            ifmin, ifmax, fstep, &jiggered_shift, lagmin,
                            lagmax, lagstep, &drift1, symfac, &s2, 2);
    out->jig_sync[idt] = s2;
    out->jig_rms[idt] = orc_symbols_rms(out->symbols[idt]);
  }
}

/* ---- test support for FDR_impl.cc:303: see uwspr_oracle.h ----
 * glibc 2.35 (Ubuntu GLIBC 2.35-0ubuntu3.x, this image's libm): sysdeps/ieee754/flt-32/e_logf.c with the table of
 * sysdeps/ieee754/flt-32/logf_data.c (LOGF_TABLE_BITS 4, LOGF_POLY_ORDER 4), and sysdeps/ieee754/flt-32/e_log10f.c
 * around it, restated operation for operation.  `fma` != 0: the multiply-adds of e_logf.c fused as the compiler fuses
 * them in the build libm selects on CPUs with FMA (sysdeps/x86_64/fpu/multiarch/e_logf.c). */
static const double glibc_logf_tab[16][2] = {
    {0x1.661ec79f8f3bep+0, -0x1.57bf7808caadep-2}, {0x1.571ed4aaf883dp+0, -0x1.2bef0a7c06ddbp-2},
    {0x1.49539f0f010bp+0, -0x1.01eae7f513a67p-2},  {0x1.3c995b0b80385p+0, -0x1.b31d8a68224e9p-3},
    {0x1.30d190c8864a5p+0, -0x1.6574f0ac07758p-3}, {0x1.25e227b0b8eap+0, -0x1.1aa2bc79c81p-3},
    {0x1.1bb4a4a1a343fp+0, -0x1.a4e76ce8c0e5ep-4}, {0x1.12358f08ae5bap+0, -0x1.1973c5a611cccp-4},
    {0x1.0953f419900a7p+0, -0x1.252f438e10c1ep-5}, {0x1p+0, 0x0p+0},
    {0x1.e608cfd9a47acp-1, 0x1.aa5aa5df25984p-5},  {0x1.ca4b31f026aap-1, 0x1.c5e53aa362eb4p-4},
    {0x1.b2036576afce6p-1, 0x1.526e57720db08p-3},  {0x1.9c2d163a1aa2dp-1, 0x1.bc2860d22477p-3},
    {0x1.886e6037841edp-1, 0x1.1058bc8a07ee1p-2},  {0x1.767dcf5534862p-1, 0x1.4043057b6ee09p-2}};
static uint32_t f2u(float f) { uint32_t u; memcpy(&u, &f, 4); return u; }
static float u2f(uint32_t u) { float f; memcpy(&f, &u, 4); return f; }

static float glibc235_logf(float x, int use_fma) {
  const double Ln2 = 0x1.62e42fefa39efp-1;
  const double A0 = -0x1.00ea348b88334p-2, A1 = 0x1.5575b0be00b6ap-2, A2 = -0x1.ffffef20a4123p-2;
  uint32_t ix = f2u(x), tmp, iz;
  int i, k;
  double z, r, r2, y, y0, invc, logc;
  if (ix == 0x3f800000u) return 0.0f;
  if (ix - 0x00800000u >= 0x7f800000u - 0x00800000u) {
    if (ix * 2 == 0) return -INFINITY;
    if (ix == 0x7f800000u) return x;
    if ((ix & 0x80000000u) || ix * 2 >= 0xff000000u) return NAN;
    ix = f2u(x * 0x1p23f);
    ix -= 23u << 23;
  }
  tmp = ix - 0x3f330000u;
  i = (int)((tmp >> 19) % 16);
  k = (int32_t)tmp >> 23;
  iz = ix - (tmp & (0x1ffu << 23));
  invc = glibc_logf_tab[i][0];
  logc = glibc_logf_tab[i][1];
  z = (double)u2f(iz);
  if (use_fma) {
    r = fma(z, invc, -1.0);
    y0 = fma((double)k, Ln2, logc);
    r2 = r * r;
    y = fma(A1, r, A2);
    y = fma(A0, r2, y);
    y = fma(y, r2, y0 + r);
  } else {
    r = z * invc - 1;
    y0 = logc + (double)k * Ln2;
    r2 = r * r;
    y = A1 * r + A2;
    y = A0 * r2 + y;
    y = y * r2 + (y0 + r);
  }
  return (float)y;
}

float orc_log10f_glibc235(float x, int use_fma) {
  const float two25 = 3.3554432000e+07f, ivln10 = 4.3429449201e-01f, log10_2hi = 3.0102920532e-01f,
              log10_2lo = 7.9034151668e-07f;
  float y, z;
  int32_t i, k = 0, hx = (int32_t)f2u(x);
  if (hx < 0x00800000) {
    if ((hx & 0x7fffffff) == 0) return -two25 / fabsf(x);
    if (hx < 0) return (x - x) / (x - x);
    k -= 25;
    x *= two25;
    hx = (int32_t)f2u(x);
  }
  if (hx >= 0x7f800000) return x + x;
  k += (hx >> 23) - 127;
  i = (int32_t)(((uint32_t)k & 0x80000000u) >> 31);
  hx = (hx & 0x007fffff) | ((0x7f - i) << 23);
  y = (float)(k + i);
  x = u2f((uint32_t)hx);
  z = y * log10_2lo + ivln10 * glibc235_logf(x, use_fma);
  return z + y * log10_2hi;
}

long orc_log10f_walk(uint32_t lo_bits, uint32_t hi_bits, uint32_t stride, int use_fma, uint32_t *first_bad) {
  long bad = 0;
  if (stride == 0) stride = 1;
  for (uint64_t b = lo_bits; b < hi_bits; b += stride) {
    float x = u2f((uint32_t)b);
    uint32_t ga = f2u(log10f(x)), gb = f2u(orc_log10f_glibc235(x, use_fma));
    if (ga != gb && !(ga * 2 > 0xff000000u && gb * 2 > 0xff000000u)) {      /* (two NaNs are equal here) */
      if (!bad && first_bad) *first_bad = (uint32_t)b;
      bad++;
    }
  }
  return bad;
}

void orc_snr_db(const float *x, float *out, long n) {
  for (long i = 0; i < n; i++) out[i] = (float)10 * orc_log10f_glibc235(x[i], 0);   /* cc:303 as orc_fdr_peaks has it */
}
