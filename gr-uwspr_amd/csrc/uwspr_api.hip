// uwspr_api.hip -- the C ABI of include/uwspr_hip.h: context set-up (the
// constructors of FDR_impl / sync_and_demodulate_impl), constant tables, HBM
// scratch management and the launch sequences.  There is no CPU fallback: with
// no usable HIP device every entry point fails with UWSPR_ERR_NODEVICE.
#include <errno.h>
#include <math.h>
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <vector>

#include "uwspr_internal.h"

namespace uwspr {
int coarse_configure(const fdr_consts &f);
size_t coarse_lds_bytes(const fdr_consts &f);
int coarse_seq_words();
size_t coarse_plan(fdr_consts &f, int pitch_opt, int tile_opt);
uint32_t coarse_seq_entry(const fdr_consts &f, int k, int off);
}  // namespace uwspr

using namespace uwspr;

// --------------------------------------------------------------- utilities
static int fail(uwspr_ctx *c, int status, const char *fmt, ...) {
  if (c) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(c->err, sizeof(c->err), fmt, ap);
    va_end(ap);
  }
  return status;
}

#define HIPCHK(c, call)                                                                  \
  do {                                                                                   \
    hipError_t e_ = (call);                                                              \
    if (e_ != hipSuccess)                                                                \
      return fail((c), UWSPR_ERR_HIP, "%s failed: %s (%s:%d)", #call, hipGetErrorString(e_), \
                  __FILE__, __LINE__);                                                   \
  } while (0)

template <typename T>
static int ensure(uwspr_ctx *c, T **buf, size_t *cap, size_t need_elems) {
  if (need_elems <= *cap && *buf) return UWSPR_OK;
  if (*buf) { HIPCHK(c, hipStreamSynchronize(c->stream)); HIPCHK(c, hipFree(*buf)); *buf = nullptr; *cap = 0; }
  size_t n = std::max<size_t>(need_elems, 1);
  hipError_t e = hipMalloc((void **)buf, n * sizeof(T));
  if (e != hipSuccess) return fail(c, UWSPR_ERR_NOMEM, "hipMalloc(%zu bytes): %s", n * sizeof(T), hipGetErrorString(e));
  *cap = n;
  return UWSPR_OK;
}

prof_scope::prof_scope(uwspr_ctx *cx, int kind, int64_t units, bool ext_) : c(cx), idx(-1), ext(ext_) {
  if (!(c->prof_mask & (1 << kind))) return;
  ev_pair p;
  auto get = [&]() {
    hipEvent_t e;
    if (!c->ev_pool.empty()) { e = c->ev_pool.back(); c->ev_pool.pop_back(); }
    else (void)hipEventCreate(&e);
    return e;
  };
  p.a = get(); p.b = get(); p.kind = kind; p.units = units;
  if (!ext) (void)hipEventRecord(p.a, c->stream);
  c->prof_events.push_back(p);
  idx = (int)c->prof_events.size() - 1;
}
prof_scope::~prof_scope() {
  if (idx >= 0 && !ext) (void)hipEventRecord(c->prof_events[idx].b, c->stream);
}

// slmFrequencyDrift, lib/slm.cc:36-73 (host side, used to build the offset table)
static float slm_frequency_drift(double V1, double V2, int p1, int p2, float cf, float t) {
  const float cs = 1500.0f;
  const double q1 = V1 * (double)t + (double)p1;
  const double q2 = V2 * (double)t + (double)p2;
  const float sign = (float)(((q1 * V1 + q2 * V2) > 0) * 2 - 1);
  const double num = fabs(V1 * q1 + V2 * q2);
  const double den = sqrt(q1 * q1 + q2 * q2);
  if (den == 0) return 0.0f;
  return (float)((double)(-sign) * num / den * (double)cf / (double)cs);
}

// ---------------------------------------------------------------- options
// Every switch of the implementation is an option of the context: set by uwspr_set_option, or -- for experiments and
// for the ones that shape the context when it is created (the "k1_" / "k3_" ones) -- listed in the one environment
// variable UWSPR_OPTIONS="name=value,name=value".  Nothing else in the library reads the environment.
static const struct { const char *name; int dflt, lo, hi; } kOptions[UWSPR_NOPT] = {
    {"sched", 1, 0, 1},             // 1: fused kernel k6_sched; 0: staged launches
    {"stage_kernels", 1, 0, 1},     // staged form: 0 flat kernel everywhere (the reference form), 1 packed / ring kernels
    {"reuse", 1, 0, 1},             // the hypothesis that repeats the previous stage's winner is not correlated again
    {"phasor_tables", 1, 0, 1},     // lag stages read their phasors from per-slot tables
    {"fast_search", 0, 0, 1},       // stages S0..S4 with fused multiply-adds and shuffle-tree sums (NOT the reference's arithmetic)
    {"k4_t", 0, 0, 4},              // flat kernel: tones per lane (0: by size)
    {"k5_lanes", -1, -1, 1},        // fold: -1 by size, 0 wave form, 1 lanes form
    {"k1_rows", 0, 0, 348},         // spectrogram: rows per wavefront walk (0: default)
    {"k3_tile", -1, -1, 2},         // coarse search: tile form (-1: by size)
    {"k3_pitch", 0, 0, 4096},       // coarse search: tile row pitch (0: default)
    {"sched_stamps", 0, 0, 1},      // diagnostics: phase times of the fused kernel (uwspr_debug_sched_stamps)
    {"sched_grid", 0, 0, 65536},    // fused kernel: workgroups (0: one per CU)
    {"dist_force_comm", 0, 0, 1},   // tests: a one-rank communicator is really created
    {"frontend", 0, 0, 1},          // K0: 0 the flowgraph's GNU Radio chain as one polyphase FIR (grc:303-400,840-956,1767-1808), 1 compact single stage
    {"k4_forms", 1, 0, 255},        // staged form: bit 0 = S5 (k4_ring<6,8>) keeps its sample pairs in a register ring
};

static void refresh_options(uwspr_ctx *c) {
  c->use_stage_kernels = c->opt[UWSPR_OPT_STAGE_KERNELS] != 0;
  c->reuse_centre = c->opt[UWSPR_OPT_REUSE] != 0;
  c->use_ptab = c->opt[UWSPR_OPT_PHASOR_TABLES] != 0;
  c->fast_search = c->opt[UWSPR_OPT_FAST_SEARCH] != 0;
  // the fast variant exists for the staged launches only: derived here, opt[] stays as the caller set it, so
  // fast_search = 0 later brings the fused kernel back and get_option("sched") reports what was asked for
  c->use_fused = c->opt[UWSPR_OPT_SCHED] != 0 && !c->fast_search;
  c->sched_grid = c->opt[UWSPR_OPT_SCHED_GRID];
}

static int option_index(const char *name) {
  if (!name) return -1;
  for (int i = 0; i < UWSPR_NOPT; i++)
    if (!strcmp(name, kOptions[i].name)) return i;
  return -1;
}

extern "C" int uwspr_set_option(uwspr_ctx *c, const char *name, int value) {
  if (!c) return UWSPR_ERR_ARG;
  const int i = option_index(name);
  if (i < 0) return fail(c, UWSPR_ERR_ARG, "unknown option '%s'", name ? name : "(null)");
  if (i == UWSPR_OPT_K1_ROWS || i == UWSPR_OPT_K3_TILE || i == UWSPR_OPT_K3_PITCH)
    return fail(c, UWSPR_ERR_ARG, "option '%s' shapes the context when it is created: UWSPR_OPTIONS=%s=%d", name, name, value);
  if (value < kOptions[i].lo || value > kOptions[i].hi)
    return fail(c, UWSPR_ERR_ARG, "option '%s' = %d: allowed %d .. %d", name, value, kOptions[i].lo, kOptions[i].hi);
  c->opt[i] = value; c->opt_set[i] = true;
  refresh_options(c);
  return UWSPR_OK;
}

extern "C" int uwspr_get_option(uwspr_ctx *c, const char *name, int *value) {
  if (!c || !value) return UWSPR_ERR_ARG;
  const int i = option_index(name);
  if (i < 0) return fail(c, UWSPR_ERR_ARG, "unknown option '%s'", name ? name : "(null)");
  *value = c->opt[i];
  return UWSPR_OK;
}

// UWSPR_OPTIONS="name=value,..."; unknown names are an error of the context being created (a typo must not pass silently)
static int options_from_environment(uwspr_ctx *c) {
  for (int i = 0; i < UWSPR_NOPT; i++) { c->opt[i] = kOptions[i].dflt; c->opt_set[i] = false; }
  const char *e = getenv("UWSPR_OPTIONS");
  if (e) {
    // hand-rolled split (no strtok: contexts are created concurrently, one block per thread); every value goes through
    // strtol with an end check and the option's range, an over-long string is an error, not a truncation
    char buf[512];
    const size_t len = strlen(e);
    if (len >= sizeof(buf)) return fail(c, UWSPR_ERR_ARG, "UWSPR_OPTIONS: %zu characters (at most %zu)", len, sizeof(buf) - 1);
    memcpy(buf, e, len + 1);
    char *tok = buf;
    while (*tok) {
      while (*tok == ',' || *tok == ' ') tok++;
      if (!*tok) break;
      char *end = tok;
      while (*end && *end != ',' && *end != ' ') end++;
      const bool last = *end == 0;
      *end = 0;
      char *eq = strchr(tok, '=');
      if (!eq) return fail(c, UWSPR_ERR_ARG, "UWSPR_OPTIONS: '%s' is not name=value", tok);
      *eq = 0;
      const int i = option_index(tok);
      if (i < 0) return fail(c, UWSPR_ERR_ARG, "UWSPR_OPTIONS: unknown option '%s'", tok);
      char *vend = nullptr;
      errno = 0;
      const long v = strtol(eq + 1, &vend, 10);
      if (vend == eq + 1 || *vend || errno || v < kOptions[i].lo || v > kOptions[i].hi)
        return fail(c, UWSPR_ERR_ARG, "UWSPR_OPTIONS: %s='%s' is not an integer in %d .. %d", tok, eq + 1, kOptions[i].lo, kOptions[i].hi);
      c->opt[i] = (int)v; c->opt_set[i] = true;
      if (last) break;
      tok = end + 1;
    }
  }
  refresh_options(c);
  return UWSPR_OK;
}

// ------------------------------------------------------------- ctx create
extern "C" int uwspr_ctx_create(const uwspr_params *p, int device, uwspr_ctx **out) {
  if (!p || !out) return UWSPR_ERR_ARG;
  *out = nullptr;
  uwspr_ctx *c = new (std::nothrow) uwspr_ctx();
  if (!c) return UWSPR_ERR_NOMEM;
  memset(c->err, 0, sizeof(c->err));
  c->p = *p;
  c->device = device;
  c->own_stream = c->stream = nullptr;
  c->d_window = c->d_twiddle = nullptr; c->d_off = nullptr; c->d_umap = nullptr; c->d_k3_tile = nullptr;
  c->d_fe_taps = nullptr; c->fe_mode = -1; c->fe_J = 0; c->fe_dcols = 0; c->cap_audio = 0; c->d_audio = nullptr;
  c->cap_frames_bytes = 0; c->d_frames = nullptr; c->cap_B = 0;
  c->d_ps = c->d_psavg = c->d_smraw = c->d_smspec = c->d_noise = nullptr;
  c->d_cands = nullptr; c->d_npk = nullptr; c->d_work = nullptr; c->last_B = 0; c->num_cus = 256;
  c->grid_cap = 0; c->cap_grid_bytes = 0; c->d_syncgrid = nullptr;
  c->cap_hyps = 0; c->d_hyps = nullptr; c->cap_grps = 0; c->d_grps = nullptr;
  c->cap_cent = 0; c->d_cent = nullptr; c->d_cent_frame = nullptr; c->cap_abi_hyps = 0; c->d_abi_hyps = nullptr;
  c->cap_p = 0; c->d_p = nullptr; c->cap_sync = 0; c->d_sync = nullptr;
  c->cap_sym = 0; c->d_sym = nullptr; c->cap_state = 0; c->d_state = nullptr;
  c->cap_dout = 0; c->d_dout = nullptr; c->prof_mask = 0;
  c->cur_cands = nullptr; c->cur_npk = nullptr; c->cur_dout = nullptr; c->last_per_frame = 0;
  c->cands_from_fdr = false; c->fast_now = false;
  c->cap_slab = 0; c->d_slab = nullptr;
  c->dist_comm = nullptr; c->dist_rank = 0; c->dist_world = 0;
  c->cap_tmpc = 0; c->d_tmpc = nullptr; c->cap_tmpn = 0; c->d_tmpn = nullptr;
  c->h_pin = nullptr; c->pin_busy[0] = c->pin_busy[1] = false;
  c->d_stream_frames = nullptr; c->cap_stream_frames = 0; c->ring_ev = nullptr;
  c->fstride = p->fl; c->np = p->fl < 45000 ? p->fl : 45000;   // sync_and_demodulate_impl.cc:92
  c->cap_ptab = 0; c->d_ptab = nullptr;
  c->ntries = UWSPR_NJIG; c->cap_pwin = 0; c->d_pwin = nullptr; c->cap_need = 0; c->d_need = nullptr;
  c->last_slots = 0; c->last_sched_B = 0; c->last_sched_per_frame = 0; c->last_sched_lazy = false; c->last_sched_out = nullptr;
  c->cap_tabs = 0; c->d_tabs = nullptr; c->d_counter = nullptr; c->d_sched_stamps = nullptr; c->cap_sched_stamps = 0;
  *out = c;  // handed back even on failure so uwspr_last_error() can be read
  {
    const int orc = options_from_environment(c);
    if (orc) return orc;
  }

  fdr_consts &f = c->fc;
  memset(&f, 0, sizeof(f));
  // FDR_impl.cc:81-141
  const int size = 2 * p->spb;
  const int maxfreq = (int)((float)p->fs / 2.0);
  if (p->halfbandwidth > maxfreq)
    return fail(c, UWSPR_ERR_PARAM, "Half pass bandwidth (%d) must be lower than max freq range (%d)",
                p->halfbandwidth, maxfreq);
  if (p->spb != 256) return fail(c, UWSPR_ERR_UNSUPPORTED, "spb=%d: this build implements the 512-point transform (spb=256)", p->spb);
  if (p->fs <= 0 || p->fl <= 0 || p->maxdrift < 0 || p->maxfreqs < 1 || p->halfbandwidth < 1 || p->cf <= 0)
    return fail(c, UWSPR_ERR_ARG, "non-positive constructor argument");
  f.fl = p->fl; f.size = size; f.m = size / 2; f.maxfreqs = p->maxfreqs; f.maxdrift = p->maxdrift;
  f.df = (float)p->fs / (float)size;
  f.hpbm = (int)ceil((float)(int)(float)p->halfbandwidth / f.df);
  f.n = (int)(floor(((float)p->fl / (float)p->spb) * 2.0) - 3);
  f.finpb = 2 * f.hpbm;
  f.noiseidx = (int)floor(0.3 * (float)f.finpb);
  f.min_snr = (float)pow(10.0, -7.0 / 10.0);
  f.min_snr_floor = (float)(0.1 * (double)f.min_snr);
  f.threshold = (float)p->threshold;
  if (f.n < 2 * (UWSPR_NSYM - 1) + UWSPR_NK0 || (f.n - 1) * (p->spb / 2) + size > p->fl)
    return fail(c, UWSPR_ERR_UNSUPPORTED, "fl=%d too short for 162 symbols + 26 half-symbol offsets", p->fl);
  if (f.finpb > 512 || f.finpb < 3) return fail(c, UWSPR_ERR_RANGE, "pass band of %d bins", f.finpb);

  // ---- offset table: ifd - ifr for every (ifr, hypothesis, symbol) --------
  f.nlin = 2 * p->maxdrift + 1;
  f.cell_hyps = f.nlin + UWSPR_NSLM;
  f.ntot = UWSPR_NIFR * UWSPR_NK0 * f.cell_hyps;
  f.ifr_lo = f.m - f.hpbm + 1 - 2;
  const int ifr_hi = f.m + f.hpbm - 2 + 2;
  f.n_ifr = ifr_hi - f.ifr_lo + 1;
  // SLM drift in bins is independent of ifr; cache slmFrequencyDrift per (instance, t)
  std::vector<float> slm((size_t)UWSPR_NSLM * 111);
  for (int s = 0; s < UWSPR_NSLM; s++) {
    const double V1 = (double)((s / 5) % 5) - 2.0, V2 = (double)(s / 25) - 2.0;  // slm.cc:103-109
    const int p2 = 50 + 200 * (s % 5);
    for (int t = 0; t < 111; t++)
      slm[(size_t)s * 111 + t] = slm_frequency_drift(V1, V2, 0, p2, (float)p->cf, (float)t);
  }
  std::vector<int8_t> off((size_t)f.n_ifr * f.cell_hyps * 164, 0);
  int omin = 1 << 30, omax = -(1 << 30);
  for (int r = 0; r < f.n_ifr; r++) {
    const int ifr = f.ifr_lo + r;
    for (int h = 0; h < f.cell_hyps; h++) {
      for (int k = 0; k < UWSPR_NSYM; k++) {
        int ifd;
        if (h < f.nlin) {
          const int drift = h - p->maxdrift;
          // FDR_impl.cc:353 (binary64 expression, truncated)
          ifd = (int)((double)ifr + ((double)(float)k - 81.0) / 81.0 * (double)(float)drift /
                                        (2.0 * (double)f.df));
        } else {
          // FDR_impl.cc:382-385: t = k*111/162 (integer division), binary32 add, truncated
          const int t = k * 111 / 162;
          ifd = (int)((float)ifr + slm[(size_t)(h - f.nlin) * 111 + t] / f.df);
        }
        const int o = ifd - ifr;
        if (o < -128 || o > 127) return fail(c, UWSPR_ERR_RANGE, "search reach %d bins", o);
        omin = std::min(omin, o); omax = std::max(omax, o);
        off[((size_t)r * f.cell_hyps + h) * 164 + k] = (int8_t)o;
      }
    }
  }
  f.off_min = omin; f.off_max = omax; f.nc = UWSPR_NIFR + (omax - omin);
  const int lo = std::min(f.m - f.hpbm - 3, f.ifr_lo + omin - 3);
  const int hi = std::max(f.m + f.hpbm - 1 + 3, ifr_hi + omax + 3);
  if (lo < 0 || hi > size - 1)
    return fail(c, UWSPR_ERR_RANGE,
                "halfbandwidth=%d: pass band + search reach needs columns %d..%d of 0..%d "
                "(the reference reads out of bounds here)", p->halfbandwidth, lo, hi, size - 1);
  f.band_lo = lo; f.band_w = hi - lo + 1;
  f.cand_slots = std::min(p->maxfreqs, std::max(1, (f.finpb - 1) / 2));

  // ---- device -------------------------------------------------------------
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0)
    return fail(c, UWSPR_ERR_NODEVICE, "no HIP device visible; this library has no CPU fallback");
  if (device < 0 || device >= ndev) return fail(c, UWSPR_ERR_ARG, "device %d of %d", device, ndev);
  HIPCHK(c, hipSetDevice(device));
  hipDeviceProp_t prop;
  HIPCHK(c, hipGetDeviceProperties(&prop, device));
  snprintf(c->device_name, sizeof(c->device_name), "%s", prop.gcnArchName);
  c->num_cus = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
  if (strncmp(prop.gcnArchName, "gfx950", 6) != 0)
    return fail(c, UWSPR_ERR_NODEVICE, "device %d is %s; kernels are built for gfx950 only", device,
                prop.gcnArchName);
  HIPCHK(c, hipStreamCreateWithFlags(&c->own_stream, hipStreamNonBlocking));
  c->stream = c->own_stream;

  // window (FDR_impl.cc:101-105) and twiddles (exp(-2*pi*i*k/512), k<256; k=0,128 exact)
  std::vector<float> w(size), tw(size);
  for (int i = 0; i < size; i++) w[i] = (float)sin((M_PI / (size - 1)) * i);
  for (int k = 0; k < size / 2; k++) {
    const double ang = 2.0 * M_PI * (double)k / 512.0;
    tw[2 * k] = (float)cos(ang);
    tw[2 * k + 1] = (float)(-sin(ang));
  }
  tw[0] = 1.0f; tw[1] = 0.0f; tw[256] = 0.0f; tw[257] = -1.0f;
  // distinct offset sequences per ifr row (identical sequences give identical metrics)
  std::vector<std::vector<int>> uniq_of(f.n_ifr);           // first hypothesis of each distinct sequence
  std::vector<uint16_t> umap((size_t)f.n_ifr * f.cell_hyps);
  f.umax = 0;
  for (int r = 0; r < f.n_ifr; r++) {
    for (int h = 0; h < f.cell_hyps; h++) {
      const int8_t *sh = &off[((size_t)r * f.cell_hyps + h) * 164];
      int u = -1;
      for (size_t q = 0; q < uniq_of[r].size(); q++)
        if (memcmp(sh, &off[((size_t)r * f.cell_hyps + uniq_of[r][q]) * 164], UWSPR_NSYM) == 0) { u = (int)q; break; }
      if (u < 0) { u = (int)uniq_of[r].size(); uniq_of[r].push_back(h); }
      umap[(size_t)r * f.cell_hyps + h] = (uint16_t)u;
    }
    f.umax = std::max(f.umax, (int)uniq_of[r].size());
  }
  // [ifr][u][84] words of 2 x u16 tile byte offsets (coarse_seq_entry); rows with fewer distinct
  // sequences repeat sequence 0
  const size_t k3_scratch = coarse_plan(f, c->opt[UWSPR_OPT_K3_PITCH], c->opt[UWSPR_OPT_K3_TILE]);   // tile form + pitch (k3_coarse.hip)
  const int sw = coarse_seq_words();
  std::vector<uint32_t> offw((size_t)f.n_ifr * f.umax * sw, 0u);
  for (int r = 0; r < f.n_ifr; r++)
    for (int u = 0; u < f.umax; u++) {
      const int h = u < (int)uniq_of[r].size() ? uniq_of[r][u] : uniq_of[r][0];
      for (int k = 0; k < UWSPR_NSYM; k++) {
        const uint32_t e = coarse_seq_entry(f, k, off[((size_t)r * f.cell_hyps + h) * 164 + k]);
        if (e > 0xffffu) return fail(c, UWSPR_ERR_UNSUPPORTED, "coarse tile of %d centre columns", f.nc);
        offw[((size_t)r * f.umax + u) * sw + k / 2] |= e << (16 * (k & 1));
      }
    }
  if (coarse_lds_bytes(f) > 160 * 1024)
    return fail(c, UWSPR_ERR_UNSUPPORTED, "coarse search needs %zu B of LDS (> 160 KiB): reduce maxdrift/cf",
                coarse_lds_bytes(f));
  if (k3_scratch) HIPCHK(c, hipMalloc((void **)&c->d_k3_tile, k3_scratch * sizeof(float) * (size_t)c->num_cus));
  HIPCHK(c, hipMalloc((void **)&c->d_umap, umap.size() * sizeof(uint16_t)));
  HIPCHK(c, hipMemcpy(c->d_umap, umap.data(), umap.size() * sizeof(uint16_t), hipMemcpyHostToDevice));
  HIPCHK(c, hipMalloc((void **)&c->d_window, size * sizeof(float)));
  HIPCHK(c, hipMalloc((void **)&c->d_twiddle, size * sizeof(float)));
  HIPCHK(c, hipMalloc((void **)&c->d_off, offw.size() * sizeof(uint32_t)));
  HIPCHK(c, hipMemcpy(c->d_window, w.data(), size * sizeof(float), hipMemcpyHostToDevice));
  HIPCHK(c, hipMemcpy(c->d_twiddle, tw.data(), size * sizeof(float), hipMemcpyHostToDevice));
  HIPCHK(c, hipMemcpy(c->d_off, offw.data(), offw.size() * sizeof(uint32_t), hipMemcpyHostToDevice));
  if (coarse_configure(f) != 0) return fail(c, UWSPR_ERR_HIP, "cannot reserve %zu B of LDS for the coarse search", coarse_lds_bytes(f));
  return UWSPR_OK;
}

extern "C" void uwspr_ctx_destroy(uwspr_ctx *c) {
  if (!c) return;
  (void)uwspr_dist_finalize(c);
  if (c->own_stream) { (void)hipStreamSynchronize(c->own_stream); }
  void *bufs[] = {c->d_window, c->d_twiddle, c->d_k3_tile, c->d_off, c->d_umap, c->d_fe_taps, c->d_audio, c->d_frames, c->d_ps, c->d_psavg, c->d_smraw,
                  c->d_smspec, c->d_noise, c->d_cands, c->d_npk, c->d_work, c->d_syncgrid, c->d_hyps, c->d_grps, c->d_cent,
                  c->d_abi_hyps, c->d_p, c->d_sync, c->d_sym, c->d_state, c->d_dout, c->d_slab, c->d_tabs, c->d_counter, c->d_sched_stamps, c->d_pwin, c->d_ptab, c->d_need, c->d_stream_frames, c->d_tmpc, c->d_tmpn};
  for (void *b : bufs) if (b) (void)hipFree(b);
  c->ring.close();
  if (c->ring_ev) (void)hipEventDestroy(c->ring_ev);
  if (c->h_pin) { (void)hipHostFree(c->h_pin); (void)hipEventDestroy(c->pin_ev[0]); (void)hipEventDestroy(c->pin_ev[1]); }
  for (auto &e : c->prof_events) { (void)hipEventDestroy(e.a); (void)hipEventDestroy(e.b); }
  for (auto &e : c->ev_pool) (void)hipEventDestroy(e);
  if (c->own_stream) (void)hipStreamDestroy(c->own_stream);
  delete c;
}

extern "C" const char *uwspr_last_error(const uwspr_ctx *c) { return c ? c->err : "null context"; }

extern "C" const char *uwspr_status_string(int s) {
  switch (s) {
    case UWSPR_OK: return "ok";
    case UWSPR_ERR_PARAM: return "half pass bandwidth above fs/2";
    case UWSPR_ERR_RANGE: return "pass band plus search reach outside the spectrum";
    case UWSPR_ERR_UNSUPPORTED: return "unsupported geometry";
    case UWSPR_ERR_HIP: return "HIP runtime error";
    case UWSPR_ERR_NOMEM: return "out of memory";
    case UWSPR_ERR_ARG: return "bad argument";
    case UWSPR_ERR_NODEVICE: return "no gfx950 device (no CPU fallback)";
    default: return "unknown status";
  }
}

extern "C" int uwspr_get_info(const uwspr_ctx *c, uwspr_info *o) {
  if (!c || !o) return UWSPR_ERR_ARG;
  const fdr_consts &f = c->fc;
  memset(o, 0, sizeof(*o));
  o->abi_version = UWSPR_ABI_VERSION;
  o->size = f.size; o->m = f.m; o->hpbm = f.hpbm; o->n = f.n; o->finpb = f.finpb;
  o->noiseidx = f.noiseidx; o->df = f.df; o->min_snr = f.min_snr;
  o->band_lo = f.band_lo; o->band_w = f.band_w; o->cell_hyps = f.cell_hyps;
  o->off_min = f.off_min; o->off_max = f.off_max; o->device = c->device;
  snprintf(o->device_name, sizeof(o->device_name), "%s", c->device_name);
  return UWSPR_OK;
}

static int ready(uwspr_ctx *c) {
  if (!c) return UWSPR_ERR_ARG;
  if (!c->own_stream) return fail(c, UWSPR_ERR_NODEVICE, "context has no device (creation failed: %s)", c->err);
  hipError_t e = hipSetDevice(c->device);
  if (e != hipSuccess) return fail(c, UWSPR_ERR_HIP, "hipSetDevice: %s", hipGetErrorString(e));
  return UWSPR_OK;
}

extern "C" int uwspr_set_stream(uwspr_ctx *c, void *s) {
  int rc = ready(c);
  if (rc) return rc;
  c->stream = s ? (hipStream_t)s : c->own_stream;
  return UWSPR_OK;
}

extern "C" int uwspr_synchronize(uwspr_ctx *c) {
  int rc = ready(c);
  if (rc) return rc;
  HIPCHK(c, hipStreamSynchronize(c->stream));
  return UWSPR_OK;
}

// Host memory -> device.  Small pieces (a stream push: a few MB) go through two pinned staging halves,
// the copy of piece k+1 into pinned memory overlapping the DMA of piece k, and return with the
// transfer merely enqueued; whole batches of frames are left to the runtime's pageable path.
static int upload(uwspr_ctx *c, void *dst, const void *src, size_t bytes) {
  const size_t PIECE = 8u << 20;
  {   // the caller's buffer is already page-locked (uwspr_host_alloc, hipHostMalloc, hipHostRegister): plain DMA
    hipPointerAttribute_t at;
    if (hipPointerGetAttributes(&at, src) == hipSuccess && at.type == hipMemoryTypeHost) {
      HIPCHK(c, hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, c->stream));
      return UWSPR_OK;
    }
    (void)hipGetLastError();   // an ordinary pointer is "invalid value" to the query: not an error here
  }
  if (bytes > 4 * PIECE) {   // a whole batch of frames: one host thread's memcpy into pinned memory (~10 GB/s)
    HIPCHK(c, hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, c->stream));   // would be slower than the runtime's own staging (measured 57 k vs 87 k frames/s)
    return UWSPR_OK;
  }
  if (!c->h_pin) {
    if (hipHostMalloc((void **)&c->h_pin, 2 * PIECE, hipHostMallocDefault) != hipSuccess) {
      c->h_pin = nullptr;
      (void)hipGetLastError();
      HIPCHK(c, hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, c->stream));
      return UWSPR_OK;
    }
    HIPCHK(c, hipEventCreateWithFlags(&c->pin_ev[0], hipEventDisableTiming));
    HIPCHK(c, hipEventCreateWithFlags(&c->pin_ev[1], hipEventDisableTiming));
    c->pin_busy[0] = c->pin_busy[1] = false;
  }
  size_t off = 0;
  int k = 0;
  while (off < bytes) {
    const size_t n = bytes - off < PIECE ? bytes - off : PIECE;
    char *half = (char *)c->h_pin + (size_t)(k & 1) * PIECE;
    if (c->pin_busy[k & 1]) HIPCHK(c, hipEventSynchronize(c->pin_ev[k & 1]));   // its previous DMA is done
    memcpy(half, (const char *)src + off, n);
    HIPCHK(c, hipMemcpyAsync((char *)dst + off, half, n, hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, hipEventRecord(c->pin_ev[k & 1], c->stream));
    c->pin_busy[k & 1] = true;
    off += n;
    k++;
  }
  return UWSPR_OK;
}

// frames -> device pointer (staged when they are host memory)
static int frames_on_device(uwspr_ctx *c, const float *frames, int B, int where, const float **dev) {
  if (!frames || B <= 0) return fail(c, UWSPR_ERR_ARG, "frames=%p B=%d", (const void *)frames, B);
  if (where == UWSPR_DEVICE || where == UWSPR_DEVICE_FRAMES) { *dev = frames; return UWSPR_OK; }
  if (where != UWSPR_HOST) return fail(c, UWSPR_ERR_ARG, "where=%d", where);
  // frame b starts at frames + 2 * fstride * b: what is uploaded is the span the B frames cover
  const size_t bytes = ((size_t)(B - 1) * c->fstride + c->fc.fl) * 2 * sizeof(float);
  size_t cap = c->cap_frames_bytes / sizeof(float);
  int rc = ensure(c, &c->d_frames, &cap, bytes / sizeof(float));
  c->cap_frames_bytes = cap * sizeof(float);
  if (rc) return rc;
  if ((rc = upload(c, c->d_frames, frames, bytes))) return rc;
  *dev = c->d_frames;
  return UWSPR_OK;
}

static int copy_out(uwspr_ctx *c, void *dst, const void *src, size_t bytes, int where) {
  if (!dst || bytes == 0) return UWSPR_OK;
  HIPCHK(c, hipMemcpyAsync(dst, src, bytes, where != UWSPR_DEVICE ? hipMemcpyDeviceToHost : hipMemcpyDeviceToDevice, c->stream));
  return UWSPR_OK;
}

// -------------------------------------------------------------- front-end
extern "C" int uwspr_frontend_batch(uwspr_ctx *c, const float *audio, int B, int nin, int where,
                                    float *frames_out) {
  int rc = ready(c);
  if (rc) return rc;
  const int nout = c->fc.fl;
  if (!audio || !frames_out || B <= 0 || nin <= 0) return fail(c, UWSPR_ERR_ARG, "uwspr_frontend_batch: audio/out/B/nin");
  if (c->p.fs != 375) return fail(c, UWSPR_ERR_UNSUPPORTED, "front-end is 12000 -> 375 S/s (fs=%d)", c->p.fs);
  if (!c->d_fe_taps || c->fe_mode != c->opt[UWSPR_OPT_FRONTEND]) {
    std::vector<float> img;
    int J = 0, dcols = 0;
    if (frontend_tap_image(c->opt[UWSPR_OPT_FRONTEND], img, &J, &dcols)) return fail(c, UWSPR_ERR_ARG, "front-end mode %d", c->opt[UWSPR_OPT_FRONTEND]);
    if (frontend_prepare()) return fail(c, UWSPR_ERR_HIP, "front-end kernel: %zu bytes of LDS refused", (size_t)160 * 1024);
    HIPCHK(c, hipStreamSynchronize(c->stream));          // (a launch with the other mode's taps may be in flight)
    if (c->d_fe_taps) { HIPCHK(c, hipFree(c->d_fe_taps)); c->d_fe_taps = nullptr; }
    HIPCHK(c, hipMalloc((void **)&c->d_fe_taps, img.size() * sizeof(float)));
    HIPCHK(c, hipMemcpy(c->d_fe_taps, img.data(), img.size() * sizeof(float), hipMemcpyHostToDevice));
    c->fe_mode = c->opt[UWSPR_OPT_FRONTEND]; c->fe_J = J; c->fe_dcols = dcols;
  }
  const float *da = audio;
  float *dout = frames_out;
  if (where == UWSPR_HOST) {
    if ((rc = ensure(c, &c->d_audio, &c->cap_audio, (size_t)B * nin))) return rc;
    size_t cap = c->cap_frames_bytes / sizeof(float);
    rc = ensure(c, &c->d_frames, &cap, (size_t)B * nout * 2);
    c->cap_frames_bytes = cap * sizeof(float);
    if (rc) return rc;
    HIPCHK(c, hipMemcpyAsync(c->d_audio, audio, (size_t)B * nin * sizeof(float), hipMemcpyHostToDevice, c->stream));
    da = c->d_audio; dout = c->d_frames;
  } else if (where != UWSPR_DEVICE) {
    return fail(c, UWSPR_ERR_ARG, "where=%d", where);
  }
  launch_frontend(c, da, B, nin, (float2 *)dout, nout);
  HIPCHK(c, hipGetLastError());
  if (where == UWSPR_HOST) {
    HIPCHK(c, hipMemcpyAsync(frames_out, c->d_frames, (size_t)B * nout * 2 * sizeof(float), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
  }
  return UWSPR_OK;
}

extern "C" int uwspr_frontend_design(int mode, int stage, double *taps, int cap, int *delay) {
  std::vector<double> g;
  int d = 0;
  const int rc = frontend_design(mode, stage, g, &d);
  if (rc) return rc;
  if (delay) *delay = d;
  const int n = stage == 0 ? (int)g.size() / 2 : (int)g.size();     // complex pairs / real taps
  if (taps) {
    const int k = (stage == 0 ? 2 : 1) * std::min(n, cap < 0 ? 0 : cap);
    for (int i = 0; i < k; i++) taps[i] = g[i];
  }
  return n;
}

// -------------------------------------------------------------------- FDR
static int ensure_fdr(uwspr_ctx *c, int B) {
  const fdr_consts &f = c->fc;
  if (B <= c->cap_B) return UWSPR_OK;
  size_t cap;
  int rc;
  c->cap_B = 0;   // a failed allocation below leaves no capacity behind that the null buffers cannot back
#define GROW(ptr, elems) cap = 0; if (ptr) { HIPCHK(c, hipStreamSynchronize(c->stream)); HIPCHK(c, hipFree(ptr)); ptr = nullptr; } \
  rc = ensure(c, &ptr, &cap, (size_t)(elems)); if (rc) return rc;
  GROW(c->d_ps, (size_t)B * f.n * f.band_w);
  GROW(c->d_psavg, (size_t)B * f.band_w);
  GROW(c->d_smraw, (size_t)B * f.finpb);
  GROW(c->d_smspec, (size_t)B * f.finpb);
  GROW(c->d_noise, (size_t)B);
  GROW(c->d_cands, (size_t)B * f.maxfreqs);
  GROW(c->d_npk, (size_t)B);
  GROW(c->d_work, (size_t)B * f.cand_slots + 1);
#undef GROW
  c->cap_B = B;
  return UWSPR_OK;
}

static int run_fdr(uwspr_ctx *c, const float *dframes, int B, uwspr_candidate *user_cands = nullptr,
                   int32_t *user_npk = nullptr) {
  int rc = ensure_fdr(c, B);
  if (rc) return rc;
  c->cur_cands = user_cands ? user_cands : c->d_cands;
  c->cur_npk = user_npk ? user_npk : c->d_npk;
  if (c->grid_cap > 0) {
    size_t cap = c->cap_grid_bytes / sizeof(float);
    rc = ensure(c, &c->d_syncgrid, &cap, (size_t)B * c->grid_cap * c->fc.ntot);
    c->cap_grid_bytes = cap * sizeof(float);
    if (rc) return rc;
  }
  launch_spectrogram(c, dframes, B);
  launch_spectrum(c, B);
  launch_coarse(c, B);
  HIPCHK(c, hipGetLastError());
  c->last_B = B;
  return UWSPR_OK;
}

extern "C" int uwspr_fdr_batch(uwspr_ctx *c, const float *frames, int B, int where,
                               uwspr_candidate *cands, int32_t *npk) {
  int rc = ready(c);
  if (rc) return rc;
  const float *d;
  if ((rc = frames_on_device(c, frames, B, where, &d))) return rc;
  const bool direct = where == UWSPR_DEVICE && cands && npk;
  if ((rc = run_fdr(c, d, B, direct ? cands : nullptr, direct ? npk : nullptr))) return rc;
  if (!direct) {
    if ((rc = copy_out(c, cands, c->d_cands, (size_t)B * c->fc.maxfreqs * sizeof(uwspr_candidate), where))) return rc;
    if ((rc = copy_out(c, npk, c->d_npk, (size_t)B * sizeof(int32_t), where))) return rc;
  }
  if (where != UWSPR_DEVICE) HIPCHK(c, hipStreamSynchronize(c->stream));
  return UWSPR_OK;
}

extern "C" int uwspr_fdr_read_spectrum(uwspr_ctx *c, int B, float *ps_band, float *psavg, float *smraw,
                                       float *smspec, float *noise) {
  int rc = ready(c);
  if (rc) return rc;
  if (B <= 0 || B > c->last_B) return fail(c, UWSPR_ERR_ARG, "B=%d but the last FDR batch held %d frames", B, c->last_B);
  const fdr_consts &f = c->fc;
  if ((rc = copy_out(c, ps_band, c->d_ps, (size_t)B * f.n * f.band_w * 4, UWSPR_HOST))) return rc;
  if ((rc = copy_out(c, psavg, c->d_psavg, (size_t)B * f.band_w * 4, UWSPR_HOST))) return rc;
  if ((rc = copy_out(c, smraw, c->d_smraw, (size_t)B * f.finpb * 4, UWSPR_HOST))) return rc;
  if ((rc = copy_out(c, smspec, c->d_smspec, (size_t)B * f.finpb * 4, UWSPR_HOST))) return rc;
  if ((rc = copy_out(c, noise, c->d_noise, (size_t)B * 4, UWSPR_HOST))) return rc;
  HIPCHK(c, hipStreamSynchronize(c->stream));
  return UWSPR_OK;
}

extern "C" int uwspr_fdr_keep_syncgrid(uwspr_ctx *c, int ncand_cap) {
  int rc = ready(c);
  if (rc) return rc;
  if (ncand_cap < 0) return fail(c, UWSPR_ERR_ARG, "ncand_cap=%d", ncand_cap);
  c->grid_cap = std::min(ncand_cap, c->fc.cand_slots);
  if (c->grid_cap == 0 && c->d_syncgrid) {
    HIPCHK(c, hipStreamSynchronize(c->stream));
    HIPCHK(c, hipFree(c->d_syncgrid));
    c->d_syncgrid = nullptr; c->cap_grid_bytes = 0;
  }
  return UWSPR_OK;
}

extern "C" int uwspr_fdr_read_syncgrid(uwspr_ctx *c, int B, float *grid) {
  int rc = ready(c);
  if (rc) return rc;
  if (!c->d_syncgrid || c->grid_cap <= 0) return fail(c, UWSPR_ERR_ARG, "sync grid not kept: call uwspr_fdr_keep_syncgrid first");
  if (B <= 0 || B > c->last_B) return fail(c, UWSPR_ERR_ARG, "B=%d but the last FDR batch held %d frames", B, c->last_B);
  if ((rc = copy_out(c, grid, c->d_syncgrid, (size_t)B * c->grid_cap * c->fc.ntot * 4, UWSPR_HOST))) return rc;
  HIPCHK(c, hipStreamSynchronize(c->stream));
  return UWSPR_OK;
}

// ------------------------------------------------------------- fine sweep
static int ensure_sweep(uwspr_ctx *c, size_t H, bool soft) {
  int rc;
  if ((rc = ensure(c, &c->d_p, &c->cap_p, H * UWSPR_NSYM))) return rc;
  if ((rc = ensure(c, &c->d_sync, &c->cap_sync, H))) return rc;
  if (soft && (rc = ensure(c, &c->d_sym, &c->cap_sym, H * UWSPR_NSYM))) return rc;
  return UWSPR_OK;
}

extern "C" int uwspr_sync_sweep(uwspr_ctx *c, const float *frames, int B, const uwspr_hyp *hyps, int H,
                                int where, float *sync, uint8_t *symbols) {
  int rc = ready(c);
  if (rc) return rc;
  if (!hyps || H <= 0) return fail(c, UWSPR_ERR_ARG, "hyps=%p H=%d", (const void *)hyps, H);
  if (where == UWSPR_DEVICE_FRAMES) return fail(c, UWSPR_ERR_ARG, "uwspr_sync_sweep: where is UWSPR_HOST or UWSPR_DEVICE");
  const float *d;
  if ((rc = frames_on_device(c, frames, B, where, &d))) return rc;
  if ((rc = ensure(c, &c->d_hyps, &c->cap_hyps, (size_t)H))) return rc;
  const uwspr_hyp *dabi = hyps;
  if (where == UWSPR_HOST) {
    for (int h = 0; h < H; h++)
      if (hyps[h].frame >= B) return fail(c, UWSPR_ERR_ARG, "hyps[%d].frame=%d >= B=%d", h, hyps[h].frame, B);
    if ((rc = ensure(c, &c->d_abi_hyps, &c->cap_abi_hyps, (size_t)H))) return rc;
    HIPCHK(c, hipMemcpyAsync(c->d_abi_hyps, hyps, (size_t)H * sizeof(uwspr_hyp), hipMemcpyHostToDevice, c->stream));
    dabi = c->d_abi_hyps;
  }
  const bool soft = symbols != nullptr;
  if ((rc = ensure_sweep(c, (size_t)H, soft))) return rc;
  launch_prep_hyps(c, dabi, c->d_hyps, H);
  launch_tonecorr(c, d, B, c->d_hyps, H, c->d_p);
  launch_fold(c, c->d_hyps, c->d_p, H, c->d_sync, soft ? c->d_sym : nullptr);
  HIPCHK(c, hipGetLastError());
  if ((rc = copy_out(c, sync, c->d_sync, (size_t)H * 4, where))) return rc;
  if (soft && (rc = copy_out(c, symbols, c->d_sym, (size_t)H * UWSPR_NSYM, where))) return rc;
  if (where == UWSPR_HOST) HIPCHK(c, hipStreamSynchronize(c->stream));
  return UWSPR_OK;
}

extern "C" int uwspr_sync_grid(uwspr_ctx *c, const float *frames, int B, int where,
                               const uwspr_candidate *centres, int nf, const float *df, int ndrift,
                               const float *ddrift, int nlag, const int32_t *dlag, float *sync,
                               uint8_t *symbols) {
  int rc = ready(c);
  if (rc) return rc;
  if (!centres || !df || !ddrift || !dlag || nf < 1 || ndrift < 1 || nlag < 1 || nf > 32 || ndrift > 32 || nlag > 4096)
    return fail(c, UWSPR_ERR_ARG, "uwspr_sync_grid: nf=%d ndrift=%d nlag=%d (nf, ndrift <= 32)", nf, ndrift, nlag);
  if (where == UWSPR_DEVICE_FRAMES) return fail(c, UWSPR_ERR_ARG, "uwspr_sync_grid: where is UWSPR_HOST or UWSPR_DEVICE");
  const float *d;
  if ((rc = frames_on_device(c, frames, B, where, &d))) return rc;
  const long long H = (long long)B * nf * ndrift * nlag;
  if (H > (1LL << 27)) return fail(c, UWSPR_ERR_ARG, "uwspr_sync_grid: %lld hypotheses in one call", H);
  const bool soft = symbols != nullptr;
  if ((rc = ensure(c, &c->d_hyps, &c->cap_hyps, (size_t)H))) return rc;
  if ((rc = ensure_sweep(c, (size_t)H, soft))) return rc;
  // centres and the lag offsets on the device
  size_t need_abi = ((size_t)B * sizeof(uwspr_candidate) + (size_t)nlag * sizeof(int32_t) + sizeof(uwspr_hyp) - 1) / sizeof(uwspr_hyp) + 1;
  if ((rc = ensure(c, &c->d_abi_hyps, &c->cap_abi_hyps, need_abi))) return rc;
  uwspr_candidate *dcent = reinterpret_cast<uwspr_candidate *>(c->d_abi_hyps);
  int32_t *ddl = reinterpret_cast<int32_t *>(dcent + B);
  HIPCHK(c, hipMemcpyAsync(dcent, centres, (size_t)B * sizeof(uwspr_candidate),
                           where == UWSPR_HOST ? hipMemcpyHostToDevice : hipMemcpyDeviceToDevice, c->stream));
  HIPCHK(c, hipMemcpyAsync(ddl, dlag, (size_t)nlag * sizeof(int32_t), hipMemcpyHostToDevice, c->stream));
  if (!launch_tonecorr_grid(c, d, B, dcent, nf, df, ndrift, ddrift, nlag, dlag, ddl, c->d_hyps, c->d_p)) {
    // window scheme does not fit (huge lag span): the flat kernel on the same hypothesis list
    launch_tonecorr(c, d, B, c->d_hyps, (int)H, c->d_p);
  }
  launch_fold(c, c->d_hyps, c->d_p, (int)H, c->d_sync, soft ? c->d_sym : nullptr);
  HIPCHK(c, hipGetLastError());
  if ((rc = copy_out(c, sync, c->d_sync, (size_t)H * 4, where))) return rc;
  if (soft && (rc = copy_out(c, symbols, c->d_sym, (size_t)H * UWSPR_NSYM, where))) return rc;
  if (where == UWSPR_HOST) HIPCHK(c, hipStreamSynchronize(c->stream));
  return UWSPR_OK;
}

extern "C" int uwspr_sync_and_demodulate_batch(uwspr_ctx *c, const float *frames, int B, int where,
                                               const uwspr_sync_call *calls, int ncalls,
                                               uwspr_sync_result *results) {
  int rc = ready(c);
  if (rc) return rc;
  if (!calls || !results || ncalls <= 0) return fail(c, UWSPR_ERR_ARG, "calls/results/ncalls");
  // expand each call into its (ifreq, lag) loop, cc:160-165
  std::vector<uwspr_hyp> hyps;
  std::vector<size_t> first(ncalls + 1, 0);
  bool soft = false;
  for (int q = 0; q < ncalls; q++) {
    const uwspr_sync_call &k = calls[q];
    if (k.mode < 0 || k.mode > 2) return fail(c, UWSPR_ERR_ARG, "calls[%d].mode=%d", q, k.mode);
    if (k.symfac != 50) return fail(c, UWSPR_ERR_UNSUPPORTED, "calls[%d].symfac=%d (the path uses 50)", q, k.symfac);
    if (k.frame < 0 || k.frame >= B) return fail(c, UWSPR_ERR_ARG, "calls[%d].frame=%d", q, k.frame);
    int ifmin = k.ifmin, ifmax = k.ifmax, lagmin = k.lagmin, lagmax = k.lagmax;
    float fstep = k.fstep;
    if (k.mode == 0) { ifmin = 0; ifmax = 0; fstep = 0.0f; }
    if (k.mode == 1) { lagmin = k.shift1; lagmax = k.shift1; }
    if (k.mode == 2) { lagmin = k.shift1; lagmax = k.shift1; ifmin = 0; ifmax = 0; soft = true; }
    if (k.lagstep <= 0) return fail(c, UWSPR_ERR_ARG, "calls[%d].lagstep=%d", q, k.lagstep);
    first[q] = hyps.size();
    for (int ifreq = ifmin; ifreq <= ifmax; ifreq++) {
      const float f0 = k.f1 + (float)ifreq * fstep;  // cc:164
      for (int lag = lagmin; lag <= lagmax; lag += k.lagstep) {
        uwspr_hyp h;
        memset(&h, 0, sizeof(h));
        h.frame = k.frame; h.m_type = k.candidate.m_type; h.f0 = f0; h.lag = lag;
        h.drift = k.drift1;
        if (k.candidate.m_type == UWSPR_NONLINEAR) {
          h.V1 = k.candidate.m_nonlinear.V1; h.V2 = k.candidate.m_nonlinear.V2;
          h.p1 = k.candidate.m_nonlinear.p1; h.p2 = k.candidate.m_nonlinear.p2;
        }
        hyps.push_back(h);
        if (hyps.size() > (size_t)1 << 26) return fail(c, UWSPR_ERR_ARG, "more than 2^26 hypotheses in one batch");
      }
    }
  }
  first[ncalls] = hyps.size();
  const int H = (int)hyps.size();
  std::vector<float> sync(std::max(H, 1));
  std::vector<uint8_t> sym(soft ? (size_t)std::max(H, 1) * UWSPR_NSYM : 0);
  if (H > 0) {
    // frames may be device memory while the call records are host memory
    const float *d;
    if ((rc = frames_on_device(c, frames, B, where, &d))) return rc;
    if ((rc = ensure(c, &c->d_hyps, &c->cap_hyps, (size_t)H))) return rc;
    if ((rc = ensure(c, &c->d_abi_hyps, &c->cap_abi_hyps, (size_t)H))) return rc;
    if ((rc = ensure_sweep(c, (size_t)H, soft))) return rc;
    HIPCHK(c, hipMemcpyAsync(c->d_abi_hyps, hyps.data(), (size_t)H * sizeof(uwspr_hyp), hipMemcpyHostToDevice, c->stream));
    launch_prep_hyps(c, c->d_abi_hyps, c->d_hyps, H);
    launch_tonecorr(c, d, B, c->d_hyps, H, c->d_p);
    launch_fold(c, c->d_hyps, c->d_p, H, c->d_sync, soft ? c->d_sym : nullptr);
    HIPCHK(c, hipGetLastError());
    HIPCHK(c, hipMemcpyAsync(sync.data(), c->d_sync, (size_t)H * 4, hipMemcpyDeviceToHost, c->stream));
    if (soft) HIPCHK(c, hipMemcpyAsync(sym.data(), c->d_sym, (size_t)H * UWSPR_NSYM, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
  }
  for (int q = 0; q < ncalls; q++) {
    const uwspr_sync_call &k = calls[q];
    uwspr_sync_result &r = results[q];
    memset(&r, 0, sizeof(r));
    float syncmax = -1e30f, fbest = 0.0f;   // cc:156-159
    int best_shift = 0;
    for (size_t h = first[q]; h < first[q + 1]; h++)
      if (sync[h] > syncmax) { syncmax = sync[h]; best_shift = hyps[h].lag; fbest = hyps[h].f0; }  // cc:227-231
    r.sync = syncmax;
    if (k.mode <= 1) { r.shift1 = best_shift; r.f1 = fbest; }            // cc:234-238
    else {
      r.shift1 = k.shift1; r.f1 = k.f1;                                  // mode 2 leaves them alone
      if (first[q + 1] > first[q]) memcpy(r.symbols, &sym[(first[q + 1] - 1) * UWSPR_NSYM], UWSPR_NSYM);
    }
  }
  return UWSPR_OK;
}

// ------------------------------------------------------ refinement schedule
static int run_schedule_impl(uwspr_ctx *c, const float *dframes, int B, const uwspr_candidate *dcands,
                             const int32_t *dnpk, int cand_stride, int per_frame, uwspr_demod_out *user_out);

// What uwspr_demod_resume may continue is recorded here: the batch shape, whether the pass was lazy
// (only then were the winner's magnitudes kept) and where its records went.  A failed call leaves
// nothing to resume.
static int run_schedule(uwspr_ctx *c, const float *dframes, int B, const uwspr_candidate *dcands,
                        const int32_t *dnpk, int cand_stride, int per_frame,
                        uwspr_demod_out *user_out = nullptr) {
  c->last_sched_B = 0; c->last_sched_per_frame = 0; c->last_sched_lazy = false; c->last_sched_out = nullptr;
  const int rc = run_schedule_impl(c, dframes, B, dcands, dnpk, cand_stride, per_frame, user_out);
  if (rc) return rc;
  c->last_slots = B * per_frame; c->last_sched_B = B; c->last_sched_per_frame = per_frame;
  c->last_sched_lazy = c->ntries < UWSPR_NJIG;
  c->last_sched_out = c->cur_dout;
  return UWSPR_OK;
}

static int run_schedule_impl(uwspr_ctx *c, const float *dframes, int B, const uwspr_candidate *dcands,
                             const int32_t *dnpk, int cand_stride, int per_frame, uwspr_demod_out *user_out) {
  static const int hpc[6] = {5, 5, 2, 5, 5, UWSPR_NJIG};
  const size_t nslots = (size_t)B * per_frame;
  c->sched_per_frame = per_frame;
  int rc;
  if ((rc = ensure(c, &c->d_state, &c->cap_state, nslots))) return rc;
  const int njig = c->ntries < UWSPR_NJIG ? c->ntries : UWSPR_NJIG;
  if (c->use_fused || njig < UWSPR_NJIG) {   // (a lazy staged pass is resumed by the fused kernel)
    if (c->sched_grid <= 0) c->sched_grid = c->num_cus;   // one 16-wave workgroup per CU
    if ((rc = ensure(c, &c->d_tabs, &c->cap_tabs, (size_t)c->sched_grid * kSchedTabFloats))) return rc;
    if (!c->d_counter) HIPCHK(c, hipMalloc((void **)&c->d_counter, 64));
  }
  if (c->use_fused) {
    // k6_sched: one workgroup per candidate, S0..S5 back to back
    if (c->opt[UWSPR_OPT_SCHED_STAMPS] &&
        (rc = ensure(c, &c->d_sched_stamps, &c->cap_sched_stamps, nslots * 64))) return rc;
    if ((rc = ensure(c, &c->d_dout, &c->cap_dout, nslots))) return rc;
    c->cur_dout = user_out ? user_out : c->d_dout;
    if (c->ntries < UWSPR_NJIG) {   // lazy tries: keep what a later uwspr_demod_resume needs
      if ((rc = ensure(c, &c->d_pwin, &c->cap_pwin, nslots * UWSPR_NSYM * 4))) return rc;
    }
    launch_sched_fused(c, dframes, B, dcands, dnpk, cand_stride, per_frame, c->cur_dout,
                       c->ntries < UWSPR_NJIG ? c->ntries : UWSPR_NJIG);
    HIPCHK(c, hipGetLastError());
    return UWSPR_OK;
  }
  if ((rc = ensure(c, &c->d_hyps, &c->cap_hyps, 2 * nslots * UWSPR_NJIG))) return rc;
  if ((rc = ensure(c, &c->d_grps, &c->cap_grps, 3 * nslots))) return rc;
  // centres (48 B) followed by their frame indices (4 B): 13 int32 per slot in one buffer
  if ((rc = ensure(c, &c->d_cent, &c->cap_cent, (nslots * 13 + 11) / 12 + 1))) return rc;
  c->d_cent_frame = reinterpret_cast<int32_t *>(c->d_cent + nslots);
  if ((rc = ensure_sweep(c, nslots * UWSPR_NJIG, true))) return rc;
  if ((rc = ensure(c, &c->d_dout, &c->cap_dout, nslots))) return rc;
  c->cur_dout = user_out ? user_out : c->d_dout;
  dev_hyp *half[2] = {c->d_hyps, c->d_hyps + nslots * UWSPR_NJIG};
  // the stage winner's magnitudes, carried from stage to stage (try 0 of stage 5 repeats the stage-4 winner),
  // and the phasor tables of the lag stages (set A is built by the init kernel: both before it)
  if ((rc = ensure(c, &c->d_pwin, &c->cap_pwin, nslots * UWSPR_NSYM * 4))) return rc;
  if ((rc = ensure(c, &c->d_ptab, &c->cap_ptab, nslots * kPtabPerSlot * kPtabFloat2))) return rc;
  launch_sched_init(c, dcands, dnpk, cand_stride, B, per_frame);
  const bool lazy = njig < UWSPR_NJIG;   // only tries idt < njig of stage 5, packed njig per slot
  // Staged form, "stage_kernels": 1 (default) = S0 sample-major on packed rows (k4_lag0), S1 / S4 packed frequency stage
  // (k4_fpack), S3 / S5 LDS ring (k4_ring), S2 mirrored pairs (k4_dpair);
  // 0 = the flat kernel for every stage (the independent reference form of the equivalence tests).
  const int sk = c->opt[UWSPR_OPT_STAGE_KERNELS];
  for (int s = 0; s < 6; s++) {
    c->fast_now = c->fast_search && s < 5;   // S5 (the soft symbols) is always the reference's arithmetic
    const int H = (int)(nslots * (s == 5 ? njig : hpc[s]));
    const dev_hyp *h = half[s & 1];
    if (sk == 0 || (lazy && s == 5)) launch_tonecorr(c, dframes, B, h, H, c->d_p);   // (lazy S5: few, unrelated lags)
#if !defined(UWSPR_S2_PAIRS) || UWSPR_S2_PAIRS   // (0: experiment builds, the flat kernel as before)
    else if (s == 2) {
      // S2: candidates that came in without drift have mirrored tries: one phasor recurrence for both (k4_pair.hip);
      // the others through the flat kernel, which leaves the former alone -- not needed after this context's own FDR
      // with maxdrift = 0
      launch_tonecorr_dpair(c, dframes, B, h, (int)nslots, c->d_p);
      if (!(c->cands_from_fdr && c->p.maxdrift == 0)) launch_tonecorr(c, dframes, B, h, H, c->d_p, nullptr, 1, true);
    }
#else
    else if (s == 2) launch_tonecorr(c, dframes, B, h, H, c->d_p);
#endif
    else if (s == 1 || s == 4) launch_tonecorr_fstage(c, dframes, B, h, (int)nslots, H, c->d_p);
    else if (s == 3) launch_tonecorr_ring(c, dframes, B, c->d_grps, (int)nslots, 5, 16, H, c->d_p, 1);
    else if (s == 5 && (c->opt[UWSPR_OPT_K4_FORMS] & 1))
      launch_tonecorr_jig(c, dframes, B, c->d_grps, (int)(3 * nslots), H, c->d_p, 3);   // register ring (k4_jig.hip)
    else if (s == 5) launch_tonecorr_ring(c, dframes, B, c->d_grps, (int)(3 * nslots), 6, 8, H, c->d_p, 3);
    else if (c->use_ptab) {
      // S0: the slots whose frequency does not depend on the symbol (they have their phasor table) sample-major;
      // the others -- drifting linear candidates -- through the flat kernel, which leaves the former alone.
      // Candidates from this context's own FDR with maxdrift = 0 have no drift: that launch is not needed.
      launch_tonecorr_lag0(c, dframes, B, c->d_grps, (int)nslots, H, c->d_p);
      if (!(c->cands_from_fdr && c->p.maxdrift == 0)) launch_tonecorr(c, dframes, B, h, H, c->d_p, c->d_grps, 5);
    }
    else launch_tonecorr(c, dframes, B, h, H, c->d_p);
    if (s < 5) {
      launch_fold_step(c, s + 1, (int)nslots, njig);   // fold of stage s + transition to stage s+1
      c->fast_now = false;
    } else {
      launch_fold(c, h, c->d_p, H, c->d_sync, c->d_sym, (const float4 *)c->d_pwin, njig);
      launch_sched_finish(c, (int)nslots, njig);
      if (lazy) launch_keep_try0(c, (int)nslots, njig);   // what uwspr_demod_resume starts from
    }
  }
  HIPCHK(c, hipGetLastError());
  return UWSPR_OK;
}

extern "C" int uwspr_demod_batch(uwspr_ctx *c, const float *frames, int B, int where,
                                 const uwspr_candidate *cands, const int32_t *npk, int cand_stride,
                                 int max_per_frame, uwspr_demod_out *out) {
  int rc = ready(c);
  if (rc) return rc;
  if (!cands || !npk || cand_stride <= 0 || max_per_frame <= 0 || !out)
    return fail(c, UWSPR_ERR_ARG, "cands/npk/cand_stride/max_per_frame/out");
  const float *d;
  if ((rc = frames_on_device(c, frames, B, where, &d))) return rc;
  const uwspr_candidate *dc = cands;
  const int32_t *dn = npk;
  const bool host_recs = where != UWSPR_DEVICE;   // UWSPR_DEVICE_FRAMES: frames on the device, records on the host
  if (host_recs) {
    if ((rc = ensure(c, &c->d_tmpc, &c->cap_tmpc, (size_t)B * cand_stride))) return rc;
    if ((rc = ensure(c, &c->d_tmpn, &c->cap_tmpn, (size_t)B))) return rc;
    HIPCHK(c, hipMemcpyAsync(c->d_tmpc, cands, (size_t)B * cand_stride * sizeof(uwspr_candidate), hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, hipMemcpyAsync(c->d_tmpn, npk, (size_t)B * sizeof(int32_t), hipMemcpyHostToDevice, c->stream));
    dc = c->d_tmpc; dn = c->d_tmpn;
  }
  c->cands_from_fdr = false;   // the caller's candidates: any drift
  rc = run_schedule(c, d, B, dc, dn, cand_stride, max_per_frame);
  if (!rc) rc = copy_out(c, out, c->d_dout, (size_t)B * max_per_frame * sizeof(uwspr_demod_out), host_recs ? UWSPR_HOST : UWSPR_DEVICE);
  if (!rc && !host_recs) c->last_sched_out = out;   // a device caller's copy of the records is what a device resume patches
  if (host_recs) (void)hipStreamSynchronize(c->stream);
  // uwspr_pack_slabs works on the buffers of a uwspr_pipeline_batch call only
  c->last_per_frame = 0;
  return rc;
}

extern "C" int uwspr_pipeline_batch(uwspr_ctx *c, const float *frames, int B, int where,
                                    int max_per_frame, uwspr_candidate *cands, int32_t *npk,
                                    uwspr_demod_out *out) {
  int rc = ready(c);
  if (rc) return rc;
  if (max_per_frame <= 0) return fail(c, UWSPR_ERR_ARG, "max_per_frame=%d", max_per_frame);
  const float *d;
  if ((rc = frames_on_device(c, frames, B, where, &d))) return rc;
  // device callers get the results written straight into their buffers
  const bool dev = where == UWSPR_DEVICE;
  if ((rc = run_fdr(c, d, B, dev ? cands : nullptr, (dev && cands) ? npk : nullptr))) return rc;
  c->cands_from_fdr = true;
  if ((rc = run_schedule(c, d, B, c->cur_cands, c->cur_npk, c->fc.maxfreqs, max_per_frame,
                         dev ? out : nullptr))) return rc;
  c->last_per_frame = max_per_frame;
  if (c->next_slab) {   // uwspr_pipeline_slabs: packed by the schedule's last kernel (staged form) or here
    if (!c->next_slab_done) launch_pack_slabs(c, c->cur_cands, c->cur_npk, c->cur_dout, max_per_frame, c->next_slab_K, c->next_slab, B);
    c->next_slab = nullptr; c->next_slab_done = false;
    HIPCHK(c, hipGetLastError());
  }
  if (c->cur_cands == c->d_cands &&
      (rc = copy_out(c, cands, c->d_cands, (size_t)B * c->fc.maxfreqs * sizeof(uwspr_candidate), where))) return rc;
  if (c->cur_npk == c->d_npk && (rc = copy_out(c, npk, c->d_npk, (size_t)B * sizeof(int32_t), where))) return rc;
  if (c->cur_dout == c->d_dout &&
      (rc = copy_out(c, out, c->d_dout, (size_t)B * max_per_frame * sizeof(uwspr_demod_out), where))) return rc;
  if (where != UWSPR_DEVICE) HIPCHK(c, hipStreamSynchronize(c->stream));
  return UWSPR_OK;
}

// ------------------------------------------------- overlap-aware stream ingest
// frame f = stream samples [f hop, f hop + fl): overlapping source rows (no memcpy2D form), contiguous destination
__global__ void k_cut_frames(const float2 *__restrict__ src, float2 *__restrict__ dst, int hop, int fl, int f0) {
  const int f = f0 + blockIdx.y;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < fl; i += gridDim.x * blockDim.x)
    dst[(size_t)f * fl + i] = src[(size_t)f * hop + i];
}

extern "C" int uwspr_set_frame_stride(uwspr_ctx *c, int stride) {
  int rc = ready(c);
  if (rc) return rc;
  if (stride < 0) return fail(c, UWSPR_ERR_ARG, "stride=%d", stride);
  c->fstride = stride > 0 ? stride : c->fc.fl;
  return UWSPR_OK;
}

extern "C" int uwspr_stream_open(uwspr_ctx *c, int hop, int max_frames) {
  int rc = ready(c);
  if (rc) return rc;
  if (hop <= 0 || hop > c->fc.fl || max_frames <= 0) return fail(c, UWSPR_ERR_ARG, "hop=%d max_frames=%d", hop, max_frames);
  HIPCHK(c, hipStreamSynchronize(c->stream));
  if (!c->ring.open(c->fc.fl, hop, max_frames))
    return fail(c, UWSPR_ERR_NOMEM, "stream buffers (%d frames, hop %d): %s", max_frames, hop, hipGetErrorString(c->ring.err));
  if (!c->ring_ev) HIPCHK(c, hipEventCreateWithFlags(&c->ring_ev, hipEventDisableTiming));
  return UWSPR_OK;
}

extern "C" int uwspr_stream_push(uwspr_ctx *c, const float *iq, int nsamples, int where, int *nready) {
  int rc = ready(c);
  if (rc) return rc;
  stream_ring &r = c->ring;
  if (!r.is_open()) return fail(c, UWSPR_ERR_ARG, "uwspr_stream_push before uwspr_stream_open");
  if (nsamples < 0 || (nsamples > 0 && !iq)) return fail(c, UWSPR_ERR_ARG, "iq/nsamples");
  if (where != UWSPR_HOST && where != UWSPR_DEVICE && where != UWSPR_HOST_ASYNC) return fail(c, UWSPR_ERR_ARG, "where=%d", where);
  if (r.have + (size_t)nsamples > r.cap)
    return fail(c, UWSPR_ERR_ARG, "stream buffer full (%zu + %d > %zu samples): take frames first", r.have, nsamples, r.cap);
  if (nsamples > 0) {
    bool ok;
    if (where == UWSPR_DEVICE) {   // produced by work on the context's stream
      HIPCHK(c, hipEventRecord(c->ring_ev, c->stream));
      ok = r.append(iq, (size_t)nsamples, true, c->ring_ev);
    } else {
      ok = r.append(iq, (size_t)nsamples, false);
      // The upload is a DMA on the ring's copy stream.  A pageable source has been staged and is free
      // again; a page-locked one is read by the DMA itself, so UWSPR_HOST waits for it (the copy stream
      // carries nothing but uploads: this is the transfer time, not a wait for the search kernels) and
      // UWSPR_HOST_ASYNC leaves that to uwspr_stream_wait_uploads.
      if (ok && where == UWSPR_HOST && r.last_direct) ok = r.wait_uploads();
    }
    if (!ok) return fail(c, UWSPR_ERR_HIP, "stream upload: %s", hipGetErrorString(r.err));
  }
  if (nready) *nready = r.ready();
  return UWSPR_OK;
}

extern "C" int uwspr_stream_wait_uploads(uwspr_ctx *c) {
  int rc = ready(c);
  if (rc) return rc;
  if (c->ring.is_open() && !c->ring.wait_uploads()) return fail(c, UWSPR_ERR_HIP, "stream upload: %s", hipGetErrorString(c->ring.err));
  return UWSPR_OK;
}

// the next k frames in place on the context's stream (shared by the two take forms)
static int stream_view(uwspr_ctx *c, int nframes, const float **view, long long *first_pos) {
  stream_ring &r = c->ring;
  if (!r.is_open() || nframes <= 0 || nframes > r.ready())
    return fail(c, UWSPR_ERR_ARG, "uwspr_stream_take(%d): %d frames are complete", nframes, r.is_open() ? r.ready() : 0);
  // whatever read earlier views has been enqueued on the stream by now (a view is valid until the next take)
  HIPCHK(c, hipEventRecord(c->ring_ev, c->stream));
  r.reader_done(0, c->ring_ev); r.reader_done(1, c->ring_ev);
  if (!r.view(nframes, c->stream, view, first_pos, nullptr)) return fail(c, UWSPR_ERR_HIP, "stream view: %s", hipGetErrorString(r.err));
  return UWSPR_OK;
}

extern "C" int uwspr_stream_take_view(uwspr_ctx *c, int nframes, const float **frames, int *stride, long long *first_pos) {
  int rc = ready(c);
  if (rc) return rc;
  if (!frames) return fail(c, UWSPR_ERR_ARG, "frames");
  if ((rc = stream_view(c, nframes, frames, first_pos))) return rc;
  if (stride) *stride = c->ring.hop;
  return UWSPR_OK;
}

extern "C" int uwspr_stream_take(uwspr_ctx *c, int nframes, float *dev_dst, const float **frames, long long *first_pos) {
  int rc = ready(c);
  if (rc) return rc;
  const float *src = nullptr;
  if ((rc = stream_view(c, nframes, &src, first_pos))) return rc;
  float *dst = dev_dst;
  if (!dst) {
    if ((rc = ensure(c, &c->d_stream_frames, &c->cap_stream_frames, (size_t)c->ring.maxf * c->fc.fl * 2))) return rc;
    dst = c->d_stream_frames;
  }
  for (int f0 = 0; f0 < nframes; f0 += 32768) {
    const int nf = std::min(32768, nframes - f0);
    hipLaunchKernelGGL(k_cut_frames, dim3(44, nf), dim3(256), 0, c->stream, (const float2 *)src, (float2 *)dst,
                       c->ring.hop, c->fc.fl, f0);
  }
  HIPCHK(c, hipGetLastError());
  if (frames) *frames = dst;
  return UWSPR_OK;
}

extern "C" int uwspr_stream_reset(uwspr_ctx *c, long long pos) {
  int rc = ready(c);
  if (rc) return rc;
  // The kernels of the LAST view may still be enqueued (a view records the readers of the views BEFORE it): put the
  // buffers' next writer -- an append after the reset switches buffers, a second reset switches back -- behind
  // everything the stream holds now, as a take does.
  if (c->ring.is_open() && c->ring_ev) {
    HIPCHK(c, hipEventRecord(c->ring_ev, c->stream));
    c->ring.reader_done(0, c->ring_ev); c->ring.reader_done(1, c->ring_ev);
  }
  c->ring.reset(pos);
  return UWSPR_OK;
}

extern "C" int uwspr_device_alloc(size_t bytes, void **ptr) {
  if (!ptr) return UWSPR_ERR_ARG;
  return hipMalloc(ptr, bytes) == hipSuccess ? UWSPR_OK : UWSPR_ERR_NOMEM;
}
extern "C" void uwspr_device_free(void *ptr) { if (ptr) (void)hipFree(ptr); }

extern "C" int uwspr_host_alloc(size_t bytes, void **ptr) {
  if (!ptr) return UWSPR_ERR_ARG;
  return hipHostMalloc(ptr, bytes, hipHostMallocDefault) == hipSuccess ? UWSPR_OK : UWSPR_ERR_NOMEM;
}
extern "C" void uwspr_host_free(void *ptr) { if (ptr) (void)hipHostFree(ptr); }

extern "C" int uwspr_set_tries(uwspr_ctx *c, int ntries) {
  int rc = ready(c);
  if (rc) return rc;
  if (ntries < 1 || ntries > UWSPR_NJIG) return fail(c, UWSPR_ERR_ARG, "ntries=%d (1..%d)", ntries, UWSPR_NJIG);
  c->ntries = ntries;
  return UWSPR_OK;
}

extern "C" int uwspr_demod_resume(uwspr_ctx *c, const float *frames, int B, int where, const uint8_t *need,
                                  int max_per_frame, uwspr_demod_out *out) {
  int rc = ready(c);
  if (rc) return rc;
  if (!need || !out || B <= 0 || max_per_frame <= 0) return fail(c, UWSPR_ERR_ARG, "need/out/B/max_per_frame");
  if (!c->d_pwin || !c->last_sched_lazy || c->last_sched_B != B || c->last_sched_per_frame != max_per_frame)
    return fail(c, UWSPR_ERR_ARG, "uwspr_demod_resume follows a schedule call of the same batch made with uwspr_set_tries(< %d)", UWSPR_NJIG);
  // the records of the first pass must be where this call patches them
  if (where == UWSPR_DEVICE ? (out != c->last_sched_out) : (c->last_sched_out != c->d_dout))
    return fail(c, UWSPR_ERR_ARG, "uwspr_demod_resume: the first pass wrote its records to %s; resume with %s",
                c->last_sched_out == c->d_dout ? "the context (a host-record call)" : "the caller's device buffer",
                c->last_sched_out == c->d_dout ? "where = UWSPR_HOST / UWSPR_DEVICE_FRAMES" : "where = UWSPR_DEVICE and that buffer");
  const float *d;
  if ((rc = frames_on_device(c, frames, B, where, &d))) return rc;
  const size_t nslots = (size_t)B * max_per_frame;
  const uint8_t *dneed = need;
  uwspr_demod_out *dout = out;
  const bool host_io = where != UWSPR_DEVICE;   // UWSPR_DEVICE_FRAMES: frames in HBM, need / out in host memory
  if (host_io) {
    if ((rc = ensure(c, &c->d_need, &c->cap_need, nslots))) return rc;
    HIPCHK(c, hipMemcpyAsync(c->d_need, need, nslots, hipMemcpyHostToDevice, c->stream));
    dneed = c->d_need;
    dout = c->d_dout;   // the records of the first pass are still there
  }
  launch_sched_fused(c, d, B, nullptr, nullptr, 0, max_per_frame, dout, UWSPR_NJIG, dneed);
  HIPCHK(c, hipGetLastError());
  if (host_io) {
    HIPCHK(c, hipMemcpyAsync(out, c->d_dout, nslots * sizeof(uwspr_demod_out), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
  }
  return UWSPR_OK;
}

extern "C" int uwspr_pipeline_slabs(uwspr_ctx *c, int K, void *slabs_device) {
  int rc = ready(c);
  if (rc) return rc;
  if (slabs_device && (K < 1 || K > c->fc.maxfreqs)) return fail(c, UWSPR_ERR_ARG, "uwspr_pipeline_slabs: K=%d", K);
  c->next_slab = (uint8_t *)slabs_device; c->next_slab_K = K; c->next_slab_done = false;
  return UWSPR_OK;
}

extern "C" int uwspr_pack_slabs(uwspr_ctx *c, int B, int K, void *slabs, int where) {
  int rc = ready(c);
  if (rc) return rc;
  if (!slabs || K < 1 || K > c->fc.maxfreqs || B <= 0 || B > c->last_B || c->last_per_frame < 1 || !c->cur_dout)
    return fail(c, UWSPR_ERR_ARG, "uwspr_pack_slabs: needs a preceding uwspr_pipeline_batch of >= %d frames", B);
  const size_t bytes = (size_t)B * (16 + (size_t)K * 48 + 16);
  uint8_t *dst = (uint8_t *)slabs;
  if (where == UWSPR_HOST) {
    size_t cap = c->cap_slab;
    if ((rc = ensure(c, &c->d_slab, &cap, bytes))) return rc;
    c->cap_slab = cap;
    dst = c->d_slab;
  }
  launch_pack_slabs(c, c->cur_cands, c->cur_npk, c->cur_dout, c->last_per_frame, K, dst, B);
  HIPCHK(c, hipGetLastError());
  if (where == UWSPR_HOST) {
    HIPCHK(c, hipMemcpyAsync(slabs, c->d_slab, bytes, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
  }
  return UWSPR_OK;
}

// -------------------------------------------------------------- profiling
// diagnostics (not part of the ABI header): phase boundary times of the last fused schedule launch
extern "C" int uwspr_debug_sched_stamps(uwspr_ctx *c, unsigned long long *out, int nslots) {
  if (!c || !out || !c->d_sched_stamps || (size_t)nslots * 64 > c->cap_sched_stamps) return UWSPR_ERR_ARG;
  HIPCHK(c, hipStreamSynchronize(c->stream));
  HIPCHK(c, hipMemcpy(out, c->d_sched_stamps, (size_t)nslots * 64 * sizeof(unsigned long long), hipMemcpyDeviceToHost));
  return UWSPR_OK;
}

extern "C" int uwspr_prof_enable(uwspr_ctx *c, int mask) {
  int rc = ready(c);
  if (rc) return rc;
  c->prof_mask = mask & UWSPR_PROF_ALL;
  return UWSPR_OK;
}

extern "C" int uwspr_prof_intervals(uwspr_ctx *c, int kind, void *epoch_event, double *start_ms,
                                    double *stop_ms, int cap, int *n) {
  int rc = ready(c);
  if (rc) return rc;
  if (!epoch_event || !start_ms || !stop_ms || !n || cap < 0) return UWSPR_ERR_ARG;
  HIPCHK(c, hipStreamSynchronize(c->stream));
  int k = 0;
  for (auto &e : c->prof_events) {
    if (e.kind != kind) continue;
    float a = 0.0f, b = 0.0f;
    if (k < cap && hipEventElapsedTime(&a, (hipEvent_t)epoch_event, e.a) == hipSuccess &&
        hipEventElapsedTime(&b, (hipEvent_t)epoch_event, e.b) == hipSuccess) {
      start_ms[k] = a; stop_ms[k] = b;
      k++;
    }
  }
  *n = k;
  return UWSPR_OK;
}

extern "C" int uwspr_prof_read(uwspr_ctx *c, uwspr_prof *o) {
  int rc = ready(c);
  if (rc) return rc;
  if (!o) return UWSPR_ERR_ARG;
  memset(o, 0, sizeof(*o));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  for (auto &e : c->prof_events) {
    float ms = 0.0f;
    if (hipEventElapsedTime(&ms, e.a, e.b) == hipSuccess) {
      o->ms[e.kind] += ms; o->launches[e.kind] += 1; o->units[e.kind] += e.units;
    }
    c->ev_pool.push_back(e.a); c->ev_pool.push_back(e.b);
  }
  c->prof_events.clear();
  return UWSPR_OK;
}
