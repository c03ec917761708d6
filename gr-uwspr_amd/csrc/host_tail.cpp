// host_tail.cpp -- the sequential tail of the path that stays on the host
// (SURVEY 8(f) next-1..3): de-interleave, the K=32 r=1/2 Fano sequential
// decoder, the gate/retry loop around it, WSPR message unpacking and the .c2
// reader.  The GPU hands over uwspr_demod_out records; nothing here touches HIP.
//
// Reference behaviour followed:
//   deinterleave        lib/sync_and_demodulate_impl.cc:265-282
//   encoder             lib/Fano.cc:54-100   (Layland-Lushbaugh polynomials)
//   Fano decoder        lib/Fano.cc:110-252  (same thresholds, cycle budget and
//                                             tail handling => same decode /
//                                             time-out decisions)
//   metric table        lib/Fano.cc:36-45    (bias 0.45, scale 10, table [2])
//   gate + retry loop   lib/sync_and_demodulate_impl.cc:457-490
//   message unpack      lib/helpers.cc:321-590 (without the on-disk hash table)
//   .c2 layout          lib/c2file_source_impl.cc:80-96
#include <math.h>
#include <sched.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <atomic>
#include <condition_variable>
#include <functional>
#include <mutex>
#include <thread>
#include <vector>

#include "../../include/uwspr_hip.h"
#include "host_pool.h"

namespace {

// mettab[0][i] = round(10*(metric_tables[2][i] - 0.45)) (Fano.cc:41-43); the
// integers below are that table as the reference's constructor produces it
// (regenerated and compared by tests/test_host_tail.py through oracle/_ref);
// mettab[1][i] = mettab[0][255-i].
const short kMet0[256] = {
    5, 5, 5, 5, 5, 5, 5, 5, 5, 5, 5, 5, 5, 5, 5, 5, 5, 5, 5, 5, 5, 5, 5, 5, 5, 5, 5, 5, 5, 5, 5, 5,
    5, 5, 5, 5, 5, 5, 5, 5, 5, 5, 5, 5, 5, 5, 5, 5, 5, 5, 5, 5, 5, 5, 5, 5, 5, 5, 5, 5, 5, 5, 5, 5,
    5, 5, 5, 5, 5, 5, 5, 5, 5, 5, 5, 5, 5, 5, 5, 5, 5, 5, 5, 5, 5, 5, 5, 5, 5, 5, 4, 4, 4, 4, 4, 4,
    4, 4, 4, 4, 3, 3, 3, 3, 3, 3, 3, 2, 2, 2, 2, 1, 1, 1, 1, 0, 0, 0, -1, -1, -1, -2, -2, -3, -3,
    -4, -4, -5, -5, -6, -6, -7, -7, -8, -9, -9, -10, -10, -11, -12, -13, -13, -14, -15, -16, -16,
    -17, -18, -19, -20, -20, -21, -22, -23, -24, -25, -26, -27, -28, -28, -29, -30, -31, -32, -33,
    -34, -35, -36, -37, -38, -39, -40, -41, -42, -43, -44, -45, -46, -47, -48, -49, -50, -51, -52,
    -53, -54, -55, -56, -57, -58, -59, -60, -61, -62, -63, -64, -65, -66, -67, -68, -69, -70, -71,
    -72, -73, -74, -75, -76, -77, -78, -79, -80, -81, -83, -83, -84, -86, -87, -87, -89, -90, -91,
    -92, -93, -94, -95, -96, -97, -97, -99, -100, -101, -102, -103, -104, -105, -106, -108, -107,
    -110, -110, -112, -112, -114, -115, -115, -116, -117, -118, -119, -120, -121, -124, -123, -126,
    -137};

inline int met(int sent, int rx) { return sent ? kMet0[255 - rx] : kMet0[rx]; }

const uint32_t kPoly1 = 0xf2d05351u, kPoly2 = 0xe4613c47u;  // Fano.cc:54-55

inline unsigned parity32(uint32_t v) { return (unsigned)__builtin_parity(v); }
// rate-1/2 symbol pair for an encoder state: POLY1 parity in bit 1, POLY2 in bit 0
inline unsigned enc_pair(uint64_t state) {
  const uint32_t s = (uint32_t)state;
  return (parity32(s & kPoly1) << 1) | parity32(s & kPoly2);
}

struct fano_node {
  uint64_t encstate;
  long gamma;
  int metrics[4];
  int tm[2];
  int i;
};

}  // namespace

namespace {
// destination p takes source j = bit-reversed 8-bit counter, skipping j >= 162 (cc:265-282)
struct deint_table {
  uint8_t src[UWSPR_NSYM];
  deint_table() {
    int p = 0;
    for (unsigned i = 0; p < UWSPR_NSYM; i++) {
      unsigned j = 0;
      for (int b = 0; b < 8; b++) j |= ((i >> b) & 1u) << (7 - b);
      if (j < UWSPR_NSYM) src[p++] = (uint8_t)j;
    }
  }
};
const deint_table kDeint;
}  // namespace

extern "C" void uwspr_deinterleave(uint8_t *sym) {
  uint8_t tmp[UWSPR_NSYM];
  for (int p = 0; p < UWSPR_NSYM; p++) tmp[p] = sym[kDeint.src[p]];
  memcpy(sym, tmp, UWSPR_NSYM);
}

extern "C" int uwspr_fano_encode(uint8_t *symbols, const uint8_t *data, uint32_t nbytes) {
  uint64_t state = 0;
  for (uint32_t n = 0; n < nbytes; n++) {
    for (int i = 7; i >= 0; i--) {
      state = (state << 1) | ((data[n] >> i) & 1u);
      const unsigned sym = enc_pair(state);
      *symbols++ = (uint8_t)(sym >> 1);
      *symbols++ = (uint8_t)(sym & 1u);
    }
  }
  return 0;
}

// Sequential (Fano) decoding: walk the code tree keeping a running path
// metric gamma against a threshold t that moves in steps of delta.
extern "C" int uwspr_fano_decode(const uint8_t *symbols, uint8_t *data, uint32_t *metric,
                                 uint32_t *cycles, uint32_t *maxnp_out, int delta,
                                 uint32_t maxcycles) {
  const unsigned nbits = 81;
  fano_node nodes[82] = {};   // per call, on the caller's stack (zeroed: a timed-out decode reports zeros for the nodes it never reached)
  const int last = (int)nbits - 1;      // index of the last node
  const int tail = (int)nbits - 31;     // first node of the all-zero tail
  unsigned maxnp = 0;
  for (unsigned k = 0; k < nbits; k++) {
    const int a = symbols[2 * k], b = symbols[2 * k + 1];
    nodes[k].metrics[0] = met(0, a) + met(0, b);
    nodes[k].metrics[1] = met(0, a) + met(1, b);
    nodes[k].metrics[2] = met(1, a) + met(0, b);
    nodes[k].metrics[3] = met(1, a) + met(1, b);
  }
  auto sort_branches = [&](int k, bool in_tail) {
    fano_node &nd = nodes[k];
    const unsigned lsym = enc_pair(nd.encstate);  // 0-branch symbols
    if (in_tail) { nd.tm[0] = nd.metrics[lsym]; return; }
    const int m0 = nd.metrics[lsym], m1 = nd.metrics[3 ^ lsym];  // both polynomials are odd
    if (m0 > m1) { nd.tm[0] = m0; nd.tm[1] = m1; }
    else { nd.tm[0] = m1; nd.tm[1] = m0; nd.encstate++; }
  };
  int np = 0;
  nodes[0].encstate = 0;
  sort_branches(0, false);
  nodes[0].i = 0;
  const uint64_t budget = (uint64_t)maxcycles * nbits;
  int t = 0;
  nodes[0].gamma = 0;
  uint64_t i;
  for (i = 1; i <= budget; i++) {
    if (np > (int)maxnp) maxnp = (unsigned)np;
    fano_node &nd = nodes[np];
    const int ngamma = (int)(nd.gamma + nd.tm[nd.i]);
    if (ngamma >= t) {
      if (nd.gamma < t + delta)            // first visit: tighten the threshold
        while (ngamma >= t + delta) t += delta;
      nodes[np + 1].gamma = ngamma;        // move forward
      nodes[np + 1].encstate = nd.encstate << 1;
      if (++np == last + 1) break;         // done
      sort_branches(np, np >= tail);
      nodes[np].i = 0;
      continue;
    }
    for (;;) {                             // threshold violated: look back
      if (np == 0 || nodes[np - 1].gamma < t) {
        t -= delta;                        // cannot back up: relax the threshold
        if (nodes[np].i != 0) { nodes[np].i = 0; nodes[np].encstate ^= 1; }
        break;
      }
      --np;
      if (np < tail && nodes[np].i != 1) { // try the next-best branch
        nodes[np].i++;
        nodes[np].encstate ^= 1;
        break;
      }
    }
  }
  if (metric) *metric = (uint32_t)nodes[np].gamma;
  for (unsigned n = 0; n < (nbits >> 3); n++) data[n] = (uint8_t)nodes[7 + 8 * n].encstate;
  if (cycles) *cycles = (uint32_t)(i + 1);
  if (maxnp_out) *maxnp_out = maxnp;
  return i >= budget ? -1 : 0;
}

int uwspr::decode_try(const uwspr_demod_out *d, int idt, int8_t *message7) {
  const float minsync2 = 0.12f;
  const float minrms = (float)(52.0 * (50 / 64.0));
  if (!d->worth_a_try || idt < 0 || idt >= UWSPR_NJIG) return -1;
  if (!(d->jig_sync[idt] > minsync2 && d->jig_rms[idt] > minrms)) return -1;   // cc:470
  uint8_t sym[UWSPR_NSYM], data[11];
  memset(data, 0, sizeof(data));
  for (int p = 0; p < UWSPR_NSYM; p++) sym[p] = d->symbols[idt][kDeint.src[p]];
  uint32_t metric, cycles, maxnp;
  if (uwspr_fano_decode(sym, data, &metric, &cycles, &maxnp, 60, 10000) != 0) return 0;
  for (int i = 0; i < 7; i++) message7[i] = (int8_t)data[i];
  return 1;
}

// cc:457-490 from try `first` on (the tries before it have been attempted already: the lazy flow)
int uwspr::decode_candidate_from(const uwspr_demod_out *d, int first, int8_t *message7, int32_t *idt_used,
                                 int *fano_calls) {
  if (!d || !message7) return 0;
  if (!d->worth_a_try) return 0;
  const float minsync2 = 0.12f;
  const float minrms = (float)(52.0 * (50 / 64.0));
  for (int idt = first < 0 ? 0 : first; idt < UWSPR_NJIG; idt++) {
    if (d->jig_sync[idt] > minsync2 && d->jig_rms[idt] > minrms) {
      uint8_t sym[UWSPR_NSYM], data[11];
      memset(data, 0, sizeof(data));
      for (int p = 0; p < UWSPR_NSYM; p++) sym[p] = d->symbols[idt][kDeint.src[p]];
      uint32_t metric, cycles, maxnp;
      if (fano_calls) (*fano_calls)++;
      if (uwspr_fano_decode(sym, data, &metric, &cycles, &maxnp, 60, 10000) == 0) {
        for (int i = 0; i < 7; i++) message7[i] = (int8_t)data[i];
        if (idt_used) *idt_used = idt;
        return 1;
      }
    }
  }
  return 0;
}

extern "C" int uwspr_decode_candidate(const uwspr_demod_out *d, int8_t *message7, int32_t *idt_used) {
  return uwspr::decode_candidate_from(d, 0, message7, idt_used);
}

// ---- the persistent host pool (host_pool.h) ------------------------------------------------------
namespace uwspr {

// The CPUs this process may actually keep busy: hardware threads, capped by the scheduler affinity
// mask and by the cgroup CPU quota (a container that sees 256 hardware threads may own 16 of them;
// more runnable threads than that only get the whole group throttled).
static std::atomic<int> g_ranks_on_host{1};     // uwspr_host_set_ranks: processes of the job that share this host's CPUs
static std::atomic<bool> g_pool_made{false};

int host_cpu_share() {
  int n = (int)std::thread::hardware_concurrency();
  if (n < 1) n = 1;
  cpu_set_t set;
  if (sched_getaffinity(0, sizeof(set), &set) == 0) {
    const int a = CPU_COUNT(&set);
    if (a > 0 && a < n) n = a;
  }
  long long quota = -1, period = -1;
  if (FILE *f = fopen("/sys/fs/cgroup/cpu.max", "r")) {            // cgroup v2: "<quota|max> <period>"
    char q[64];
    if (fscanf(f, "%63s %lld", q, &period) == 2 && strcmp(q, "max") != 0) quota = atoll(q);
    fclose(f);
  } else {
    if (FILE *g = fopen("/sys/fs/cgroup/cpu/cpu.cfs_quota_us", "r")) { if (fscanf(g, "%lld", &quota) != 1) quota = -1; fclose(g); }
    if (FILE *g = fopen("/sys/fs/cgroup/cpu/cpu.cfs_period_us", "r")) { if (fscanf(g, "%lld", &period) != 1) period = -1; fclose(g); }
  }
  if (quota > 0 && period > 0) {
    const int c = (int)((quota + period - 1) / period);
    if (c >= 1 && c < n) n = c;
  }
  const int ranks = g_ranks_on_host.load(std::memory_order_relaxed);
  if (ranks > 1) n = n / ranks > 1 ? n / ranks : 1;
  return n;
}

host_pool::host_pool(int nthreads) {
  int n = nthreads > 0 ? nthreads : host_cpu_share();
  if (n < 1) n = 1;
  nworkers_ = n - 1;   // the caller of run() is the n-th
  for (int t = 0; t < nworkers_; t++) threads_.emplace_back([this]() { worker(); });
}

host_pool::~host_pool() {
  {
    std::lock_guard<std::mutex> lk(m_);
    stop_ = true;
  }
  cv_.notify_all();
  for (auto &t : threads_) t.join();
}

// items of one job until none is left to hand out; returns how many this thread ran
int host_pool::drain(job &j) {
  int ran = 0;
  for (;;) {
    const int i0 = j.next.fetch_add(j.chunk, std::memory_order_relaxed);
    if (i0 >= j.n) break;
    const int i1 = i0 + j.chunk < j.n ? i0 + j.chunk : j.n;
    for (int i = i0; i < i1; i++) (*j.fn)(i);
    ran += i1 - i0;
  }
  return ran;
}

// a job that still has items to hand out and room for another thread (m_ held)
host_pool::job *host_pool::pick() {
  const size_t nj = jobs_.size();
  for (size_t k = 0; k < nj; k++) {
    job *j = jobs_[(rr_ + k) % nj];
    if (j->next.load(std::memory_order_relaxed) < j->n && j->threads < j->max_threads) {
      rr_ = (rr_ + k + 1) % nj;   // the next worker looks at the next job first: jobs share the pool
      return j;
    }
  }
  return nullptr;
}

void host_pool::worker() {
  for (;;) {
    // a short spin first (no lock): between the batches of a busy pipeline the next job is microseconds away
    const uint64_t seen = gen_a_.load(std::memory_order_acquire);
    for (int spin = 0; spin < 200 && gen_a_.load(std::memory_order_acquire) == seen && pending_.load(std::memory_order_relaxed) == 0; spin++)
      __builtin_ia32_pause();
    job *j;
    {
      std::unique_lock<std::mutex> lk(m_);
      cv_.wait(lk, [&]() { return stop_ || (j = pick()) != nullptr; });
      if (stop_) return;
      j->threads++;
    }
    const int ran = drain(*j);
    {
      std::lock_guard<std::mutex> lk(m_);
      j->threads--;
      j->done += ran;
      if (j->next.load(std::memory_order_relaxed) >= j->n) pending_.store(count_pending(), std::memory_order_relaxed);
    }
    cv_done_.notify_all();
  }
}

int host_pool::count_pending() {   // jobs with items still to hand out (m_ held)
  int c = 0;
  for (job *j : jobs_) c += j->next.load(std::memory_order_relaxed) < j->n;
  return c;
}

// Several callers may have jobs in the pool at once (the pipe's coordinators: the Fano time-outs of one batch
// -- milliseconds each -- leave most threads idle, the next batch's decodes fill them).  Items are handed out
// per job by an atomic counter; a worker takes the next job that has items left and room for a thread.
void host_pool::run(int n, int max_threads, const std::function<void(int)> &fn) {
  if (n <= 0) return;
  job j;
  j.fn = &fn; j.n = n;
  int want = max_threads > 0 ? max_threads : nworkers_ + 1;
  if (want > n) want = n;
  j.max_threads = want;
  j.chunk = n / (8 * want) > 0 ? n / (8 * want) : 1;
  if (j.chunk > 16) j.chunk = 16;
  {
    std::lock_guard<std::mutex> lk(m_);
    jobs_.push_back(&j);
    j.threads = 1;   // the caller
    pending_.store(count_pending(), std::memory_order_relaxed);
    gen_a_.fetch_add(1, std::memory_order_release);
  }
  if (want > 1) cv_.notify_all();
  const int ran = drain(j);
  std::unique_lock<std::mutex> lk(m_);
  j.threads--;
  j.done += ran;
  cv_done_.wait(lk, [&]() { return j.done >= j.n && j.threads == 0; });
  for (size_t k = 0; k < jobs_.size(); k++)
    if (jobs_[k] == &j) { jobs_.erase(jobs_.begin() + k); break; }
  if (rr_ >= jobs_.size()) rr_ = 0;
  pending_.store(count_pending(), std::memory_order_relaxed);
}

host_pool &host_pool::shared() {
  g_pool_made.store(true);     // (before the pool is sized: a concurrent uwspr_host_set_ranks is refused, not half applied)
  static host_pool p(0);
  return p;
}

int host_set_ranks(int ranks) {
  if (ranks < 1) return UWSPR_ERR_ARG;
  if (g_pool_made.load()) return UWSPR_ERR_UNSUPPORTED;
  g_ranks_on_host.store(ranks);
  return UWSPR_OK;
}

}  // namespace uwspr

// Candidates are independent (cc:389 loops over them one by one), so a batch of records is decoded
// by the process-wide persistent pool (no thread is created or joined per call), indices handed out
// by one counter; record i's result does not depend on the thread count.
extern "C" int uwspr_host_threads(void) { return uwspr::host_cpu_share(); }
extern "C" int uwspr_host_set_ranks(int ranks) { return uwspr::host_set_ranks(ranks); }

extern "C" int uwspr_decode_batch(const uwspr_demod_out *d, int n, int nthreads, int8_t *messages,
                                  int32_t *idt_used, uint8_t *decoded) {
  if (n < 0 || (n > 0 && (!d || !messages || !decoded))) return UWSPR_ERR_ARG;
  std::atomic<int> good(0);
  uwspr::host_pool::shared().run(n, nthreads, [&](int i) {
    int32_t idt = -1;
    const int r = uwspr_decode_candidate(&d[i], messages + 7 * (size_t)i, &idt);
    decoded[i] = (uint8_t)r;
    if (!r) memset(messages + 7 * (size_t)i, 0, 7);
    if (idt_used) idt_used[i] = idt;
    if (r) good.fetch_add(1, std::memory_order_relaxed);
  });
  return good.load();
}

// ---- WSPR message unpack (types 1 and 2; type 3 without a hash table) ------
namespace {
const char kAlnum[] = "0123456789ABCDEFGHIJKLMNOPQRSTUVWXYZ ";

bool unpack_call(int32_t n, char *call /* >= 7 */) {
  if (n >= 262177560) return false;
  char tmp[7];
  tmp[5] = kAlnum[n % 27 + 10]; n /= 27;
  tmp[4] = kAlnum[n % 27 + 10]; n /= 27;
  tmp[3] = kAlnum[n % 27 + 10]; n /= 27;
  tmp[2] = kAlnum[n % 10]; n /= 10;
  tmp[1] = kAlnum[n % 36]; n /= 36;
  tmp[0] = kAlnum[n];
  tmp[6] = 0;
  int lead = 0;
  while (lead < 5 && tmp[lead] == ' ') lead++;
  snprintf(call, 7, "%s", tmp + lead);
  for (int i = (int)strlen(call) - 1; i >= 0 && call[i] == ' '; i--) call[i] = 0;
  char *sp = strchr(call, ' ');
  if (sp) *sp = 0;
  return true;
}

bool unpack_grid(int32_t ngrid, char *grid /* >= 5 */) {
  ngrid >>= 7;
  if (ngrid >= 32400) { strcpy(grid, "XXXX"); return false; }
  const int dlat = ngrid % 180 - 90;
  int dlong = (ngrid / 180) * 2 - 180 + 2;
  if (dlong < -180) dlong += 360;
  if (dlong > 180) dlong += 360;
  const int nlong = (int)(60.0 * (180.0 - dlong) / 5.0);
  int n1 = nlong / 240, n2 = (nlong - 240 * n1) / 24;
  grid[0] = kAlnum[10 + n1]; grid[2] = kAlnum[n2];
  const int nlat = (int)(60.0 * (dlat + 90) / 2.5);
  n1 = nlat / 240; n2 = (nlat - 240 * n1) / 24;
  grid[1] = kAlnum[10 + n1]; grid[3] = kAlnum[n2];
  grid[4] = 0;
  return true;
}

bool unpack_prefix(int32_t nprefix, char *call /* in: base call, out: full, >= 13 */) {
  char base[13];
  snprintf(base, sizeof(base), "%s", call);
  if (nprefix < 60000) {
    char pfx[4] = {0, 0, 0, 0};
    int32_t n = nprefix;
    for (int i = 2; i >= 0; i--) {
      const int nc = n % 37;
      pfx[i] = nc <= 9 ? (char)('0' + nc) : nc <= 35 ? (char)('A' + nc - 10) : ' ';
      n /= 37;
    }
    const char *p = strrchr(pfx, ' ');
    snprintf(call, 13, "%s/%s", p ? p + 1 : pfx, base);
    return true;
  }
  const int32_t nc = nprefix - 60000;
  if (nc >= 0 && nc <= 9) { snprintf(call, 13, "%s/%c", base, '0' + nc); return true; }
  if (nc >= 10 && nc <= 35) { snprintf(call, 13, "%s/%c", base, 'A' + nc - 10); return true; }
  if (nc >= 36 && nc <= 125) { snprintf(call, 13, "%s/%c%c", base, '0' + (nc - 26) / 10, '0' + (nc - 26) % 10); return true; }
  return false;
}
}  // namespace

extern "C" int uwspr_unpack_message(const int8_t *m, char *out, size_t out_len) {
  if (!m || !out || out_len < 23) return -1;
  const uint8_t *d = (const uint8_t *)m;
  const int32_t n1 = ((int32_t)d[0] << 20) | ((int32_t)d[1] << 12) | ((int32_t)d[2] << 4) | (d[3] >> 4);
  const int32_t n2 = ((int32_t)(d[3] & 15) << 18) | ((int32_t)d[4] << 10) | ((int32_t)d[5] << 2) | (d[6] >> 6);
  char call[13], grid[5];
  memset(call, 0, sizeof(call));
  out[0] = 0;
  if (!unpack_call(n1, call)) return 1;
  if (!unpack_grid(n2, grid)) return 1;
  const int ntype = (n2 & 127) - 64;
  if (ntype >= 0 && ntype <= 62) {
    const int nu = ntype % 10;
    if (nu == 0 || nu == 3 || nu == 7) {
      snprintf(out, out_len, "%s %s %2d", call, grid, ntype);
      return 0;
    }
    int nadd = nu;
    if (nu > 3) nadd = nu - 3;
    if (nu > 7) nadd = nu - 7;
    const int32_t n3 = n2 / 128 + 32768 * (nadd - 1);
    if (!unpack_prefix(n3, call)) return 1;
    const int ndbm = ntype - nadd;
    snprintf(out, out_len, "%s %2d", call, ndbm);
    const int nv = ndbm % 10;
    return (nv == 0 || nv == 3 || nv == 7 || nv == 10) ? 0 : 1;
  }
  if (ntype < 0) {
    // hashed callsign + 6-character locator: needs the receiver's hash table
    const int ndbm = -(ntype + 1);
    char grid6[8];
    char six[7];
    snprintf(six, sizeof(six), "%-6s", call);
    grid6[0] = six[5]; memcpy(grid6 + 1, six, 5); grid6[6] = 0;
    snprintf(out, out_len, "<...> %s %2d", grid6, ndbm);
    return ntype == -64 ? 1 : 0;
  }
  return 1;
}

extern "C" int uwspr_c2_read(const char *path, float *iq, double *dial_freq, int32_t *type) {
  if (!path || !iq) return UWSPR_ERR_ARG;
  FILE *fp = fopen(path, "rb");
  if (!fp) return UWSPR_ERR_ARG;
  char name[14];
  int32_t ntrmin = 0;
  double dfreq = 0.0;
  int ok = fread(name, 1, 14, fp) == 14 && fread(&ntrmin, 4, 1, fp) == 1 && fread(&dfreq, 8, 1, fp) == 1;
  size_t nread = ok ? fread(iq, sizeof(float), 2 * 45000, fp) : 0;
  fclose(fp);
  if (nread != 2 * 45000) return UWSPR_ERR_ARG;  // "invalid number of samples"
  for (int i = 0; i < 45000; i++) iq[2 * i + 1] = -iq[2 * i + 1];  // cc:91
  if (dial_freq) *dial_freq = dfreq;
  if (type) *type = ntrmin;
  return UWSPR_OK;
}
