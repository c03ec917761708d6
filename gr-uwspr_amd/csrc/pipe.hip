// pipe.hip -- the pipelined end-to-end decoder of include/uwspr_hip.h (uwspr_pipe_*): host code only,
// built on the C ABI's own entry points, so what it produces is by construction what the sequential
// calls produce -- only the ORDER IN TIME of the stages of consecutive batches differs:
//
//   producer thread (the caller)      copy stream          lane k's HIP stream           coordinator + pool
//   acquire / fill / commit  ------>  H2D of new samples
//   batch complete: view in place --------------------->   K1 K2 K3, schedule (try 0),
//                                                          D2H of records, event  ---->  wait event
//   (next batch: lane k+1 ...)                                                           Fano on try 0 (pool; one
//                                                                                        coordinator per lane)
//                                                          resume (tries 1..16) <------  for what did not decode
//                                                          D2H, event            ---->   Fano on tries 1..16
//                                                                                        results in frame order
//
// Reference: the flowgraph chain sliding_window_stream_to_pdu -> FDR -> sync_and_demodulate
// (lib/sliding_window_stream_to_pdu_impl.cc:97-138, lib/FDR_impl.cc:214-456,
// lib/sync_and_demodulate_impl.cc:315-534; the lazy tries are cc:457-490's early exit).
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <atomic>
#include <chrono>
#include <condition_variable>
#include <deque>
#include <mutex>
#include <thread>
#include <vector>

#include "host_pool.h"
#include "uwspr_internal.h"

using namespace uwspr;

namespace {

constexpr int kPipeStreams = 3;   // HIP streams the lanes share (see uwspr_pipe_open)

struct pipe_lane {
  uwspr_ctx *ctx = nullptr;
  hipStream_t stream = nullptr;
  uwspr_candidate *d_cands = nullptr; int32_t *d_npk = nullptr; uwspr_demod_out *d_out = nullptr; uint8_t *d_need = nullptr;
  uwspr_candidate *h_cands = nullptr; int32_t *h_npk = nullptr; uwspr_demod_out *h_out = nullptr; uint8_t *h_need = nullptr;
  hipEvent_t ev_done = nullptr;
  // the batch in flight
  bool busy = false;       // taken by the producer, until its records have been emitted
  bool launched = false;   // its GPU work is enqueued: a coordinator may pick it up
  bool claimed = false;    // a coordinator has
  double host_since = 0.0; // > 0: its first GPU pass is complete and it has been in its host tail since then (now_s())
  int64_t seq = 0;         // batch number (emission order)
  std::vector<uwspr_decode> recs;   // the batch's records, built by the coordinator, emitted in batch order
  int B = 0, stride = 0;
  const float *frames = nullptr;
  int64_t frame0 = 0, pos0 = -1;
  std::vector<uint8_t> dec;          // [B*per] decoded flags
  std::vector<int32_t> idt;          // [B*per]
  std::vector<int8_t> msg;           // [B*per][7]
  std::vector<int> first;            // records whose first pass goes to the pool
  std::vector<int> redo;             // records resumed
  std::vector<int> tasks;            // (record << 5) | try: the resumed tries, decoded in parallel
  std::vector<uint8_t> task_ok;
  std::vector<int8_t> task_msg;
};

double now_s() {
  return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

}  // namespace

struct uwspr_pipe {
  uwspr_params p;
  uwspr_pipe_opts o;
  int device = 0, fl = 0, maxfreqs = 0, per = 1;
  char err[512];
  // Sticky status of the first RUNTIME failure (HIP, a lane's context), written by coordinators and the producer, read
  // by both without the lock.  Argument errors are not sticky: the call that made them returns UWSPR_ERR_ARG (with
  // its message), nothing in flight is harmed and the pipe goes on.
  std::atomic<int> failed{0};
  std::atomic<int64_t> inject_seq{-1}; std::atomic<int> inject_where{0};   // uwspr_pipe_inject_failure (tests): read by the coordinators and the producer without q->m

  std::vector<pipe_lane> lanes;
  double spare_after = 2.5e-3;   // seconds of host tail after which a spare lane may open (take_lane)
  int next_lane = 0;
  int64_t next_frame = 0;

  stream_ring ring;
  static constexpr int NSTAGE = 4;
  float *h_stage[NSTAGE] = {nullptr, nullptr, nullptr, nullptr};
  hipEvent_t stage_ev[NSTAGE] = {nullptr, nullptr, nullptr, nullptr};
  bool stage_busy[NSTAGE] = {false, false, false, false};
  int stage_next = 0, stage_cur = -1;
  size_t stage_samples = 0;

  std::mutex m;
  std::condition_variable cv_lane, cv_work, cv_done, cv_turn;
  std::deque<uwspr_decode> done;
  bool stop = false;
  int64_t next_seq = 0, emit_seq = 0;     // batches launched / emitted (records leave in batch order)
  std::vector<std::thread> coords;         // one coordinator per lane: the host tails of consecutive batches overlap
  host_pool *pool = nullptr;
  bool own_pool = false;
  uwspr_pipe_stats st;
};

static int pfail(uwspr_pipe *q, int status, const char *fmt, ...) {
  if (q) {
    std::lock_guard<std::mutex> lk(q->m);
    if (!q->failed.load()) {
      va_list ap;
      va_start(ap, fmt);
      vsnprintf(q->err, sizeof(q->err), fmt, ap);
      va_end(ap);
      q->failed.store(status);
    }
  }
  if (q) q->cv_done.notify_all();   // (a collector waiting for records learns of the failure)
  return status;
}
// an argument error of the calling thread: message, status, nothing sticky
static int parg(uwspr_pipe *q, const char *fmt, ...) {
  if (q) {
    std::lock_guard<std::mutex> lk(q->m);
    if (!q->failed.load()) {
      va_list ap;
      va_start(ap, fmt);
      vsnprintf(q->err, sizeof(q->err), fmt, ap);
      va_end(ap);
    }
  }
  return UWSPR_ERR_ARG;
}

#define PHIP(q, call)                                                                               \
  do {                                                                                              \
    hipError_t e_ = (call);                                                                         \
    if (e_ != hipSuccess) return pfail((q), UWSPR_ERR_HIP, "%s failed: %s (%s:%d)", #call,          \
                                       hipGetErrorString(e_), __FILE__, __LINE__);                  \
  } while (0)

// ---- coordinator: finishes the batches in launch order ------------------------------------------
static int finish_batch(uwspr_pipe *q, pipe_lane &L) {
  const int per = q->per, B = L.B, nrec = B * per;
  L.recs.clear();   // (a failure below emits nothing for this batch)
  double t0 = now_s();
  PHIP(q, hipEventSynchronize(L.ev_done));
  if (q->inject_where.load() == 1 && L.seq == q->inject_seq.load()) return pfail(q, UWSPR_ERR_HIP, "injected failure in the host tail of batch %lld", (long long)L.seq);
  double t1 = now_s();
  {
    std::lock_guard<std::mutex> lk(q->m);
    L.host_since = t1;   // the host tail of this batch starts: the producer may open a spare lane if it lasts (take_lane)
  }
  auto valid = [&](int i) { const int b = i / per, j = i - b * per; return j < L.h_npk[b] && j < q->maxfreqs; };
  // cc:457-490 on what the first pass produced (try 0 alone in the lazy flow)
  std::atomic<long long> calls(0), fails(0);
  // only the records that are worth a try go to the pool (a quiet stream has almost none: no wake-ups for it)
  L.first.clear();
  for (int i = 0; i < nrec; i++) {
    L.dec[i] = 0; L.idt[i] = -1;
    memset(&L.msg[7 * (size_t)i], 0, 7);
    if (valid(i) && L.h_out[i].worth_a_try) L.first.push_back(i);
  }
  auto first_pass = [&](int k) {
    const int i = L.first[k];
    int32_t idt = -1;
    int nc = 0;
    const int r = decode_candidate_from(&L.h_out[i], 0, &L.msg[7 * (size_t)i], &idt, &nc);
    if (nc) { calls.fetch_add(nc, std::memory_order_relaxed); fails.fetch_add(nc - r, std::memory_order_relaxed); }
    if (!r) memset(&L.msg[7 * (size_t)i], 0, 7);
    L.dec[i] = (uint8_t)r;
    L.idt[i] = idt;
  };
  if (L.first.size() <= 2) for (size_t k = 0; k < L.first.size(); k++) first_pass((int)k);
  else q->pool->run((int)L.first.size(), q->o.host_threads, first_pass);
  double t2 = now_s(), t3 = t2;
  L.redo.clear();
  if (!q->o.eager) {
    for (int i = 0; i < nrec; i++) {
      const bool need = valid(i) && L.h_out[i].worth_a_try && !L.dec[i];
      L.h_need[i] = need ? 1 : 0;
      if (need) L.redo.push_back(i);
    }
  }
  if (!L.redo.empty()) {
    // the other 16 tries of those candidates, then Fano from try 1 on
    PHIP(q, hipMemcpyAsync(L.d_need, L.h_need, (size_t)nrec, hipMemcpyHostToDevice, L.stream));
    int rc = uwspr_demod_resume(L.ctx, L.frames, B, UWSPR_DEVICE, L.d_need, per, L.d_out);
    if (rc) return pfail(q, rc, "uwspr_demod_resume: %s", uwspr_last_error(L.ctx));
    if (L.redo.size() * 4 > (size_t)nrec) {
      PHIP(q, hipMemcpyAsync(L.h_out, L.d_out, (size_t)nrec * sizeof(uwspr_demod_out), hipMemcpyDeviceToHost, L.stream));
    } else {
      for (int i : L.redo)
        PHIP(q, hipMemcpyAsync(&L.h_out[i], &L.d_out[i], sizeof(uwspr_demod_out), hipMemcpyDeviceToHost, L.stream));
    }
    PHIP(q, hipEventRecord(L.ev_done, L.stream));
    PHIP(q, hipEventSynchronize(L.ev_done));
    t3 = now_s();
    // The reference walks a candidate's tries one after the other and stops at the first that decodes
    // (cc:457-490).  A try that does not decode runs Fano to its 10000-cycles-per-bit time-out (~4 ms of one
    // core), so a candidate nothing decodes would hold ONE thread for 16 time-outs while the pool idles.
    // The tries are independent decodes: all (candidate, try) pairs go to the pool at once and the first
    // try in the reference's order that decoded is the answer -- the same message and idt; the tries after
    // a decoding one are work the reference would not have done.
    L.tasks.clear();
    for (int i : L.redo)
      for (int idt = 1; idt < UWSPR_NJIG; idt++) L.tasks.push_back((i << 5) | idt);
    L.task_ok.assign(L.tasks.size(), 0);
    L.task_msg.resize(L.tasks.size() * 7);
    q->pool->run((int)L.tasks.size(), q->o.host_threads, [&](int t) {
      const int i = L.tasks[t] >> 5, idt = L.tasks[t] & 31;
      const int r = decode_try(&L.h_out[i], idt, &L.task_msg[7 * (size_t)t]);
      if (r >= 0) calls.fetch_add(1, std::memory_order_relaxed);
      if (r == 0) fails.fetch_add(1, std::memory_order_relaxed);
      L.task_ok[t] = (uint8_t)(r > 0);
    });
    for (size_t t = 0; t < L.tasks.size(); t++) {   // tasks are in (candidate, try) order
      const int i = L.tasks[t] >> 5, idt = L.tasks[t] & 31;
      if (L.task_ok[t] && !L.dec[i]) {
        L.dec[i] = 1; L.idt[i] = idt;
        memcpy(&L.msg[7 * (size_t)i], &L.task_msg[7 * t], 7);
      }
    }
  }
  double t4 = now_s();
  int ncand = 0, ndec = 0;
  {
    for (int b = 0; b < B; b++) {
      for (int j = 0; j < per; j++) {
        const int i = b * per + j;
        if (!valid(i)) continue;
        uwspr_decode d;
        memset(&d, 0, sizeof(d));
        d.frame = L.frame0 + b;
        d.stream_pos = L.pos0 >= 0 ? L.pos0 + (int64_t)b * L.stride : -1;
        d.cand = j; d.npk = L.h_npk[b];
        d.coarse = L.h_cands[(size_t)b * per + j];
        const uwspr_demod_out &o = L.h_out[i];
        d.f1 = o.f1; d.drift1 = o.drift1; d.sync1 = o.sync1; d.shift1 = o.shift1; d.worth_a_try = o.worth_a_try;
        d.decoded = L.dec[i]; d.idt = L.idt[i];
        memcpy(d.message, &L.msg[7 * (size_t)i], 7);
        L.recs.push_back(d);
        ncand++; ndec += L.dec[i];
      }
    }
  }
  {
    std::lock_guard<std::mutex> lk(q->m);
    q->st.frames += B; q->st.batches += 1; q->st.candidates += ncand; q->st.decoded += ndec;
    q->st.resumed += (int64_t)L.redo.size();
    q->st.fano_calls += calls.load(); q->st.fano_timeouts += fails.load();   // (tries run after a decoding one count as calls)
    q->st.gpu_wait_s += (t1 - t0);
    q->st.fano_s += (t2 - t1) + (t4 - t3);
    q->st.resume_s += (t3 - t2);
  }
  return UWSPR_OK;
}

// A coordinator takes the oldest launched batch nobody has taken yet, finishes it and emits its records when the
// batches before it have been emitted.  ncoords threads run this loop (default: one per lane, so that the host
// tails of consecutive batches overlap).
static void coordinator(uwspr_pipe *q) {
  (void)hipSetDevice(q->device);
  for (;;) {
    pipe_lane *Lp = nullptr;
    {
      std::unique_lock<std::mutex> lk(q->m);
      q->cv_work.wait(lk, [&]() {
        Lp = nullptr;
        for (auto &L : q->lanes)
          if (L.launched && !L.claimed && (!Lp || L.seq < Lp->seq)) Lp = &L;
        return q->stop || Lp != nullptr;
      });
      if (!Lp) return;   // stop, nothing left to take
      Lp->claimed = true;
    }
    pipe_lane &L = *Lp;
    (void)finish_batch(q, L);   // a failure is sticky in q->failed; the lane is released either way
    {
      std::unique_lock<std::mutex> lk(q->m);
      q->cv_turn.wait(lk, [&]() { return q->emit_seq == L.seq; });
      for (const uwspr_decode &d : L.recs) q->done.push_back(d);
      q->emit_seq++;
      L.launched = false; L.claimed = false; L.host_since = 0.0;
      L.busy = false;
    }
    q->cv_turn.notify_all();
    q->cv_lane.notify_all();
    q->cv_done.notify_all();
  }
}

// ---- producer side --------------------------------------------------------------------------------
// The first kPipeStreams lanes (one per HIP stream) take the batches in turn: that is what a GPU-bound stream wants
// (more batches in flight only cost it cache and 6 % of its rate).  The lanes beyond them are SPARES for a stream whose
// host tail is the bottleneck -- Fano time-outs, 4 ms of a core each: a spare is opened only while every base lane is busy
// and one of them has been in its host tail (first GPU pass complete) for longer than kSpareAfter (2.5 ms).
constexpr double kSpareAfter = 2.5e-3;   // seconds (a Fano time-out is ~4 ms; the host tail of a batch that decodes at once ~0.3 ms)
// (uwspr_pipe_opts::spare_after_us: tests open the spares at once with 1)
static pipe_lane *take_lane(uwspr_pipe *q) {
  std::unique_lock<std::mutex> lk(q->m);
  const int n = (int)q->lanes.size(), base = n < kPipeStreams ? n : kPipeStreams;
  pipe_lane *pick = nullptr;
  for (;;) {
    for (int k = 0; k < base && !pick; k++) {
      const int idx = (q->next_lane + k) % base;
      if (!q->lanes[idx].busy) { pick = &q->lanes[idx]; q->next_lane = (idx + 1) % base; }
    }
    if (pick) break;
    if (n > base) {
      const double now = now_s();
      bool slow_tail = false;
      for (int k = 0; k < base; k++) slow_tail |= q->lanes[k].host_since > 0.0 && now - q->lanes[k].host_since > q->spare_after;
      if (slow_tail)
        for (int k = base; k < n && !pick; k++) if (!q->lanes[k].busy) pick = &q->lanes[k];
      if (pick) break;
      q->cv_lane.wait_for(lk, std::chrono::microseconds(500));   // (a host tail grows old without anybody notifying)
    } else {
      q->cv_lane.wait(lk);
    }
  }
  pick->busy = true;
  return pick;
}

static int launch(uwspr_pipe *q, pipe_lane &L, const float *frames, int B, int stride, int64_t pos0, int ringbuf) {
  const int per = q->per;
  L.B = B; L.stride = stride; L.frames = frames; L.pos0 = pos0; L.frame0 = q->next_frame;
  q->next_frame += B;
  if (q->inject_where.load() == 0 && q->next_seq == q->inject_seq.load())
    return pfail(q, UWSPR_ERR_HIP, "injected failure at the launch of batch %lld", (long long)q->next_seq);
  int rc = uwspr_set_frame_stride(L.ctx, stride);
  if (!rc) rc = uwspr_set_tries(L.ctx, q->o.eager ? UWSPR_NJIG : 1);
  if (!rc) rc = uwspr_pipeline_batch(L.ctx, frames, B, UWSPR_DEVICE, per, L.d_cands, L.d_npk, L.d_out);
  if (rc) return pfail(q, rc, "uwspr_pipeline_batch: %s", uwspr_last_error(L.ctx));
  PHIP(q, hipMemcpyAsync(L.h_npk, L.d_npk, (size_t)B * sizeof(int32_t), hipMemcpyDeviceToHost, L.stream));
  PHIP(q, hipMemcpy2DAsync(L.h_cands, (size_t)per * sizeof(uwspr_candidate), L.d_cands,
                           (size_t)q->maxfreqs * sizeof(uwspr_candidate), (size_t)per * sizeof(uwspr_candidate), B,
                           hipMemcpyDeviceToHost, L.stream));
  PHIP(q, hipMemcpyAsync(L.h_out, L.d_out, (size_t)B * per * sizeof(uwspr_demod_out), hipMemcpyDeviceToHost, L.stream));
  PHIP(q, hipEventRecord(L.ev_done, L.stream));
  if (ringbuf >= 0) q->ring.reader_done(ringbuf, L.ev_done);
  {
    std::lock_guard<std::mutex> lk(q->m);
    L.seq = q->next_seq++;
    L.launched = true;
  }
  q->cv_work.notify_all();
  return UWSPR_OK;
}

static void release_lane(uwspr_pipe *q, pipe_lane &L) {   // a launch that failed before it was queued
  {
    std::lock_guard<std::mutex> lk(q->m);
    L.busy = false;
  }
  q->cv_lane.notify_all();
}

static int launch_from_ring(uwspr_pipe *q, int k) {
  pipe_lane *L = take_lane(q);
  const float *frames = nullptr;
  long long pos = 0;
  int buf = 0;
  if (!q->ring.view(k, L->stream, &frames, &pos, &buf)) {
    release_lane(q, *L);
    return pfail(q, UWSPR_ERR_HIP, "stream view: %s", hipGetErrorString(q->ring.err));
  }
  const int rc = launch(q, *L, frames, k, q->ring.hop, pos, buf);
  if (rc) release_lane(q, *L);
  return rc;
}

extern "C" const char *uwspr_pipe_last_error(const uwspr_pipe *q) { return q ? q->err : "null pipe"; }

extern "C" void uwspr_pipe_close(uwspr_pipe *q) {
  if (!q) return;
  {
    std::lock_guard<std::mutex> lk(q->m);
    q->stop = true;
  }
  q->cv_work.notify_all();
  for (auto &t : q->coords) if (t.joinable()) t.join();
  (void)hipSetDevice(q->device);
  for (auto &L : q->lanes) if (L.stream) (void)hipStreamSynchronize(L.stream);
  for (auto &L : q->lanes)   // lanes beyond the stream count ran on another lane's stream: back to their own before any is destroyed
    if (L.ctx && L.stream != L.ctx->own_stream) { (void)uwspr_set_stream(L.ctx, nullptr); L.stream = L.ctx->own_stream; }
  for (auto &L : q->lanes) {
    void *dev[] = {L.d_cands, L.d_npk, L.d_out, L.d_need};
    for (void *b : dev) if (b) (void)hipFree(b);
    void *host[] = {L.h_cands, L.h_npk, L.h_out, L.h_need};
    for (void *b : host) if (b) (void)hipHostFree(b);
    if (L.ev_done) (void)hipEventDestroy(L.ev_done);
    if (L.ctx) uwspr_ctx_destroy(L.ctx);
  }
  q->ring.close();
  for (int k = 0; k < uwspr_pipe::NSTAGE; k++) {
    if (q->h_stage[k]) (void)hipHostFree(q->h_stage[k]);
    if (q->stage_ev[k]) (void)hipEventDestroy(q->stage_ev[k]);
  }
  if (q->own_pool) delete q->pool;
  delete q;
}

extern "C" int uwspr_pipe_open(const uwspr_params *p, int device, const uwspr_pipe_opts *o, uwspr_pipe **out) {
  if (!p || !o || !out) return UWSPR_ERR_ARG;
  *out = nullptr;
  uwspr_pipe *q = new (std::nothrow) uwspr_pipe();
  if (!q) return UWSPR_ERR_NOMEM;
  memset(q->err, 0, sizeof(q->err));
  memset(&q->st, 0, sizeof(q->st));
  q->p = *p; q->o = *o; q->device = device; q->fl = p->fl; q->maxfreqs = p->maxfreqs;
  *out = q;   // handed back even on failure so uwspr_pipe_last_error() can be read
  if (q->o.batch_frames <= 0) q->o.batch_frames = 256;
  if (q->o.max_per_frame <= 0) q->o.max_per_frame = 1;
  if (q->o.lanes <= 0) q->o.lanes = 9;   // three streams + six spares (take_lane)
  if (q->o.lanes > 12) q->o.lanes = 12;
  q->spare_after = q->o.spare_after_us > 0 ? 1e-6 * (double)q->o.spare_after_us : kSpareAfter;
  if (q->o.hop <= 0) q->o.hop = p->fl;
  if (q->o.hop > p->fl) return pfail(q, UWSPR_ERR_ARG, "hop=%d > fl=%d", q->o.hop, p->fl);
  q->per = q->o.max_per_frame < p->maxfreqs ? q->o.max_per_frame : p->maxfreqs;
  const int Bm = q->o.batch_frames, per = q->per;
  q->lanes.resize(q->o.lanes);
  for (auto &L : q->lanes) {
    const int rc = uwspr_ctx_create(p, device, &L.ctx);
    if (rc) return pfail(q, rc, "uwspr_ctx_create: %s", L.ctx ? uwspr_last_error(L.ctx) : uwspr_status_string(rc));
    L.stream = L.ctx->own_stream;
    // At most kPipeStreams HIP streams: lane k >= kPipeStreams launches on the stream of lane k % kPipeStreams (a
    // fourth stream shares a hardware queue with another and the GPU-bound rate drops: 4 / 6 lanes 734 / 833 k
    // against 861 k for 3, round-3 probe).  The extra lanes are batches in flight on the HOST side -- their own
    // buffers and coordinator -- which is what a stream with many Fano time-outs needs (busy stream: 39 k frames/s
    // with 3 lanes, 49 k with 6 on 3 streams).
    const int lane_idx = (int)(&L - &q->lanes[0]);
    if (lane_idx >= kPipeStreams) {
      const int rcs = uwspr_set_stream(L.ctx, q->lanes[lane_idx % kPipeStreams].ctx->own_stream);
      if (rcs) return pfail(q, rcs, "uwspr_set_stream: %s", uwspr_last_error(L.ctx));
      L.stream = q->lanes[lane_idx % kPipeStreams].ctx->own_stream;
    }
    // schedule form: with three or more batches in flight the staged launches leave room for the other
    // lanes' kernels and win (861 k against 773 k decoded frames/s at 3 lanes); alone or in pairs the fused
    // kernel does (tools/pipe_lanes_probe.py).  A "sched" option in UWSPR_OPTIONS still decides when set.
    if (q->o.sched_form == 1) (void)uwspr_set_option(L.ctx, "sched", 1);
    else if (q->o.sched_form == 2) (void)uwspr_set_option(L.ctx, "sched", 0);
    else if (!L.ctx->opt_set[UWSPR_OPT_SCHED]) (void)uwspr_set_option(L.ctx, "sched", q->o.lanes < kPipeStreams ? 1 : 0);
    PHIP(q, hipMalloc((void **)&L.d_cands, (size_t)Bm * p->maxfreqs * sizeof(uwspr_candidate)));
    PHIP(q, hipMalloc((void **)&L.d_npk, (size_t)Bm * sizeof(int32_t)));
    PHIP(q, hipMalloc((void **)&L.d_out, (size_t)Bm * per * sizeof(uwspr_demod_out)));
    PHIP(q, hipMalloc((void **)&L.d_need, (size_t)Bm * per));
    PHIP(q, hipHostMalloc((void **)&L.h_cands, (size_t)Bm * per * sizeof(uwspr_candidate), hipHostMallocDefault));
    PHIP(q, hipHostMalloc((void **)&L.h_npk, (size_t)Bm * sizeof(int32_t), hipHostMallocDefault));
    PHIP(q, hipHostMalloc((void **)&L.h_out, (size_t)Bm * per * sizeof(uwspr_demod_out), hipHostMallocDefault));
    PHIP(q, hipHostMalloc((void **)&L.h_need, (size_t)Bm * per, hipHostMallocDefault));
    // the coordinator sleeps on this event (it does not spin: the cores belong to the Fano pool)
    PHIP(q, hipEventCreateWithFlags(&L.ev_done, hipEventDisableTiming | hipEventBlockingSync));
    L.dec.resize((size_t)Bm * per); L.idt.resize((size_t)Bm * per); L.msg.resize((size_t)Bm * per * 7);
  }
  // pushed streams: the device ring and the page-locked staging buffers are made by the first acquire (open_ingest):
  // a pipe that only takes device frames (uwspr_pipe_submit_device) never pays their 2 x 83 MB of HBM + 4 x 6.9 MB page-locked (hop 3375; 2.2 GB + 0.37 GB at hop 0 = the frame length)
  q->stage_samples = (size_t)Bm * q->o.hop;
  if (q->o.host_threads <= 0) q->o.host_threads = host_cpu_share() > 3 ? host_cpu_share() - 2 : 1;
  q->pool = &host_pool::shared();   // the process-wide pool; this pipe's jobs use host_threads of it
  const int ncoords = (int)q->lanes.size();
  for (int k = 0; k < ncoords; k++) q->coords.emplace_back(coordinator, q);
  return UWSPR_OK;
}

// the device ring (a few batches of slack beyond what can be in flight) and the page-locked staging buffers
static int open_ingest(uwspr_pipe *q) {
  if (q->ring.is_open()) return UWSPR_OK;
  if (!q->ring.open(q->fl, q->o.hop, q->o.batch_frames, q->o.lanes + 3))
    return pfail(q, UWSPR_ERR_NOMEM, "stream ring: %s", hipGetErrorString(q->ring.err));
  for (int k = 0; k < uwspr_pipe::NSTAGE; k++) {
    PHIP(q, hipHostMalloc((void **)&q->h_stage[k], q->stage_samples * 2 * sizeof(float), hipHostMallocDefault));
    PHIP(q, hipEventCreateWithFlags(&q->stage_ev[k], hipEventDisableTiming));
  }
  return UWSPR_OK;
}

extern "C" int uwspr_pipe_inject_failure(uwspr_pipe *q, long long batch, int where) {
  if (!q || where < 0 || where > 1) return UWSPR_ERR_ARG;
  q->inject_where.store(where); q->inject_seq.store(batch);
  return UWSPR_OK;
}

extern "C" int uwspr_pipe_acquire(uwspr_pipe *q, int nsamples, float **iq) {
  if (!q || !iq) return UWSPR_ERR_ARG;
  if (const int f = q->failed.load()) return f;
  if (nsamples <= 0 || (size_t)nsamples > q->stage_samples)
    return parg(q, "uwspr_pipe_acquire(%d): at most %zu samples per piece", nsamples, q->stage_samples);
  (void)hipSetDevice(q->device);
  if (const int rc = open_ingest(q)) return rc;
  const int s = q->stage_next;
  if (q->stage_busy[s]) { PHIP(q, hipEventSynchronize(q->stage_ev[s])); q->stage_busy[s] = false; }
  q->stage_cur = s;
  *iq = q->h_stage[s];
  return UWSPR_OK;
}

extern "C" int uwspr_pipe_commit(uwspr_pipe *q, int nsamples) {
  if (!q) return UWSPR_ERR_ARG;
  if (const int f = q->failed.load()) return f;
  if (q->stage_cur < 0 || nsamples < 0 || (size_t)nsamples > q->stage_samples)
    return parg(q, "uwspr_pipe_commit(%d) without a matching uwspr_pipe_acquire", nsamples);
  (void)hipSetDevice(q->device);
  const int s = q->stage_cur;
  q->stage_cur = -1;
  if (nsamples > 0) {
    // room behind the unconsumed samples: launch what is complete first if the ring is full
    while (q->ring.have + (size_t)nsamples > q->ring.cap && q->ring.ready() > 0) {
      const int rc = launch_from_ring(q, q->ring.ready() < q->o.batch_frames ? q->ring.ready() : q->o.batch_frames);
      if (rc) return rc;
    }
    if (!q->ring.append(q->h_stage[s], (size_t)nsamples, false))
      return pfail(q, UWSPR_ERR_HIP, "stream upload: %s", hipGetErrorString(q->ring.err));
    PHIP(q, hipEventRecord(q->stage_ev[s], q->ring.copy));
    q->stage_busy[s] = true;
    q->stage_next = (s + 1) % uwspr_pipe::NSTAGE;
  }
  while (q->ring.ready() >= q->o.batch_frames) {
    const int rc = launch_from_ring(q, q->o.batch_frames);
    if (rc) return rc;
  }
  return UWSPR_OK;
}

extern "C" int uwspr_pipe_push(uwspr_pipe *q, const float *iq, int nsamples) {
  if (!q || (nsamples > 0 && !iq) || nsamples < 0) return UWSPR_ERR_ARG;
  size_t off = 0;
  while (off < (size_t)nsamples) {
    const size_t n = (size_t)nsamples - off < q->stage_samples ? (size_t)nsamples - off : q->stage_samples;
    float *dst = nullptr;
    int rc = uwspr_pipe_acquire(q, (int)n, &dst);
    if (rc) return rc;
    memcpy(dst, iq + 2 * off, n * 2 * sizeof(float));
    if ((rc = uwspr_pipe_commit(q, (int)n))) return rc;
    off += n;
  }
  return UWSPR_OK;
}

extern "C" int uwspr_pipe_submit_device(uwspr_pipe *q, const float *dev_frames, int B, int stride) {
  if (!q || !dev_frames) return UWSPR_ERR_ARG;
  if (const int f = q->failed.load()) return f;
  if (B <= 0 || B > q->o.batch_frames || stride < 0)
    return parg(q, "uwspr_pipe_submit_device: B=%d (1..%d) stride=%d", B, q->o.batch_frames, stride);
  (void)hipSetDevice(q->device);
  pipe_lane *L = take_lane(q);
  const int rc = launch(q, *L, dev_frames, B, stride > 0 ? stride : q->fl, -1, -1);
  if (rc) release_lane(q, *L);
  return rc;
}

// An option of every lane's context (uwspr_set_option).  Only while nothing is in flight -- the lanes read their
// options when they launch -- and not "sched": the pipe picks the schedule form by its lane count (opts.sched_form).
extern "C" int uwspr_pipe_set_option(uwspr_pipe *q, const char *name, int value) {
  if (!q || !name) return UWSPR_ERR_ARG;
  if (const int f = q->failed.load()) return f;
  if (!strcmp(name, "sched")) return parg(q, "uwspr_pipe_set_option: \"sched\" belongs to uwspr_pipe_opts.sched_form");
  // checked AND applied under q->m: a lane is marked busy under the same lock when a batch is launched on it, so no
  // batch starts between the check and the last lane's option (the producer is single-threaded by contract, the
  // coordinators are not)
  std::lock_guard<std::mutex> lk(q->m);
  for (auto &L : q->lanes)
    if (L.busy) { snprintf(q->err, sizeof(q->err), "uwspr_pipe_set_option(%s): batches in flight (flush first)", name); return UWSPR_ERR_ARG; }
  for (auto &L : q->lanes) {
    const int rc = uwspr_set_option(L.ctx, name, value);
    if (rc) {
      if (!q->failed.load()) snprintf(q->err, sizeof(q->err), "uwspr_pipe_set_option(%s, %d): %s", name, value, uwspr_last_error(L.ctx));
      return UWSPR_ERR_ARG;
    }
  }
  return UWSPR_OK;
}

extern "C" int uwspr_pipe_flush(uwspr_pipe *q) {
  if (!q) return UWSPR_ERR_ARG;
  (void)hipSetDevice(q->device);
  while (!q->failed.load() && q->ring.is_open() && q->ring.ready() > 0) {
    const int k = q->ring.ready() < q->o.batch_frames ? q->ring.ready() : q->o.batch_frames;
    const int rc = launch_from_ring(q, k);
    if (rc) return rc;
  }
  std::unique_lock<std::mutex> lk(q->m);
  q->cv_done.wait(lk, [&]() {
    for (auto &L : q->lanes) if (L.busy) return false;
    return true;
  });
  return q->failed.load();
}

extern "C" int uwspr_pipe_collect(uwspr_pipe *q, uwspr_decode *out, int cap, int wait) {
  if (!q || (cap > 0 && !out) || cap < 0) return UWSPR_ERR_ARG;
  std::unique_lock<std::mutex> lk(q->m);
  if (wait)
    q->cv_done.wait(lk, [&]() {
      if (!q->done.empty() || q->failed.load()) return true;
      for (auto &L : q->lanes) if (L.busy) return false;
      return true;
    });
  if (q->failed.load() && q->done.empty()) return q->failed.load();
  int n = 0;
  while (n < cap && !q->done.empty()) { out[n++] = q->done.front(); q->done.pop_front(); }
  return n;
}

extern "C" int uwspr_pipe_get_stats(uwspr_pipe *q, uwspr_pipe_stats *st) {
  if (!q || !st) return UWSPR_ERR_ARG;
  std::lock_guard<std::mutex> lk(q->m);
  *st = q->st;
  return UWSPR_OK;
}
