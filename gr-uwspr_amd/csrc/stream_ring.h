// stream_ring.h -- the tail of a 375 S/s (I,Q) stream kept in device memory so that the overlapping
// frames sliding_window_stream_to_pdu cuts from it (lib/sliding_window_stream_to_pdu_impl.cc:113-135:
// frame k = samples [k hop, k hop + fl)) can be read IN PLACE: frame k of a take starts at
// view + 2 * hop * k floats, nothing is cut or copied per frame, and a take's unique samples
// ((B-1) hop + fl of them) are what the kernels' working set is.
//
// Uploads run on the ring's own copy stream, so they overlap the search kernels of the batches before:
//   producer:  append()  H2D on the copy stream (page-locked source: one DMA; pageable: staged through
//                        two page-locked halves), ev_up recorded behind it
//   consumer:  view()    makes its stream wait for ev_up, returns the in-place pointer, consumes k hop
//                        samples; reader_done() tells the ring when the kernels reading a view have
//                        been enqueued (an event on the consumer's stream)
// Memory: two linear buffers of `cap` samples.  Appends go behind the unconsumed samples; when the
// current buffer is full the unconsumed tail (< fl + a take) moves to the front of the other buffer
// (rare: cap is several takes) after every reader of THAT buffer has finished.
#pragma once

#include <hip/hip_runtime.h>
#include <string.h>

#include <vector>

namespace uwspr {

struct stream_ring {
  int fl = 0, hop = 0, maxf = 0;
  float *buf[2] = {nullptr, nullptr};
  size_t cap = 0;                      // samples per buffer
  int cur = 0;
  size_t base = 0, have = 0;           // unconsumed samples: buf[cur][base, base + have)
  long long pos = 0;                   // stream index of buf[cur][base]
  hipStream_t copy = nullptr;
  hipEvent_t ev_up = nullptr;          // behind the last append / compaction
  bool up_pending = false;
  std::vector<hipEvent_t> readers[2];  // events after which buf[k] is no longer read
  // page-locked staging for pageable sources
  static constexpr size_t PIECE = 4u << 20;
  char *pin = nullptr;
  hipEvent_t pin_ev[2] = {nullptr, nullptr};
  bool pin_busy[2] = {false, false};
  int pin_next = 0;
  bool last_direct = false;            // the last append DMAs straight from the caller's (page-locked) buffer
  hipError_t err = hipSuccess;

  bool is_open() const { return buf[0] != nullptr; }

  void close() {
    if (copy) (void)hipStreamSynchronize(copy);
    for (int k = 0; k < 2; k++) { if (buf[k]) (void)hipFree(buf[k]); buf[k] = nullptr; readers[k].clear(); }
    if (pin) { (void)hipHostFree(pin); pin = nullptr; }
    for (int k = 0; k < 2; k++) if (pin_ev[k]) { (void)hipEventDestroy(pin_ev[k]); pin_ev[k] = nullptr; }
    if (ev_up) { (void)hipEventDestroy(ev_up); ev_up = nullptr; }
    if (copy) { (void)hipStreamDestroy(copy); copy = nullptr; }
    cap = 0; have = 0; base = 0; up_pending = false;
  }

  // takes_of_slack: how many full takes fit behind one another before the tail has to move
  bool open(int fl_, int hop_, int max_frames, int takes_of_slack = 6) {
    close();
    fl = fl_; hop = hop_; maxf = max_frames;
    cap = (size_t)takes_of_slack * max_frames * hop + fl;
    for (int k = 0; k < 2; k++)
      if ((err = hipMalloc((void **)&buf[k], cap * 2 * sizeof(float))) != hipSuccess) { close(); return false; }
    if ((err = hipStreamCreateWithFlags(&copy, hipStreamNonBlocking)) != hipSuccess) { close(); return false; }
    if ((err = hipEventCreateWithFlags(&ev_up, hipEventDisableTiming)) != hipSuccess) { close(); return false; }
    cur = 0; base = 0; have = 0; pos = 0; up_pending = false; pin_next = 0;
    pin_busy[0] = pin_busy[1] = false;
    return true;
  }

  int ready() const {
    if (have < (size_t)fl) return 0;
    const size_t n = (have - fl) / hop + 1;
    return (int)(n < (size_t)maxf ? n : (size_t)maxf);
  }
  size_t room() const { return 2 * cap > 0 ? cap - have : 0; }   // samples an append can take (after moving the tail)

  // drop what is buffered.  Kernels may still be reading views of the current buffer, so the next append
  // goes through make_room() (base = cap: nothing fits) to the OTHER buffer, behind its readers.
  void reset(long long p) { have = 0; base = cap; pos = p; }

  // make space for n more samples behind the unconsumed ones
  bool make_room(size_t n) {
    if (base + have + n <= cap) return true;
    if (have + n > cap) return false;
    const int other = cur ^ 1;
    for (hipEvent_t e : readers[other])
      if ((err = hipStreamWaitEvent(copy, e, 0)) != hipSuccess) return false;
    readers[other].clear();
    if (have && (err = hipMemcpyAsync(buf[other], buf[cur] + 2 * base, have * 2 * sizeof(float),
                                      hipMemcpyDeviceToDevice, copy)) != hipSuccess) return false;
    cur = other; base = 0;
    return true;
  }

  // src: host (page-locked or pageable) when !on_device, else device memory that `src_ready` (may be
  // null: already complete) orders.  Returns with the transfer enqueued on the copy stream; a
  // page-locked source must stay unmodified until wait_uploads().
  bool append(const float *src, size_t n, bool on_device, hipEvent_t src_ready = nullptr) {
    if (n == 0) return true;
    last_direct = false;
    if (!make_room(n)) { if (err == hipSuccess) err = hipErrorOutOfMemory; return false; }
    float *dst = buf[cur] + 2 * (base + have);
    const size_t bytes = n * 2 * sizeof(float);
    if (on_device) {
      if (src_ready && (err = hipStreamWaitEvent(copy, src_ready, 0)) != hipSuccess) return false;
      if ((err = hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToDevice, copy)) != hipSuccess) return false;
    } else {
      hipPointerAttribute_t at;
      const bool locked = hipPointerGetAttributes(&at, src) == hipSuccess && at.type == hipMemoryTypeHost;
      if (!locked) (void)hipGetLastError();   // an ordinary pointer is "invalid value" to the query: not an error here
      if (locked) {
        last_direct = true;
        if ((err = hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, copy)) != hipSuccess) return false;
      } else {
        if (!pin) {
          if ((err = hipHostMalloc((void **)&pin, 2 * PIECE, hipHostMallocDefault)) != hipSuccess) { pin = nullptr; return false; }
          for (int k = 0; k < 2; k++)
            if ((err = hipEventCreateWithFlags(&pin_ev[k], hipEventDisableTiming)) != hipSuccess) return false;
        }
        size_t off = 0;
        while (off < bytes) {
          const size_t m = bytes - off < PIECE ? bytes - off : PIECE;
          const int h = pin_next;
          pin_next ^= 1;
          if (pin_busy[h] && (err = hipEventSynchronize(pin_ev[h])) != hipSuccess) return false;
          memcpy(pin + (size_t)h * PIECE, (const char *)src + off, m);
          if ((err = hipMemcpyAsync((char *)dst + off, pin + (size_t)h * PIECE, m, hipMemcpyHostToDevice, copy)) != hipSuccess) return false;
          if ((err = hipEventRecord(pin_ev[h], copy)) != hipSuccess) return false;
          pin_busy[h] = true;
          off += m;
        }
      }
    }
    have += n;
    if ((err = hipEventRecord(ev_up, copy)) != hipSuccess) return false;
    up_pending = true;
    return true;
  }

  bool wait_uploads() {
    if (up_pending && (err = hipEventSynchronize(ev_up)) != hipSuccess) return false;
    up_pending = false;
    return true;
  }

  // The next k frames in place; `consumer` is the stream whose kernels will read them.
  // *bufidx = which buffer they live in (for reader_done).
  bool view(int k, hipStream_t consumer, const float **frames, long long *first_pos, int *bufidx) {
    if (k <= 0 || k > ready()) { err = hipErrorInvalidValue; return false; }
    if ((err = hipStreamWaitEvent(consumer, ev_up, 0)) != hipSuccess) return false;
    *frames = buf[cur] + 2 * base;
    if (first_pos) *first_pos = pos;
    if (bufidx) *bufidx = cur;
    const size_t used = (size_t)k * hop;
    base += used; have -= used; pos += (long long)used;
    return true;
  }
  void reader_done(int bufidx, hipEvent_t e) {
    for (hipEvent_t q : readers[bufidx]) if (q == e) return;   // re-recorded: a wait sees its newest record
    readers[bufidx].push_back(e);
  }
};

}  // namespace uwspr
