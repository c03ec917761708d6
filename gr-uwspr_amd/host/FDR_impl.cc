// FDR_impl.cc -- host block mirror of gr::uwspr::FDR on top of the C ABI.
// The handler keeps the reference's contract (lib/FDR_impl.cc:214-456): one
// samples PDU in on port "in", one candidates PDU out on port "out", the input
// sample vector re-used in the output (cc:450).  All arithmetic happens in
// uwspr_fdr_batch on the GPU.  Consecutive frames of a stream (PDUs that carry their
// stream position) are ingested through uwspr_stream_*: each sample is uploaded once.
#include <string.h>

#include <stdexcept>
#include <string>

#include "uwspr/FDR.h"

namespace gr {
namespace uwspr {

class FDR_impl : public FDR {
 public:
  FDR_impl(int fs, int fl, int spb, int maxdrift, int maxfreqs, int halfbandwidth, int cf,
           int threshold)
      : block("FDR"), d_ctx(nullptr), d_fl(fl), d_maxfreqs(maxfreqs), d_batch(1) {
    // lib/FDR_impl.cc:55-61
    message_port_register_in("in");
    set_msg_handler("in", [this](message_sptr m) { transform(std::move(m)); });
    message_port_register_out("out");
    uwspr_params p = {fs, fl, spb, maxdrift, maxfreqs, halfbandwidth, cf, threshold};
    int rc = uwspr_ctx_create(&p, 0, &d_ctx);
    if (rc != UWSPR_OK) {
      std::string msg = d_ctx ? uwspr_last_error(d_ctx) : uwspr_status_string(rc);
      if (d_ctx) uwspr_ctx_destroy(d_ctx);
      d_ctx = nullptr;
      if (rc == UWSPR_ERR_PARAM || rc == UWSPR_ERR_RANGE || rc == UWSPR_ERR_ARG ||
          rc == UWSPR_ERR_UNSUPPORTED)
        throw std::invalid_argument("uwspr.FDR: " + msg);   // reference: exit(-1), cc:85-90
      throw std::runtime_error("uwspr.FDR: " + msg);
    }
  }
  ~FDR_impl() override {
    if (d_ctx) uwspr_ctx_destroy(d_ctx);
  }
  void set_batch(int n) override { d_batch = n < 1 ? 1 : n; }
  void flush() override { run(); }

 private:
  void transform(message_sptr msg) {
    auto in = std::dynamic_pointer_cast<const samples_pdu>(msg);
    if (!in || (int)in->samples.size() != d_fl) return;  // not a frame PDU
    d_pending.push_back(in);
    if ((int)d_pending.size() >= d_batch) run();
  }
  // Device batch buffers [batch][fl] of (I,Q) pairs, recycled through a pool: a candidates PDU
  // carries a handle to its frame (runtime.h: device_frame) and the buffer returns to the pool when
  // the last PDU of the batch is gone -- sync_and_demodulate reads the frame where FDR left it.
  struct dev_pool {
    std::vector<void *> free_;
    size_t bytes = 0;
    ~dev_pool() { for (void *p : free_) uwspr_device_free(p); }
  };
  std::shared_ptr<void> take_buffer(size_t bytes) {
    if (d_pool->bytes != bytes) {   // batch size changed: start a new pool (old buffers die with their last user)
      d_pool = std::make_shared<dev_pool>();
      d_pool->bytes = bytes;
    }
    void *p = nullptr;
    if (!d_pool->free_.empty()) { p = d_pool->free_.back(); d_pool->free_.pop_back(); }
    else if (uwspr_device_alloc(bytes, &p) != UWSPR_OK) throw std::runtime_error("uwspr.FDR: device memory");
    std::shared_ptr<dev_pool> pool = d_pool;
    return std::shared_ptr<void>(p, [pool, bytes](void *q) {
      if (pool->bytes == bytes) pool->free_.push_back(q); else uwspr_device_free(q);
    });
  }

  void run() {
    const int B = (int)d_pending.size();
    if (B == 0) return;
    const size_t fbytes = (size_t)d_fl * 2 * sizeof(float);
    // Overlap-aware ingest (cc:113-135: consecutive PDUs share fl - hop samples): when the PDUs
    // carry consecutive stream positions only the samples the device has not seen are uploaded and
    // the frames are cut there; otherwise the frames go up whole.  The hop is the distance between
    // consecutive PDUs: within the batch, or -- one PDU per call, the default -- from the PDU before.
    bool streamed = d_pending[0]->stream_pos >= 0;
    const long long hop = B > 1 ? d_pending[1]->stream_pos - d_pending[0]->stream_pos
                                : (d_last_pos >= 0 ? d_pending[0]->stream_pos - d_last_pos : d_hop);
    for (int b = 1; b < B && streamed; b++)
      streamed = d_pending[b]->stream_pos == d_pending[0]->stream_pos + b * hop;
    streamed = streamed && hop > 0 && hop <= d_fl;
    d_last_pos = d_pending[B - 1]->stream_pos;
    std::shared_ptr<void> buf;
    float *dev = nullptr;
    if (streamed) {            // a device batch buffer only when frames are going to be handed on in it
      buf = take_buffer((size_t)B * fbytes);
      dev = static_cast<float *>(buf.get());
    }
    int rc = UWSPR_OK;
    if (streamed) {
      if (hop != d_hop || B > d_stream_frames) {
        rc = uwspr_stream_open(d_ctx, (int)hop, B);
        d_hop = hop; d_stream_frames = B; d_next_pos = -1;
      }
      if (rc == UWSPR_OK && d_next_pos < 0) {          // (re)start at this frame
        rc = uwspr_stream_reset(d_ctx, d_pending[0]->stream_pos);
        d_next_pos = d_pending[0]->stream_pos;
      }
      // d_next_pos: stream index of the first sample the device lacks
      for (int b = 0; b < B && rc == UWSPR_OK; b++) {
        const long long p0 = d_pending[b]->stream_pos, p1 = p0 + d_fl;
        if (p0 > d_next_pos || p1 <= d_next_pos) { streamed = false; break; }   // a gap or a repeat: not a continuation
        const long long skip = d_next_pos - p0;
        rc = uwspr_stream_push(d_ctx, reinterpret_cast<const float *>(d_pending[b]->samples.data() + skip),
                               (int)(p1 - d_next_pos), UWSPR_HOST, nullptr);
        d_next_pos = p1;
      }
      if (rc == UWSPR_OK && streamed) {
        const float *fr = nullptr;
        long long first = 0;
        rc = uwspr_stream_take(d_ctx, B, dev, &fr, &first);
        if (rc == UWSPR_OK && first != d_pending[0]->stream_pos) streamed = false;
      }
      if (!streamed || rc != UWSPR_OK) { d_next_pos = -1; d_hop = 0; }   // fall back below, restart the stream next time
    }
    std::vector<uwspr_candidate> cands((size_t)B * d_maxfreqs);
    std::vector<int32_t> npk(B);
    if (streamed && rc == UWSPR_OK) {
      rc = uwspr_fdr_batch(d_ctx, dev, B, UWSPR_DEVICE_FRAMES, cands.data(), npk.data());
    } else {
      // std::complex<float> is an (I,Q) pair of binary32: the PDU payloads go up as they are
      std::vector<float> frames((size_t)B * d_fl * 2);
      for (int b = 0; b < B; b++)
        memcpy(&frames[(size_t)b * d_fl * 2], d_pending[b]->samples.data(), fbytes);
      rc = uwspr_fdr_batch(d_ctx, frames.data(), B, UWSPR_HOST, cands.data(), npk.data());
      buf.reset();   // no device copy to hand on
    }
    if (rc != UWSPR_OK) throw std::runtime_error(std::string("uwspr.FDR: ") + uwspr_last_error(d_ctx));
    for (int b = 0; b < B; b++) {
      auto out = std::make_shared<candidates_pdu>();
      out->samples = d_pending[b];
      out->npk = npk[b];
      out->candidates.assign(cands.begin() + (size_t)b * d_maxfreqs,
                             cands.begin() + (size_t)b * d_maxfreqs + npk[b]);
      if (buf) { out->dev.ptr = dev + (size_t)b * d_fl * 2; out->dev.keep = buf; }
      message_port_pub("out", out);  // cc:455
    }
    d_pending.clear();
  }

  uwspr_ctx *d_ctx;
  int d_fl, d_maxfreqs, d_batch;
  long long d_hop = 0, d_next_pos = -1, d_last_pos = -1;
  int d_stream_frames = 0;
  std::shared_ptr<dev_pool> d_pool = std::make_shared<dev_pool>();
  std::vector<std::shared_ptr<const samples_pdu> > d_pending;
};

FDR::sptr FDR::make(int fs, int fl, int spb, int maxdrift, int maxfreqs, int halfbandwidth, int cf,
                    int threshold) {
  return sptr(new FDR_impl(fs, fl, spb, maxdrift, maxfreqs, halfbandwidth, cf, threshold));
}

}  // namespace uwspr
}  // namespace gr
