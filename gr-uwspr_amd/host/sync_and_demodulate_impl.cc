// sync_and_demodulate_impl.cc -- host block mirror of
// gr::uwspr::sync_and_demodulate (lib/sync_and_demodulate_impl.cc:315-534):
// candidates PDU in, one 7-byte blob PDU out per decoded candidate, in candidate
// order.  The refinement schedule S0..S5 runs on the GPU for all candidates of
// the frame(s) in one uwspr_demod_batch call (set_batch(n): n candidates PDUs per call; frames
// that FDR left on the device are read there); the gate/retry/Fano loop (cc:457-490)
// is replayed on the host by uwspr_decode_batch (uwspr_decode_candidate per record).
// Like the reference, which stops at the first of its 17 jiggered shifts that decodes, the
// schedule first produces try 0 only (uwspr_set_tries(1)); the records Fano rejects get their
// other 16 tries from uwspr_demod_resume and a second Fano pass.  Same blobs, in the same order.
#include <stdio.h>
#include <string.h>
#include <time.h>

#include <stdexcept>
#include <string>

#include "uwspr/sync_and_demodulate.h"

namespace gr {
namespace uwspr {

class sync_and_demodulate_impl : public sync_and_demodulate {
 public:
  sync_and_demodulate_impl(int fs, int fl, int spb, int maxdrift, int maxfreqs, int cf)
      : block("sync_and_demodulate"), d_ctx(nullptr), d_fl(fl), d_framecount(0), d_log(nullptr) {
    // lib/sync_and_demodulate_impl.cc:69-75
    message_port_register_in("in");
    set_msg_handler("in", [this](message_sptr m) { demodulate(std::move(m)); });
    message_port_register_out("out");
    // fs/fl/spb/maxdrift are stored but unused by the reference's math (cc:77-93);
    // the FDR-only parameters take the flowgraph values.
    uwspr_params p = {fs, fl, spb, maxdrift, maxfreqs, 10, cf, 10};
    int rc = uwspr_ctx_create(&p, 0, &d_ctx);
    if (rc != UWSPR_OK) {
      std::string msg = d_ctx ? uwspr_last_error(d_ctx) : uwspr_status_string(rc);
      if (d_ctx) uwspr_ctx_destroy(d_ctx);
      d_ctx = nullptr;
      throw std::runtime_error("uwspr.sync_and_demodulate: " + msg);
    }
    if (uwspr_set_tries(d_ctx, 1) != UWSPR_OK) {
      std::string msg = uwspr_last_error(d_ctx);
      uwspr_ctx_destroy(d_ctx);
      d_ctx = nullptr;
      throw std::runtime_error("uwspr.sync_and_demodulate: " + msg);
    }
    time(&d_start);
  }
  ~sync_and_demodulate_impl() override {
    if (d_ctx) uwspr_ctx_destroy(d_ctx);
    if (d_log) fclose(d_log);
  }
  void set_messagelog(bool on) override {
    if (on && !d_log) {
      d_log = fopen("messagelog.txt", "a");  // cc:98-108
      if (d_log) { fprintf(d_log, "Start time: %s\n", asctime(localtime(&d_start))); fflush(d_log); }
    } else if (!on && d_log) {
      fclose(d_log);
      d_log = nullptr;
    }
  }
  unsigned framecount() const override { return d_framecount; }
  void set_batch(int n) override { d_batch = n < 1 ? 1 : n; }
  void flush() override { run(); }

 private:
  void demodulate(message_sptr msg) {
    auto in = std::dynamic_pointer_cast<const candidates_pdu>(msg);
    if (!in || !in->samples || (int)in->samples->samples.size() != d_fl) return;
    if (in->npk <= 0) return;
    d_pending.push_back(in);
    if ((int)d_pending.size() >= d_batch) run();
  }
  void run() {
    const int B = (int)d_pending.size();
    if (B == 0) return;
    int per = 0;
    for (auto &p : d_pending) per = p->npk > per ? p->npk : per;
    std::vector<uwspr_candidate> cands((size_t)B * per);
    std::vector<int32_t> npk(B);
    for (int b = 0; b < B; b++) {
      npk[b] = d_pending[b]->npk;
      for (int j = 0; j < npk[b]; j++) cands[(size_t)b * per + j] = d_pending[b]->candidates[j];
    }
    std::vector<uwspr_demod_out> out((size_t)B * per);
    // FDR left the frames of its batch in device memory, one after the other: if these PDUs point
    // at such a run the schedule reads them there (no second upload); else the payloads go up
    bool on_device = d_pending[0]->dev.ptr != nullptr;
    for (int b = 1; b < B && on_device; b++)
      on_device = d_pending[b]->dev.ptr == d_pending[0]->dev.ptr + (size_t)b * d_fl * 2;
    int rc;
    std::vector<float> frames;
    const float *fptr;
    int where;
    if (on_device) {
      fptr = d_pending[0]->dev.ptr; where = UWSPR_DEVICE_FRAMES;
    } else {
      frames.resize((size_t)B * d_fl * 2);
      for (int b = 0; b < B; b++)   // cc:344-345: std::complex<float> is the (I,Q) pair
        memcpy(&frames[(size_t)b * d_fl * 2], d_pending[b]->samples->samples.data(), (size_t)d_fl * 2 * sizeof(float));
      fptr = frames.data(); where = UWSPR_HOST;
    }
    rc = uwspr_demod_batch(d_ctx, fptr, B, where, cands.data(), npk.data(), per, per, out.data());
    if (rc != UWSPR_OK)
      throw std::runtime_error(std::string("uwspr.sync_and_demodulate: ") + uwspr_last_error(d_ctx));
    // cc:389: the candidates are independent, so their Fano runs go to the host
    // cores together; results are published in frame, then candidate order as the reference does
    const int n = B * per;
    std::vector<int8_t> msgs((size_t)n * 7);
    std::vector<uint8_t> got(n);
    if (uwspr_decode_batch(out.data(), n, 0, msgs.data(), nullptr, got.data()) < 0)
      throw std::runtime_error("uwspr.sync_and_demodulate: decode_batch");
    // cc:457-490: try 0 did not decode -> the other jiggered shifts of those candidates
    std::vector<uint8_t> need(n, 0);
    std::vector<int> again;
    for (int i = 0; i < n; i++)
      if (out[i].worth_a_try && !got[i]) { need[i] = 1; again.push_back(i); }
    if (!again.empty()) {
      rc = uwspr_demod_resume(d_ctx, fptr, B, where, need.data(), per, out.data());
      if (rc != UWSPR_OK)
        throw std::runtime_error(std::string("uwspr.sync_and_demodulate: ") + uwspr_last_error(d_ctx));
      std::vector<uwspr_demod_out> sub(again.size());
      for (size_t q = 0; q < again.size(); q++) sub[q] = out[again[q]];
      std::vector<int8_t> m2(again.size() * 7);
      std::vector<uint8_t> g2(again.size());
      if (uwspr_decode_batch(sub.data(), (int)again.size(), 0, m2.data(), nullptr, g2.data()) < 0)
        throw std::runtime_error("uwspr.sync_and_demodulate: decode_batch");
      for (size_t q = 0; q < again.size(); q++) {
        got[again[q]] = g2[q];
        memcpy(&msgs[(size_t)again[q] * 7], &m2[q * 7], 7);
      }
    }
    for (int b = 0; b < B; b++) {
      for (int j = 0; j < npk[b]; j++) {
        if (!got[(size_t)b * per + j]) continue;
        const int8_t *m7 = &msgs[((size_t)b * per + j) * 7];
        d_framecount++;  // cc:492
        if (d_log) {
          const candidate_t &c = d_pending[b]->candidates[j];
          fprintf(d_log, "Frame: %u\nBaseband freq is %2.2f Hz\n(6 Hz) SNR is %2.2f dB\nData: ",
                  d_framecount, c.freq, c.snr);
          for (int i = 0; i < 7; i++) fprintf(d_log, "%02x", (unsigned)(unsigned char)m7[i]);
          fprintf(d_log, "\n\n");
          fflush(d_log);
        }
        auto blob = std::make_shared<blob_pdu>();
        for (int i = 0; i < 7; i++) blob->bytes[i] = m7[i];
        message_port_pub("out", blob);  // cc:528-530
      }
    }
    d_pending.clear();
  }

  uwspr_ctx *d_ctx;
  int d_batch = 1;
  std::vector<std::shared_ptr<const candidates_pdu> > d_pending;
  int d_fl;
  unsigned d_framecount;
  FILE *d_log;
  time_t d_start;
};

sync_and_demodulate::sptr sync_and_demodulate::make(int fs, int fl, int spb, int maxdrift,
                                                    int maxfreqs, int cf) {
  return sptr(new sync_and_demodulate_impl(fs, fl, spb, maxdrift, maxfreqs, cf));
}

}  // namespace uwspr
}  // namespace gr
