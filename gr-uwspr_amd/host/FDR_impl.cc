// FDR_impl.cc -- host block mirror of gr::uwspr::FDR on top of the C ABI.
// The handler keeps the reference's contract (lib/FDR_impl.cc:214-456): one
// samples PDU in on port "in", one candidates PDU out on port "out", the input
// sample vector re-used in the output (cc:450).  All arithmetic happens in
// uwspr_fdr_batch on the GPU.
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
  void run() {
    const int B = (int)d_pending.size();
    if (B == 0) return;
    std::vector<float> frames((size_t)B * d_fl * 2);
    for (int b = 0; b < B; b++)
      for (int i = 0; i < d_fl; i++) {
        frames[((size_t)b * d_fl + i) * 2] = d_pending[b]->samples[i].real();
        frames[((size_t)b * d_fl + i) * 2 + 1] = d_pending[b]->samples[i].imag();
      }
    std::vector<uwspr_candidate> cands((size_t)B * d_maxfreqs);
    std::vector<int32_t> npk(B);
    int rc = uwspr_fdr_batch(d_ctx, frames.data(), B, UWSPR_HOST, cands.data(), npk.data());
    if (rc != UWSPR_OK) throw std::runtime_error(std::string("uwspr.FDR: ") + uwspr_last_error(d_ctx));
    for (int b = 0; b < B; b++) {
      auto out = std::make_shared<candidates_pdu>();
      out->samples = d_pending[b];
      out->npk = npk[b];
      out->candidates.assign(cands.begin() + (size_t)b * d_maxfreqs,
                             cands.begin() + (size_t)b * d_maxfreqs + npk[b]);
      message_port_pub("out", out);  // cc:455
    }
    d_pending.clear();
  }

  uwspr_ctx *d_ctx;
  int d_fl, d_maxfreqs, d_batch;
  std::vector<std::shared_ptr<const samples_pdu> > d_pending;
};

FDR::sptr FDR::make(int fs, int fl, int spb, int maxdrift, int maxfreqs, int halfbandwidth, int cf,
                    int threshold) {
  return sptr(new FDR_impl(fs, fl, spb, maxdrift, maxfreqs, halfbandwidth, cf, threshold));
}

}  // namespace uwspr
}  // namespace gr
