/* runtime.h -- the few pieces of the GNU Radio runtime the three blocks of the
 * hot path touch (message ports, handlers, publish), so that the block classes
 * below keep the reference's shape when GNU Radio itself is absent (it is, in
 * this image).  With a real GNU Radio these map one-to-one onto gr::block /
 * pmt: message_port_register_in/out, set_msg_handler, message_port_pub,
 * msg_connect (lib/FDR_impl.cc:55-61, lib/sync_and_demodulate_impl.cc:69-75,
 * lib/sliding_window_stream_to_pdu_impl.cc:54-55).
 *
 * PDUs are typed structs instead of pmt trees; the field lists are the
 * reference's PDU schemas:
 *   samples_pdu     cons(NIL, vector<complex> [fl])                sliding_window cc:130
 *   candidates_pdu  cons(NIL, tuple(samples, npk, vector<tuple>))  FDR_impl.cc:414-455
 *   blob_pdu        cons(NIL, blob(7 bytes))                       sync_and_demodulate_impl.cc:528-530
 *   text_pdu        cons(NIL, blob(text))                          WSPR_unpacker_impl.cc:133-135
 */
#ifndef INCLUDED_UWSPR_RUNTIME_H
#define INCLUDED_UWSPR_RUNTIME_H

#include <complex>
#include <functional>
#include <map>
#include <memory>
#include <string>
#include <utility>
#include <vector>

#include "../../../include/uwspr_hip.h"
#include "api.h"

typedef std::complex<float> gr_complex;

namespace gr {
namespace uwspr {

/* candidate_t, lib/candidate_t.h:27-50, is layout-identical to uwspr_candidate */
typedef uwspr_candidate candidate_t;
enum Modes { linear = UWSPR_LINEAR, nonlinear = UWSPR_NONLINEAR };

struct message {
  virtual ~message() {}
};
typedef std::shared_ptr<const message> message_sptr;

struct samples_pdu : message {
  std::vector<gr_complex> samples;  /* fl samples; I = real, Q = imag */
  /* PDU metadata (the dictionary in the car of a real PDU, NIL in the reference): index of samples[0]
   * in the framer's input stream, -1 = unknown.  Lets FDR upload only the samples it has not seen. */
  long long stream_pos = -1;
};
/* A frame that already lives in device memory: [fl] (I,Q) binary32 pairs.  `keep` owns the memory
 * (returned to the producer's pool when the last reference goes), so the handle stays valid however
 * late the consumer runs -- what a real GNU Radio message queue needs. */
struct device_frame {
  const float *ptr = nullptr;
  std::shared_ptr<void> keep;
};
struct candidates_pdu : message {
  std::shared_ptr<const samples_pdu> samples;  /* the input vector is re-used, FDR_impl.cc:450 */
  int npk;
  std::vector<candidate_t> candidates;
  device_frame dev;   /* metadata: the same frame on the device (FDR has it there anyway); ptr == nullptr: none */
};
struct blob_pdu : message {
  signed char bytes[7];
};
struct text_pdu : message {
  std::string text;
};

class UWSPR_API block {
 public:
  typedef std::function<void(message_sptr)> handler_t;
  explicit block(const std::string &name) : d_name(name) {}
  virtual ~block() {}
  const std::string &name() const { return d_name; }
  void message_port_register_in(const std::string &port) { d_in[port]; }
  void message_port_register_out(const std::string &port) { d_out[port]; }
  void set_msg_handler(const std::string &port, handler_t h) { d_in[port] = std::move(h); }
  /* deliver a message to an input port (what the scheduler does) */
  void post(const std::string &port, message_sptr msg) {
    auto it = d_in.find(port);
    if (it != d_in.end() && it->second) it->second(std::move(msg));
  }
  void message_port_pub(const std::string &port, message_sptr msg) {
    for (auto &sub : d_out[port]) sub.first->post(sub.second, msg);
  }
  bool has_in(const std::string &p) const { return d_in.count(p) != 0; }
  bool has_out(const std::string &p) const { return d_out.count(p) != 0; }
  /* top_block::msg_connect(src, "out", dst, "in") */
  static void msg_connect(block *src, const std::string &out, block *dst, const std::string &in) {
    src->d_out[out].push_back(std::make_pair(dst, in));
  }

 private:
  std::string d_name;
  std::map<std::string, handler_t> d_in;
  std::map<std::string, std::vector<std::pair<block *, std::string> > > d_out;
};

/* terminal block collecting whatever reaches it (blocks.message_debug) */
class UWSPR_API message_sink : public block {
 public:
  message_sink() : block("message_sink") {
    message_port_register_in("in");
    set_msg_handler("in", [this](message_sptr m) { received.push_back(m); });
  }
  std::vector<message_sptr> received;
};

}  // namespace uwspr
}  // namespace gr
#endif
