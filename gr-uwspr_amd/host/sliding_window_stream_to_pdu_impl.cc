// sliding_window_stream_to_pdu_impl.cc -- the stream -> PDU framer
// (lib/sliding_window_stream_to_pdu_impl.cc:97-138): a ring buffer of capacity
// C*fl; each work() call pushes its items and, once fl samples are buffered,
// emits ONE fl-sample PDU (the first shift*fs samples are popped, the rest
// peeked) so consecutive PDUs overlap by fl - shift*fs samples.  Pure host code.
// Unlike the reference it does not allocate a fresh fl-element vector on every
// work() call (cc:106), only when a PDU is emitted.
#include <deque>

#include "uwspr/sliding_window_stream_to_pdu.h"

namespace gr {
namespace uwspr {

class sliding_window_stream_to_pdu_impl : public sliding_window_stream_to_pdu {
 public:
  sliding_window_stream_to_pdu_impl(int fs, int fl, int shift, int C)
      : block("sliding_window_stream_to_pdu"), d_fs(fs), d_fl(fl), d_shift(shift),
        d_cap((size_t)C * fl), d_count(0), d_popped(0) {
    message_port_register_out("out");  // cc:54-55
  }
  int work(int noutput_items, const gr_complex *in) override {
    for (int i = 0; i < noutput_items; i++) {
      if (d_buf.size() == d_cap) { d_buf.pop_front(); d_popped++; }  // boost::circular_buffer overwrite
      d_buf.push_back(in[i]);                        // cc:108-110
    }
    d_count += noutput_items;
    if (d_count >= d_fl) {  // cc:113
      const int hop = d_shift * d_fs;
      auto pdu = std::make_shared<samples_pdu>();
      pdu->samples.resize(d_fl);
      pdu->stream_pos = d_popped;      // index of the frame's first sample in the input stream
      for (int i = 0; i < hop; i++) {  // peek + pop, cc:117-124
        pdu->samples[i] = d_buf.front();
        d_buf.pop_front();
      }
      d_popped += hop;
      for (int i = 0; i < d_fl - hop; i++) pdu->samples[hop + i] = d_buf[i];  // cc:126-129
      message_port_pub("out", pdu);  // cc:133
      d_count -= hop;                // cc:134
    }
    return noutput_items;
  }

 private:
  int d_fs, d_fl, d_shift;
  size_t d_cap;
  long d_count;
  long long d_popped;   // samples that have left the front of the ring
  std::deque<gr_complex> d_buf;
};

sliding_window_stream_to_pdu::sptr sliding_window_stream_to_pdu::make(int fs, int fl, int shift,
                                                                      int C) {
  return sptr(new sliding_window_stream_to_pdu_impl(fs, fl, shift, C));
}

}  // namespace uwspr
}  // namespace gr
