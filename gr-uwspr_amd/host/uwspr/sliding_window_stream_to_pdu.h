/* gr::uwspr::sliding_window_stream_to_pdu --
 * include/uwspr/sliding_window_stream_to_pdu.h:37-53. */
#ifndef INCLUDED_UWSPR_SLIDING_WINDOW_STREAM_TO_PDU_H
#define INCLUDED_UWSPR_SLIDING_WINDOW_STREAM_TO_PDU_H
#include "runtime.h"
namespace gr {
namespace uwspr {
class UWSPR_API sliding_window_stream_to_pdu : virtual public block {
 public:
  typedef std::shared_ptr<sliding_window_stream_to_pdu> sptr;
  /* include/uwspr/sliding_window_stream_to_pdu.h:52 */
  static sptr make(int fs, int fl, int shift, int C);
  /* gr::sync_block::work for the one stream input of gr_complex items
   * (lib/sliding_window_stream_to_pdu_impl.cc:97-138); returns items consumed. */
  virtual int work(int noutput_items, const gr_complex *in) = 0;
  sliding_window_stream_to_pdu() : block("sliding_window_stream_to_pdu") {}
};
}  // namespace uwspr
}  // namespace gr
#endif
