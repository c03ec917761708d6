/* gr::uwspr::sync_and_demodulate -- include/uwspr/sync_and_demodulate.h:37-50. */
#ifndef INCLUDED_UWSPR_SYNC_AND_DEMODULATE_H
#define INCLUDED_UWSPR_SYNC_AND_DEMODULATE_H
#include "runtime.h"
namespace gr {
namespace uwspr {
class UWSPR_API sync_and_demodulate : virtual public block {
 public:
  typedef std::shared_ptr<sync_and_demodulate> sptr;
  /* include/uwspr/sync_and_demodulate.h:49 */
  static sptr make(int fs, int fl, int spb, int maxdrift, int maxfreqs, int cf);
  /* side effect of the reference kept behind an option: append decodes to
   * ./messagelog.txt (sync_and_demodulate_impl.cc:98-108,506-526). Default off. */
  virtual void set_messagelog(bool on) = 0;
  virtual unsigned framecount() const = 0;
  /* GPU batching knob (not in the reference): collect n candidates PDUs per device call; blobs are
   * published in arrival order.  Default 1 = one call per PDU. */
  virtual void set_batch(int n) = 0;
  virtual void flush() = 0;
  sync_and_demodulate() : block("sync_and_demodulate") {}
};
}  // namespace uwspr
}  // namespace gr
#endif
