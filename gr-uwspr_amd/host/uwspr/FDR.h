/* gr::uwspr::FDR -- same public interface as include/uwspr/FDR.h:37-51 of the
 * reference: an abstract block whose only entry is the static make(). */
#ifndef INCLUDED_UWSPR_FDR_H
#define INCLUDED_UWSPR_FDR_H
#include "runtime.h"
namespace gr {
namespace uwspr {
class UWSPR_API FDR : virtual public block {
 public:
  typedef std::shared_ptr<FDR> sptr;
  /* include/uwspr/FDR.h:49-50.  Throws std::invalid_argument where the
   * reference prints and exit(-1)s (FDR_impl.cc:85-90) and std::runtime_error
   * when no gfx950 device is usable (there is no CPU fallback). */
  static sptr make(int fs, int fl, int spb, int maxdrift, int maxfreqs, int halfbandwidth, int cf,
                   int threshold);
  /* GPU batching knob (not in the reference): collect n PDUs per device call;
   * results are published in arrival order.  Default 1 = one call per PDU. */
  virtual void set_batch(int n) = 0;
  virtual void flush() = 0;
  FDR() : block("FDR") {}
};
}  // namespace uwspr
}  // namespace gr
#endif
