/* gr::uwspr::WSPR_unpacker -- include/uwspr/WSPR_unpacker.h:37-50 (after the
 * hot path; provided so the receive flowgraph can be wired end to end). */
#ifndef INCLUDED_UWSPR_WSPR_UNPACKER_H
#define INCLUDED_UWSPR_WSPR_UNPACKER_H
#include "runtime.h"
namespace gr {
namespace uwspr {
class UWSPR_API WSPR_unpacker : virtual public block {
 public:
  typedef std::shared_ptr<WSPR_unpacker> sptr;
  static sptr make();
  WSPR_unpacker() : block("WSPR_unpacker") {}
};
}  // namespace uwspr
}  // namespace gr
#endif
