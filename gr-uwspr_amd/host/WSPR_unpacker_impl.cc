// WSPR_unpacker_impl.cc -- blob PDU -> "CALL GRID dBm" text PDU
// (lib/WSPR_unpacker_impl.cc:121-139), without the on-disk hash table / log files.
#include "uwspr/WSPR_unpacker.h"

namespace gr {
namespace uwspr {

class WSPR_unpacker_impl : public WSPR_unpacker {
 public:
  WSPR_unpacker_impl() : block("WSPR_unpacker") {
    message_port_register_in("in");
    message_port_register_out("out");
    set_msg_handler("in", [this](message_sptr m) {
      auto b = std::dynamic_pointer_cast<const blob_pdu>(m);
      if (!b) return;
      char txt[32];
      int8_t m7[7];
      for (int i = 0; i < 7; i++) m7[i] = b->bytes[i];
      uwspr_unpack_message(m7, txt, sizeof(txt));
      auto out = std::make_shared<text_pdu>();
      out->text = txt;
      message_port_pub("out", out);
    });
  }
};

WSPR_unpacker::sptr WSPR_unpacker::make() { return sptr(new WSPR_unpacker_impl()); }

}  // namespace uwspr
}  // namespace gr
