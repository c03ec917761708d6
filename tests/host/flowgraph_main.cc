// flowgraph_main.cc -- drives the host block mirror the way the reference's
// receive flowgraph does (examples/WaveFilePlusNoiseDecode.grc, the part after
// the resampler):  stream -> sliding_window_stream_to_pdu(375,45000,9,2)
//   -> FDR(375,45000,256,0,200,10,1500,10) -> sync_and_demodulate(375,45000,256,0,200,1500)
//   -> WSPR_unpacker() -> sink
// modes:  framer | errors | decode <file.c2> | stream <file.c2> | weak <file.c2> <sigma>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <stdexcept>
#include <vector>

#include "uwspr/FDR.h"
#include "uwspr/WSPR_unpacker.h"
#include "uwspr/sliding_window_stream_to_pdu.h"
#include "uwspr/sync_and_demodulate.h"

using namespace gr::uwspr;

static int framer() {
  auto sw = sliding_window_stream_to_pdu::make(375, 45000, 9, 2);
  message_sink sink;
  block::msg_connect(sw.get(), "out", &sink, "in");
  std::vector<gr_complex> ramp(4096);
  long n = 0;
  for (int call = 0; call < 30; call++) {  // 122880 samples
    for (auto &v : ramp) { v = gr_complex((float)n, -(float)n); n++; }
    if (sw->work((int)ramp.size(), ramp.data()) != (int)ramp.size()) return 2;
  }
  printf("pdus %zu\n", sink.received.size());
  for (auto &m : sink.received) {
    auto p = std::dynamic_pointer_cast<const samples_pdu>(m);
    printf("first %.0f last %.0f size %zu\n", p->samples.front().real(), p->samples.back().real(),
           p->samples.size());
  }
  return 0;
}

static int errors() {
  int seen = 0;
  try { FDR::make(375, 45000, 256, 0, 200, 200, 1500, 10); } catch (const std::invalid_argument &e) {
    printf("invalid_argument: %s\n", e.what()); seen |= 1; }
  try { FDR::make(375, 45000, 256, 0, 200, 187, 1500, 10); } catch (const std::invalid_argument &e) {
    printf("invalid_argument: %s\n", e.what()); seen |= 2; }
  try { auto f = FDR::make(375, 45000, 256, 0, 200, 10, 1500, 10); printf("FDR created on a GPU\n"); seen |= 4; }
  catch (const std::runtime_error &e) { printf("runtime_error: %s\n", e.what()); seen |= 8; }
  printf("seen %d\n", seen);
  return ((seen & 3) == 3 && (seen & 12)) ? 0 : 1;
}

static int decode(const char *path) {
  std::vector<float> iq(2 * 45000);
  if (uwspr_c2_read(path, iq.data(), nullptr, nullptr) != 0) { printf("cannot read %s\n", path); return 2; }
  auto sw = sliding_window_stream_to_pdu::make(375, 45000, 9, 2);
  auto fdr = FDR::make(375, 45000, 256, 0, 200, 10, 1500, 10);
  auto sad = sync_and_demodulate::make(375, 45000, 256, 0, 200, 1500);
  auto unp = WSPR_unpacker::make();
  message_sink cands, texts;
  block::msg_connect(sw.get(), "out", fdr.get(), "in");
  block::msg_connect(fdr.get(), "out", sad.get(), "in");
  block::msg_connect(fdr.get(), "out", &cands, "in");
  block::msg_connect(sad.get(), "out", unp.get(), "in");
  block::msg_connect(unp.get(), "out", &texts, "in");
  std::vector<gr_complex> s(45000);
  for (int i = 0; i < 45000; i++) s[i] = gr_complex(iq[2 * i], iq[2 * i + 1]);
  for (int off = 0; off < 45000; off += 4500) sw->work(4500, s.data() + off);
  printf("ports %d%d%d%d\n", fdr->has_in("in"), fdr->has_out("out"), sad->has_in("in"), sad->has_out("out"));
  for (auto &m : cands.received) {
    auto c = std::dynamic_pointer_cast<const candidates_pdu>(m);
    printf("npk %d\n", c->npk);
    for (auto &k : c->candidates)
      printf("cand type %d freq %.9f sync %.9f shift %d V1 %.1f V2 %.1f p1 %d p2 %d\n", k.m_type, k.freq,
             k.sync, k.shift, k.m_nonlinear.V1, k.m_nonlinear.V2, k.m_nonlinear.p1, k.m_nonlinear.p2);
  }
  for (auto &m : texts.received) printf("text %s\n", std::dynamic_pointer_cast<const text_pdu>(m)->text.c_str());
  printf("frames %u\n", sad->framecount());
  return 0;
}

// A continuous stream (the .c2 frame twice, 5 hops apart, on a noise floor) through the chain twice:
// batched (FDR / sync 4 PDUs per call: stream ingest, frames handed over on the device) and one
// PDU per call (whole-frame uploads).  Prints both result lists; they must be identical.
static int stream(const char *path) {
  std::vector<float> iq(2 * 45000);
  if (uwspr_c2_read(path, iq.data(), nullptr, nullptr) != 0) { printf("cannot read %s\n", path); return 2; }
  const int hop = 3375, nfr = 8, total = 45000 + (nfr - 1) * hop;
  std::vector<gr_complex> s(total);
  unsigned lcg = 12345u;
  for (int i = 0; i < total; i++) {
    lcg = lcg * 1664525u + 1013904223u; const float a = ((lcg >> 8) & 0xffff) / 65536.0f - 0.5f;
    lcg = lcg * 1664525u + 1013904223u; const float b = ((lcg >> 8) & 0xffff) / 65536.0f - 0.5f;
    s[i] = gr_complex(0.05f * a, 0.05f * b);
  }
  for (int i = 0; i < 45000; i++) s[i] += gr_complex(iq[2 * i], iq[2 * i + 1]);
  for (int i = 0; i + 5 * hop < total && i < 45000; i++) s[i + 5 * hop] += 0.7f * gr_complex(iq[2 * i], iq[2 * i + 1]);
  for (int pass = 0; pass < 2; pass++) {
    auto sw = sliding_window_stream_to_pdu::make(375, 45000, 9, 2);
    auto fdr = FDR::make(375, 45000, 256, 0, 200, 10, 1500, 10);
    auto sad = sync_and_demodulate::make(375, 45000, 256, 0, 200, 1500);
    auto unp = WSPR_unpacker::make();
    message_sink cands, texts;
    block::msg_connect(sw.get(), "out", fdr.get(), "in");
    block::msg_connect(fdr.get(), "out", sad.get(), "in");
    block::msg_connect(fdr.get(), "out", &cands, "in");
    block::msg_connect(sad.get(), "out", unp.get(), "in");
    block::msg_connect(unp.get(), "out", &texts, "in");
    if (pass == 0) { fdr->set_batch(4); sad->set_batch(4); }
    // the framer emits at most one PDU per work() call (cc:113): feed it hop-sized pieces
    for (int off = 0; off < total; off += 1125) sw->work(off + 1125 <= total ? 1125 : total - off, s.data() + off);
    fdr->flush();
    sad->flush();
    int ondev = 0;
    printf("pass %d pdus %zu\n", pass, cands.received.size());
    for (auto &m : cands.received) {
      auto c = std::dynamic_pointer_cast<const candidates_pdu>(m);
      ondev += c->dev.ptr != nullptr;
      printf("p%d pos %lld npk %d", pass, c->samples->stream_pos, c->npk);
      for (auto &k : c->candidates) printf(" | %d %.9f %.9f %d", k.m_type, k.freq, k.sync, k.shift);
      printf("\n");
    }
    for (auto &m : texts.received) printf("p%d text %s\n", pass, std::dynamic_pointer_cast<const text_pdu>(m)->text.c_str());
    printf("pass %d frames %u on_device %d\n", pass, sad->framecount(), ondev);
  }
  return 0;
}

// The .c2 frame under heavy noise, eight independent copies: through the block mirror (which
// produces try 0 first and resumes the records Fano rejects) and through the C ABI with all 17
// tries at once.  Prints the decoded blobs of both ways and the try each eager decode used.
static int weak(const char *path, double sigma) {
  std::vector<float> iq(2 * 45000);
  if (uwspr_c2_read(path, iq.data(), nullptr, nullptr) != 0) { printf("cannot read %s\n", path); return 2; }
  const int NB = 8;
  std::vector<float> frames((size_t)NB * 2 * 45000);
  unsigned long long st = 88172645463325252ull;
  auto u01 = [&]() { st ^= st << 13; st ^= st >> 7; st ^= st << 17; return (double)(st >> 11) / 9007199254740992.0; };
  for (size_t i = 0; i < frames.size(); i += 2) {       // Box-Muller, two normals per complex sample
    const double r = sqrt(-2.0 * log(u01() + 1e-300)), a = 6.283185307179586 * u01();
    frames[i] = iq[i % (2 * 45000)] + (float)(sigma * r * cos(a));
    frames[i + 1] = iq[(i + 1) % (2 * 45000)] + (float)(sigma * r * sin(a));
  }
  // (a) the mirror: FDR -> sync_and_demodulate, all eight PDUs in one device call
  auto fdr = FDR::make(375, 45000, 256, 0, 200, 10, 1500, 10);
  auto sad = sync_and_demodulate::make(375, 45000, 256, 0, 200, 1500);
  message_sink blobs;
  block::msg_connect(fdr.get(), "out", sad.get(), "in");
  block::msg_connect(sad.get(), "out", &blobs, "in");
  fdr->set_batch(NB); sad->set_batch(NB);
  for (int b = 0; b < NB; b++) {
    auto pdu = std::make_shared<samples_pdu>();
    pdu->samples.resize(45000);
    for (int i = 0; i < 45000; i++)
      pdu->samples[i] = gr_complex(frames[((size_t)b * 45000 + i) * 2], frames[((size_t)b * 45000 + i) * 2 + 1]);
    fdr->post("in", pdu);
  }
  fdr->flush(); sad->flush();
  for (auto &m : blobs.received) {
    auto bl = std::dynamic_pointer_cast<const blob_pdu>(m);
    printf("mirror");
    for (int i = 0; i < 7; i++) printf(" %02x", (unsigned)(unsigned char)bl->bytes[i]);
    printf("\n");
  }
  // (b) the C ABI, every try produced before the host sees any
  uwspr_params p = {375, 45000, 256, 0, 200, 10, 1500, 10};
  uwspr_ctx *c = nullptr;
  if (uwspr_ctx_create(&p, 0, &c) != UWSPR_OK) { printf("ctx\n"); return 3; }
  const int per = 26;
  std::vector<uwspr_candidate> cands((size_t)NB * 200);
  std::vector<int32_t> npk(NB);
  std::vector<uwspr_demod_out> out((size_t)NB * per);
  if (uwspr_pipeline_batch(c, frames.data(), NB, UWSPR_HOST, per, cands.data(), npk.data(), out.data()) != UWSPR_OK) {
    printf("pipeline: %s\n", uwspr_last_error(c)); return 4;
  }
  std::vector<int8_t> msgs((size_t)NB * per * 7);
  std::vector<int32_t> idt((size_t)NB * per);
  std::vector<uint8_t> got((size_t)NB * per);
  uwspr_decode_batch(out.data(), NB * per, 0, msgs.data(), idt.data(), got.data());
  for (int b = 0; b < NB; b++)
    for (int j = 0; j < npk[b] && j < per; j++) {
      const size_t q = (size_t)b * per + j;
      if (!got[q]) continue;
      printf("eager");
      for (int i = 0; i < 7; i++) printf(" %02x", (unsigned)(unsigned char)msgs[q * 7 + i]);
      printf("\ntry %d\n", idt[q]);
    }
  uwspr_ctx_destroy(c);
  return 0;
}

int main(int argc, char **argv) {
  if (argc >= 4 && !strcmp(argv[1], "weak")) return weak(argv[2], atof(argv[3]));
  if (argc >= 3 && !strcmp(argv[1], "stream")) return stream(argv[2]);
  if (argc >= 2 && !strcmp(argv[1], "framer")) return framer();
  if (argc >= 2 && !strcmp(argv[1], "errors")) return errors();
  if (argc >= 3 && !strcmp(argv[1], "decode")) return decode(argv[2]);
  fprintf(stderr, "usage: %s framer | errors | decode file.c2\n", argv[0]);
  return 64;
}
