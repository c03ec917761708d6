// fake_rccl.cc -- TEST DOUBLE, never part of the product: a librccl.so.1 with the eight entry points
// gr-uwspr_amd/csrc/dist.hip resolves (ncclGetUniqueId, ncclCommInitRank, ncclCommDestroy, ncclSend, ncclRecv,
// ncclGroupStart, ncclGroupEnd, ncclGetErrorString), so that uwspr_dist_gather's own logic -- the root's layout
// recv + p * bytes, the root's self copy, who sends and who receives, root != 0, error propagation -- can run with
// world > 1 on a box with ONE GPU (RCCL itself refuses two ranks on one device).
//
// Transport: one UNIX stream socket per rank, <base>.<rank>, base = the path in the unique id.  The operations of a
// group are queued and executed at ncclGroupEnd: hipStreamSynchronize(stream) (what was enqueued before the gather
// has finished, as it would have on the stream), then sends = D2H copy + connect + write, receives = accept + read +
// H2D copy, matched by the sender's rank.  Synchronous where RCCL is asynchronous; a peer that never shows up is a
// time-out (FAKE_RCCL_TIMEOUT_S, default 20 s) -> ncclSystemError, where the real library would wait (bench.py's
// watchdog is for that).  FAKE_RCCL_LOG=<path>: every operation is appended to <path>.<rank> ("send 0 7072").
//
// Build (tests/test_dist.py does it): hipcc -shared -fPIC -o tests/host/_fake_rccl/librccl.so.1 tests/host/fake_rccl.cc
#include <errno.h>
#include <hip/hip_runtime.h>
#include <poll.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/socket.h>
#include <sys/time.h>
#include <sys/un.h>
#include <time.h>
#include <unistd.h>

#include <string>
#include <vector>

namespace {

struct uid_t128 { char internal[128]; };

struct early_msg { int src; std::vector<char> data; };

struct comm {
  int rank, world;
  std::string base;
  int lfd;
  std::vector<early_msg> early;   // messages of a LATER group that arrived while this rank was still in an earlier one
};

struct op {
  bool send;
  void *buf;
  size_t bytes;
  int peer;
  comm *c;
  hipStream_t stream;
};

thread_local int g_depth = 0;
thread_local std::vector<op> g_ops;

enum { kOk = 0, kSystemError = 2, kInvalidArgument = 4 };

double now() {
  struct timespec ts;
  clock_gettime(CLOCK_MONOTONIC, &ts);
  return ts.tv_sec + 1e-9 * ts.tv_nsec;
}
double timeout_s() {
  const char *e = getenv("FAKE_RCCL_TIMEOUT_S");
  return e ? atof(e) : 20.0;
}
void log_op(const comm *c, const char *what, int peer, size_t bytes) {
  const char *e = getenv("FAKE_RCCL_LOG");
  if (!e) return;
  char path[512];
  snprintf(path, sizeof(path), "%s.%d", e, c->rank);
  FILE *f = fopen(path, "a");
  if (!f) return;
  fprintf(f, "%s %d %zu\n", what, peer, bytes);
  fclose(f);
}
bool rd(int fd, void *p, size_t n, double t_end) {
  char *q = (char *)p;
  while (n) {
    struct pollfd pf = {fd, POLLIN, 0};
    const double left = t_end - now();
    if (left <= 0 || poll(&pf, 1, (int)(left * 1e3) + 1) <= 0) return false;
    const ssize_t k = read(fd, q, n);
    if (k <= 0) return false;
    q += k; n -= (size_t)k;
  }
  return true;
}
bool wr(int fd, const void *p, size_t n) {
  const char *q = (const char *)p;
  while (n) {
    const ssize_t k = write(fd, q, n);
    if (k <= 0) return false;
    q += k; n -= (size_t)k;
  }
  return true;
}
sockaddr_un addr_of(const std::string &base, int rank) {
  sockaddr_un a;
  memset(&a, 0, sizeof(a));
  a.sun_family = AF_UNIX;
  snprintf(a.sun_path, sizeof(a.sun_path), "%s.%d", base.c_str(), rank);
  return a;
}

int do_send(const op &o) {
  std::vector<char> h(o.bytes);
  if (hipMemcpy(h.data(), o.buf, o.bytes, hipMemcpyDeviceToHost) != hipSuccess) return kSystemError;
  const double t_end = now() + timeout_s();
  const sockaddr_un a = addr_of(o.c->base, o.peer);
  int fd = -1;
  for (;;) {                                         // the peer's socket appears when it has initialised
    fd = socket(AF_UNIX, SOCK_STREAM, 0);
    if (fd < 0) return kSystemError;
    if (connect(fd, (const sockaddr *)&a, sizeof(a)) == 0) break;
    close(fd);
    if (now() > t_end) return kSystemError;
    usleep(20000);
  }
  const uint64_t hdr[2] = {(uint64_t)o.c->rank, (uint64_t)o.bytes};
  const bool ok = wr(fd, hdr, sizeof(hdr)) && wr(fd, h.data(), o.bytes);
  close(fd);
  return ok ? kOk : kSystemError;
}

int do_recvs(comm *c, std::vector<op *> &rs) {
  const double t_end = now() + timeout_s();
  size_t done = 0;
  // a peer may already be one gather ahead (it sends, returns, sends again while the root is still receiving the first
  // round from a slower peer): transfers between two ranks are ordered, so such a message waits for its own group
  for (op *r : rs)
    for (size_t e = 0; e < c->early.size(); e++)
      if (c->early[e].src == r->peer && c->early[e].data.size() == r->bytes) {
        if (hipMemcpy(r->buf, c->early[e].data.data(), r->bytes, hipMemcpyHostToDevice) != hipSuccess) return kSystemError;
        c->early.erase(c->early.begin() + (long)e);
        r->buf = nullptr;
        done++;
        break;
      }
  while (done < rs.size()) {
    struct pollfd pf = {c->lfd, POLLIN, 0};
    const double left = t_end - now();
    if (left <= 0 || poll(&pf, 1, (int)(left * 1e3) + 1) <= 0) return kSystemError;      // a peer never arrived
    const int fd = accept(c->lfd, nullptr, nullptr);
    if (fd < 0) return kSystemError;
    uint64_t hdr[2];
    if (!rd(fd, hdr, sizeof(hdr), t_end)) { close(fd); return kSystemError; }
    if (hdr[0] >= (uint64_t)c->world || hdr[1] > (1ull << 32)) { close(fd); return kInvalidArgument; }
    op *m = nullptr;
    for (op *r : rs)
      if (r->buf && r->peer == (int)hdr[0] && r->bytes == hdr[1]) { m = r; break; }
    std::vector<char> h(hdr[1]);
    const bool ok = rd(fd, h.data(), h.size(), t_end);
    close(fd);
    if (!ok) return kSystemError;
    if (!m) {                                       // not of this group: the sender is a gather ahead
      c->early.push_back(early_msg{(int)hdr[0], std::move(h)});
      continue;
    }
    if (hipMemcpy(m->buf, h.data(), m->bytes, hipMemcpyHostToDevice) != hipSuccess) return kSystemError;
    m->buf = nullptr;                                                                      // matched
    done++;
  }
  return kOk;
}

int run_group() {
  std::vector<op> ops;
  ops.swap(g_ops);
  int rc = kOk;
  for (const op &o : ops)
    if (hipStreamSynchronize(o.stream) != hipSuccess) rc = kSystemError;
  for (const op &o : ops)
    if (rc == kOk && o.send) rc = do_send(o);
  std::vector<op *> rs;
  comm *c = nullptr;
  for (op &o : ops)
    if (!o.send) { rs.push_back(&o); c = o.c; }
  if (rc == kOk && !rs.empty()) rc = do_recvs(c, rs);
  return rc;
}

}  // namespace

extern "C" {

int ncclGetUniqueId(uid_t128 *id) {
  if (!id) return kInvalidArgument;
  memset(id, 0, sizeof(*id));
  const char *dir = getenv("FAKE_RCCL_DIR");
  struct timeval tv;
  gettimeofday(&tv, nullptr);
  snprintf(id->internal, sizeof(id->internal), "%s/fake_rccl_%d_%ld%06ld", dir ? dir : "/tmp", (int)getpid(), (long)tv.tv_sec,
           (long)tv.tv_usec);
  return kOk;
}

int ncclCommInitRank(void **out, int nranks, uid_t128 id, int rank) {
  if (!out || nranks < 1 || rank < 0 || rank >= nranks || !id.internal[0]) return kInvalidArgument;
  id.internal[127] = 0;
  comm *c = new comm{rank, nranks, std::string(id.internal), -1, {}};
  const sockaddr_un a = addr_of(c->base, rank);
  unlink(a.sun_path);
  c->lfd = socket(AF_UNIX, SOCK_STREAM, 0);
  if (c->lfd < 0 || bind(c->lfd, (const sockaddr *)&a, sizeof(a)) != 0 || listen(c->lfd, 64) != 0) {
    if (c->lfd >= 0) close(c->lfd);
    delete c;
    return kSystemError;
  }
  *out = c;
  return kOk;
}

int ncclCommDestroy(void *p) {
  comm *c = (comm *)p;
  if (!c) return kInvalidArgument;
  close(c->lfd);
  const sockaddr_un a = addr_of(c->base, c->rank);
  unlink(a.sun_path);
  delete c;
  return kOk;
}

int ncclGroupStart() { g_depth++; return kOk; }

int ncclGroupEnd() {
  if (g_depth <= 0) return kInvalidArgument;
  if (--g_depth > 0) return kOk;
  return run_group();
}

int ncclSend(const void *buf, size_t count, int dtype, int peer, void *cm, hipStream_t stream) {
  comm *c = (comm *)cm;
  if (!c || !buf || dtype != 0 || peer < 0 || peer >= c->world || peer == c->rank) return kInvalidArgument;
  log_op(c, "send", peer, count);
  g_ops.push_back(op{true, (void *)buf, count, peer, c, stream});
  return g_depth > 0 ? kOk : run_group();
}

int ncclRecv(void *buf, size_t count, int dtype, int peer, void *cm, hipStream_t stream) {
  comm *c = (comm *)cm;
  if (!c || !buf || dtype != 0 || peer < 0 || peer >= c->world || peer == c->rank) return kInvalidArgument;
  log_op(c, "recv", peer, count);
  g_ops.push_back(op{false, buf, count, peer, c, stream});
  return g_depth > 0 ? kOk : run_group();
}

const char *ncclGetErrorString(int rc) {
  switch (rc) {
    case kOk: return "no error";
    case kSystemError: return "unhandled system error (fake_rccl: a peer did not arrive, or a socket / copy failed)";
    case kInvalidArgument: return "invalid argument (fake_rccl)";
    default: return "unknown (fake_rccl)";
  }
}

}  // extern "C"
