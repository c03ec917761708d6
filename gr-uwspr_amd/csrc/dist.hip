// dist.hip -- the one exchange of the multi-GPU path behind the C ABI: the final gather of the
// per-frame candidate slabs to the root rank over RCCL (xGMI within a node).
//
// Frames are independent units (lib/FDR_impl.cc:214 and lib/sync_and_demodulate_impl.cc:315 read only
// their own PDU), so the path shards data-parallel over frames -- global frame b on rank b mod G -- with
// NO data-path collective; each rank runs the whole path on its shard and packs uwspr_pack_slabs.  The
// gather is a group of point-to-point transfers (ncclSend on the peers, ncclRecv x (G-1) on the root):
// every peer -> root transfer rides its own xGMI link, nothing is sent to ranks that do not need it.
//
// librccl is loaded on first use (dlopen): a single-GPU process never touches it.  If the process
// already holds an RCCL (PyTorch brings its own copy) that one is used -- one collective library and
// one HIP runtime per process.
#include <dlfcn.h>
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "uwspr_internal.h"

namespace {

// the few RCCL entry points used (rccl.h: ncclGetUniqueId :187, ncclCommInitRank :220, ncclCommDestroy :260,
// ncclSend :700, ncclRecv :722, ncclGroupStart/End :923); types reduced to what crosses the call
typedef struct { char internal[128]; } nccl_uid;
typedef void *nccl_comm;
struct rccl_api {
  void *lib = nullptr;
  int (*GetUniqueId)(nccl_uid *) = nullptr;
  int (*CommInitRank)(nccl_comm *, int, nccl_uid, int) = nullptr;
  int (*CommDestroy)(nccl_comm) = nullptr;
  int (*Send)(const void *, size_t, int, int, nccl_comm, hipStream_t) = nullptr;
  int (*Recv)(void *, size_t, int, int, nccl_comm, hipStream_t) = nullptr;
  int (*GroupStart)() = nullptr;
  int (*GroupEnd)() = nullptr;
  const char *(*GetErrorString)(int) = nullptr;
  char why[256] = {0};
};
constexpr int kNcclInt8 = 0;   // ncclInt8 / ncclChar

rccl_api api;   // process-wide; api.lib == nullptr: unavailable, api.why says why

rccl_api *rccl() {
  static bool tried = false;
  if (tried) return api.lib ? &api : nullptr;
  tried = true;
  const char *names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
  void *h = nullptr;
  for (const char *n : names) if ((h = dlopen(n, RTLD_NOW | RTLD_NOLOAD))) break;   // the one already in the process
  if (!h) for (const char *n : names) if ((h = dlopen(n, RTLD_NOW | RTLD_LOCAL))) break;
  if (!h) { snprintf(api.why, sizeof(api.why), "librccl not found: %s", dlerror()); return nullptr; }
#define SYM(field, name)                                                        \
  api.field = reinterpret_cast<decltype(api.field)>(dlsym(h, name));           \
  if (!api.field) { snprintf(api.why, sizeof(api.why), "librccl lacks %s", name); return nullptr; }
  SYM(GetUniqueId, "ncclGetUniqueId") SYM(CommInitRank, "ncclCommInitRank") SYM(CommDestroy, "ncclCommDestroy")
  SYM(Send, "ncclSend") SYM(Recv, "ncclRecv") SYM(GroupStart, "ncclGroupStart") SYM(GroupEnd, "ncclGroupEnd")
  SYM(GetErrorString, "ncclGetErrorString")
#undef SYM
  api.lib = h;
  return &api;
}

int dfail(uwspr_ctx *c, int status, const char *fmt, ...) {
  if (c) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(c->err, sizeof(c->err), fmt, ap);
    va_end(ap);
  }
  return status;
}

}  // namespace

extern "C" int uwspr_dist_unique_id(void *id128) {
  if (!id128) return UWSPR_ERR_ARG;
  rccl_api *r = rccl();
  if (!r) return UWSPR_ERR_UNSUPPORTED;
  nccl_uid id;
  if (r->GetUniqueId(&id) != 0) return UWSPR_ERR_HIP;
  memcpy(id128, &id, sizeof(id));
  return UWSPR_OK;
}

extern "C" int uwspr_dist_init(uwspr_ctx *c, int rank, int world, const void *id128) {
  if (!c) return UWSPR_ERR_ARG;
  if (!c->own_stream) return dfail(c, UWSPR_ERR_NODEVICE, "context has no device");
  if (world < 1 || rank < 0 || rank >= world || (world > 1 && !id128)) return dfail(c, UWSPR_ERR_ARG, "rank %d of %d", rank, world);
  if (c->dist_comm) return dfail(c, UWSPR_ERR_ARG, "uwspr_dist_init: already initialised (uwspr_dist_finalize first)");
  c->dist_rank = rank; c->dist_world = world;
  // one rank: the gather is a copy -- no communicator, librccl stays unloaded (option "dist_force_comm"
  // makes a one-rank communicator anyway: a self-test of the RCCL binding on a single-GPU box)
  if (world == 1 && !(c->opt[UWSPR_OPT_DIST_FORCE_COMM] && id128)) return UWSPR_OK;
  rccl_api *r = rccl();
  if (!r) return dfail(c, UWSPR_ERR_UNSUPPORTED, "RCCL unavailable: %s", api.why);
  if (hipSetDevice(c->device) != hipSuccess) return dfail(c, UWSPR_ERR_HIP, "hipSetDevice(%d)", c->device);
  nccl_uid id;
  memcpy(&id, id128, sizeof(id));
  nccl_comm comm = nullptr;
  const int rc = r->CommInitRank(&comm, world, id, rank);
  if (rc != 0) return dfail(c, UWSPR_ERR_HIP, "ncclCommInitRank(rank %d of %d): %s", rank, world, r->GetErrorString(rc));
  c->dist_comm = comm;
  return UWSPR_OK;
}

extern "C" int uwspr_dist_gather(uwspr_ctx *c, const void *send, size_t bytes, void *recv, int root, int where) {
  if (!c || !send || bytes == 0) return UWSPR_ERR_ARG;
  if (c->dist_world < 1) return dfail(c, UWSPR_ERR_ARG, "uwspr_dist_gather before uwspr_dist_init");
  if (root < 0 || root >= c->dist_world) return dfail(c, UWSPR_ERR_ARG, "root %d of %d", root, c->dist_world);
  if (where != UWSPR_DEVICE) return dfail(c, UWSPR_ERR_ARG, "uwspr_dist_gather: device buffers (UWSPR_DEVICE)");
  const bool is_root = c->dist_rank == root;
  if (is_root && !recv) return dfail(c, UWSPR_ERR_ARG, "the root rank needs a receive buffer of world * bytes");
  if (hipSetDevice(c->device) != hipSuccess) return dfail(c, UWSPR_ERR_HIP, "hipSetDevice(%d)", c->device);
  if (is_root) {   // own shard: a copy on the stream
    hipError_t e = hipMemcpyAsync((char *)recv + (size_t)root * bytes, send, bytes, hipMemcpyDeviceToDevice, c->stream);
    if (e != hipSuccess) return dfail(c, UWSPR_ERR_HIP, "hipMemcpyAsync: %s", hipGetErrorString(e));
  }
  if (c->dist_world == 1) return UWSPR_OK;
  rccl_api *r = rccl();
  if (!r || !c->dist_comm) return dfail(c, UWSPR_ERR_ARG, "no communicator");
  int rc = r->GroupStart();
  if (rc == 0) {
    if (is_root) {
      for (int p = 0; p < c->dist_world && rc == 0; p++)
        if (p != root) rc = r->Recv((char *)recv + (size_t)p * bytes, bytes, kNcclInt8, p, c->dist_comm, c->stream);
    } else {
      rc = r->Send(send, bytes, kNcclInt8, root, c->dist_comm, c->stream);
    }
    const int rc2 = r->GroupEnd();
    if (rc == 0) rc = rc2;
  }
  if (rc != 0) return dfail(c, UWSPR_ERR_HIP, "RCCL gather: %s", r->GetErrorString(rc));
  return UWSPR_OK;
}

extern "C" int uwspr_dist_finalize(uwspr_ctx *c) {
  if (!c) return UWSPR_ERR_ARG;
  if (c->dist_comm) {
    rccl_api *r = rccl();
    if (c->own_stream) (void)hipStreamSynchronize(c->stream);
    if (r) (void)r->CommDestroy(c->dist_comm);
    c->dist_comm = nullptr;
  }
  c->dist_world = 0; c->dist_rank = 0;
  return UWSPR_OK;
}
