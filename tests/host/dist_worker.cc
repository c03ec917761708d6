// dist_worker.cc -- one rank of tests/test_dist.py's world > 1 test of uwspr_dist_gather (include/uwspr_hip.h).
// A plain C-ABI client without PyTorch in the process: the library's dlopen("librccl.so.1") then finds the test double
// tests/host/fake_rccl.cc first on LD_LIBRARY_PATH (RCCL itself refuses two ranks on one device), and everything of
// uwspr_dist_* runs as it would over RCCL: unique id on rank 0, init on every rank, the root's layout, the self copy,
// the peers' sends, error propagation.
//
//   dist_worker <rank> <world> <root> <rows> <uid file> <mode: ok | exit_early>
//
// Row r of rank p's shard holds bytes f(p, r, k, round); the root checks recv + p * bytes for every p, twice (the
// communicator is reused).  exit_early: every rank but the root leaves after uwspr_dist_init; the root's gather must
// come back with a status code and a message, not hang.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <unistd.h>

#include <vector>

#include "uwspr_hip.h"

static const int SLAB = 32 + 48 * 8;   // uwspr_pack_slabs' record at K = 8

static uint8_t pat(int rank, int row, int k, int round) {
  uint32_t x = (uint32_t)rank * 2654435761u ^ (uint32_t)row * 40503u ^ (uint32_t)k * 2246822519u ^ (uint32_t)round * 3266489917u;
  x ^= x >> 15; x *= 2246822519u; x ^= x >> 13;
  return (uint8_t)x;
}

int main(int argc, char **argv) {
  if (argc < 7) { fprintf(stderr, "usage: dist_worker rank world root rows uidfile mode\n"); return 64; }
  const int rank = atoi(argv[1]), world = atoi(argv[2]), root = atoi(argv[3]), rows = atoi(argv[4]);
  const char *uidfile = argv[5];
  const bool exit_early = !strcmp(argv[6], "exit_early");
  const uwspr_params p = {375, 45000, 256, 0, 200, 10, 1500, 10};
  uwspr_ctx *c = nullptr;
  int rc = uwspr_ctx_create(&p, 0, &c);
  if (rc != UWSPR_OK) { printf("CTX_STATUS %d\n", rc); return 1; }
  char uid[128];
  if (rank == 0) {
    rc = uwspr_dist_unique_id(uid);
    if (rc != UWSPR_OK) { printf("UID_STATUS %d\n", rc); return 1; }
    char tmp[600];
    snprintf(tmp, sizeof(tmp), "%s.tmp", uidfile);
    FILE *f = fopen(tmp, "wb");
    if (!f || fwrite(uid, 1, 128, f) != 128) return 1;
    fclose(f);
    rename(tmp, uidfile);
  } else {
    FILE *f = nullptr;
    for (int t = 0; t < 3000 && !(f = fopen(uidfile, "rb")); t++) usleep(10000);
    if (!f || fread(uid, 1, 128, f) != 128) { printf("UID_MISSING\n"); return 1; }
    fclose(f);
  }
  rc = uwspr_dist_init(c, rank, world, uid);
  if (rc != UWSPR_OK) { printf("INIT_STATUS %d %s\n", rc, uwspr_last_error(c)); return 1; }
  // a second init without finalize is an error of the caller, not a second communicator
  if (uwspr_dist_init(c, rank, world, uid) == UWSPR_OK) { printf("SECOND_INIT_ACCEPTED\n"); return 1; }
  if (exit_early && rank != root) {
    printf("EXIT_EARLY\n");
    fflush(stdout);
    _exit(0);                                   // no finalize, no destroy: the peer is simply gone
  }
  const size_t bytes = (size_t)rows * SLAB;
  void *d_send = nullptr, *d_recv = nullptr;
  if (uwspr_device_alloc(bytes, &d_send) != UWSPR_OK) return 1;
  if (rank == root && uwspr_device_alloc(bytes * world, &d_recv) != UWSPR_OK) return 1;
  std::vector<uint8_t> h(bytes), got(bytes * world);
  for (int round = 0; round < 2; round++) {
    for (int r = 0; r < rows; r++)
      for (int k = 0; k < SLAB; k++) h[(size_t)r * SLAB + k] = pat(rank, r, k, round);
    if (hipMemcpy(d_send, h.data(), bytes, hipMemcpyHostToDevice) != hipSuccess) return 1;
    if (d_recv && hipMemset(d_recv, 0xEE, bytes * world) != hipSuccess) return 1;
    if (hipDeviceSynchronize() != hipSuccess) return 1;
    rc = uwspr_dist_gather(c, d_send, bytes, d_recv, root, UWSPR_DEVICE);
    if (rc != UWSPR_OK) {
      printf("GATHER_STATUS %d %s\n", rc, uwspr_last_error(c));
      fflush(stdout);
      uwspr_ctx_destroy(c);
      return exit_early ? 0 : 1;
    }
    if (uwspr_synchronize(c) != UWSPR_OK) return 1;
    if (rank == root) {
      if (hipMemcpy(got.data(), d_recv, bytes * world, hipMemcpyDeviceToHost) != hipSuccess) return 1;
      for (int q = 0; q < world; q++)
        for (int r = 0; r < rows; r++)
          for (int k = 0; k < SLAB; k++)
            if (got[(size_t)q * bytes + (size_t)r * SLAB + k] != pat(q, r, k, round)) {
              printf("GATHER_WRONG round %d: block %d row %d byte %d\n", round, q, r, k);
              return 1;
            }
    }
  }
  // a gather with root out of range / without a receive buffer on the root: argument errors, nothing is sent
  if (uwspr_dist_gather(c, d_send, bytes, d_recv, world, UWSPR_DEVICE) != UWSPR_ERR_ARG) { printf("BAD_ROOT_ACCEPTED\n"); return 1; }
  if (rank == root && uwspr_dist_gather(c, d_send, bytes, nullptr, root, UWSPR_DEVICE) != UWSPR_ERR_ARG) { printf("NULL_RECV_ACCEPTED\n"); return 1; }
  printf("GATHER_OK rank %d of %d root %d rows %d\n", rank, world, root, rows);
  fflush(stdout);
  uwspr_dist_finalize(c);
  uwspr_device_free(d_send);
  if (d_recv) uwspr_device_free(d_recv);
  uwspr_ctx_destroy(c);
  return 0;
}
