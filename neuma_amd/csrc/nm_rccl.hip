// Library-owned RCCL communicator (round 4): the particle-sharded roll-out issues its collectives itself -
// ncclAllGather once per roll-out, ncclAllReduce once per substep and direction, on the caller's stream, from the C loop in
// nm_rollout.hip - instead of calling back into Python (two ctypes thunks + torch.distributed per collective: ~25 us of host
// time each, 40 per frame).  The reference has no counterpart (single device); this is the build's own design.
//
// librccl is NOT a link-time dependency: it is looked up with dlopen on first use - first the copy the process already holds
// (torch's, RTLD_NOLOAD, so that one RCCL serves both communicators), then librccl.so.1 of the ROCm install - and the six
// entry points are resolved with dlsym.  A box without RCCL loads libneuma_hip.so as before; nm_rccl_* then return
// NM_ERR_INVALID with an explanation.  The nm_comm callback table stays: the gloo tests (and any other transport) use it.
#include <dlfcn.h>
#include <mutex>

#include "nm_common.h"

// the handful of RCCL declarations this file needs (rccl.h: ncclUniqueId = 128 opaque bytes, ncclInt32 = 2, ncclFloat32 = 7,
// ncclSum = 0; every function returns ncclResult_t, 0 = success)
typedef struct { char internal[128]; } nmNcclUniqueId;
typedef void* nmNcclComm;
typedef int (*fnGetUniqueId)(nmNcclUniqueId*);
typedef int (*fnCommInitRank)(nmNcclComm*, int, nmNcclUniqueId, int);
typedef int (*fnCommDestroy)(nmNcclComm);
typedef int (*fnAllReduce)(const void*, void*, size_t, int, int, nmNcclComm, hipStream_t);
typedef int (*fnAllGather)(const void*, void*, size_t, int, nmNcclComm, hipStream_t);
typedef const char* (*fnGetErrorString)(int);
typedef int (*fnSend)(const void*, size_t, int, int, nmNcclComm, hipStream_t);
typedef int (*fnRecv)(void*, size_t, int, int, nmNcclComm, hipStream_t);
typedef int (*fnGroup)(void);

static struct RcclApi {
  void* dl = nullptr;
  fnGetUniqueId get_unique_id = nullptr;
  fnCommInitRank comm_init_rank = nullptr;
  fnCommDestroy comm_destroy = nullptr;
  fnAllReduce all_reduce = nullptr;
  fnAllGather all_gather = nullptr;
  fnGetErrorString error_string = nullptr;
  fnSend send = nullptr;          // (optional: only the neighbour-only exchange needs the four point-to-point entry points)
  fnRecv recv = nullptr;
  fnGroup group_start = nullptr, group_end = nullptr;
  char path[64] = "";
  char err[256] = "-";        // why RCCL could not be bound (recorded once by rccl_load)
} g_rccl;
static std::once_flag g_rccl_once;

static void rccl_load() {
  const char* names[] = {"librccl.so.1", "librccl.so"};
  // (dlerror() clears its state when read: the text of the failure that matters - the last real dlopen, or the first missing
  //  symbol - is recorded HERE, once, and rccl_ready() prints the stored copy)
  snprintf(g_rccl.err, sizeof(g_rccl.err), "-");
  for (int pass = 0; pass < 2 && !g_rccl.dl; ++pass)
    for (const char* nm : names) {
      g_rccl.dl = dlopen(nm, RTLD_NOW | RTLD_GLOBAL | (pass == 0 ? RTLD_NOLOAD : 0));
      if (g_rccl.dl) { snprintf(g_rccl.path, sizeof(g_rccl.path), "%s%s", nm, pass == 0 ? " (already loaded)" : ""); break; }
      const char* e = dlerror();
      if (pass == 1 && e) snprintf(g_rccl.err, sizeof(g_rccl.err), "%s", e);
    }
  if (!g_rccl.dl) return;
  dlerror();      // (clear: the RTLD_NOLOAD misses of the first pass are not errors)
  g_rccl.get_unique_id = (fnGetUniqueId)dlsym(g_rccl.dl, "ncclGetUniqueId");
  g_rccl.comm_init_rank = (fnCommInitRank)dlsym(g_rccl.dl, "ncclCommInitRank");
  g_rccl.comm_destroy = (fnCommDestroy)dlsym(g_rccl.dl, "ncclCommDestroy");
  g_rccl.all_reduce = (fnAllReduce)dlsym(g_rccl.dl, "ncclAllReduce");
  g_rccl.all_gather = (fnAllGather)dlsym(g_rccl.dl, "ncclAllGather");
  g_rccl.error_string = (fnGetErrorString)dlsym(g_rccl.dl, "ncclGetErrorString");
  g_rccl.send = (fnSend)dlsym(g_rccl.dl, "ncclSend");
  g_rccl.recv = (fnRecv)dlsym(g_rccl.dl, "ncclRecv");
  g_rccl.group_start = (fnGroup)dlsym(g_rccl.dl, "ncclGroupStart");
  g_rccl.group_end = (fnGroup)dlsym(g_rccl.dl, "ncclGroupEnd");
  if (!g_rccl.get_unique_id || !g_rccl.comm_init_rank || !g_rccl.comm_destroy || !g_rccl.all_reduce || !g_rccl.all_gather) {
    const char* e = dlerror();
    snprintf(g_rccl.err, sizeof(g_rccl.err), "%s: %s", g_rccl.path, e ? e : "a collective entry point is missing");
  }
}
static int rccl_ready() {
  std::call_once(g_rccl_once, rccl_load);
  if (!g_rccl.dl || !g_rccl.get_unique_id || !g_rccl.comm_init_rank || !g_rccl.comm_destroy || !g_rccl.all_reduce || !g_rccl.all_gather) {
    nm_set_error("RCCL is not available in this process (dlopen librccl.so.1 failed or a symbol is missing): %s", g_rccl.err);
    return NM_ERR_INVALID;
  }
  return NM_OK;
}
#define NM_RCCL_CHECK(call, what)                                                                              \
  do {                                                                                                         \
    const int r_ = (call);                                                                                     \
    if (r_ != 0) {                                                                                             \
      nm_set_error("%s failed: %s (ncclResult %d)", what, g_rccl.error_string ? g_rccl.error_string(r_) : "?", r_); \
      return NM_ERR_HIP;                                                                                       \
    }                                                                                                          \
  } while (0)

struct nm_rccl {
  nmNcclComm comm;
  int world, rank;
};

extern "C" const char* nm_rccl_library(void) { return rccl_ready() == NM_OK ? g_rccl.path : ""; }

extern "C" int nm_rccl_unique_id(void* id128) {
  NM_REQUIRE(id128, "null id buffer (128 bytes)");
  int rc = rccl_ready();
  if (rc) return rc;
  NM_RCCL_CHECK(g_rccl.get_unique_id((nmNcclUniqueId*)id128), "ncclGetUniqueId");
  return NM_OK;
}

// one communicator per rank, on the CURRENT device; collective: every rank of the group calls it with rank 0's id
extern "C" int nm_rccl_create(const void* id128, int32_t world, int32_t rank, nm_rccl** out) {
  NM_REQUIRE(id128 && out, "null pointer");
  NM_REQUIRE(world >= 1 && rank >= 0 && rank < world, "bad world / rank");
  int rc = rccl_ready();
  if (rc) return rc;
  nmNcclUniqueId id;
  memcpy(&id, id128, sizeof(id));
  nm_rccl* c = new nm_rccl();
  c->world = world; c->rank = rank; c->comm = nullptr;
  const int r = g_rccl.comm_init_rank(&c->comm, world, id, rank);
  if (r != 0) {
    nm_set_error("ncclCommInitRank failed: %s (ncclResult %d)", g_rccl.error_string ? g_rccl.error_string(r) : "?", r);
    delete c;
    return NM_ERR_HIP;
  }
  *out = c;
  return NM_OK;
}

extern "C" int nm_rccl_destroy(nm_rccl* c) {
  if (!c) return NM_OK;
  if (c->comm && g_rccl.comm_destroy) g_rccl.comm_destroy(c->comm);
  delete c;
  return NM_OK;
}

extern "C" int nm_rccl_all_reduce_sum_f32(nm_rccl* c, float* buf, int64_t count, void* stream) {
  NM_REQUIRE(c && (buf || count == 0) && count >= 0, "bad arguments");
  if (count == 0) return NM_OK;
  NM_RCCL_CHECK(g_rccl.all_reduce(buf, buf, (size_t)count, 7 /* ncclFloat32 */, 0 /* ncclSum */, c->comm, (hipStream_t)stream), "ncclAllReduce");
  return NM_OK;
}
extern "C" int nm_rccl_all_gather_i32(nm_rccl* c, const int32_t* send, int32_t* recv, int64_t count, void* stream) {
  NM_REQUIRE(c && send && recv && count >= 0, "bad arguments");
  if (count == 0) return NM_OK;
  NM_RCCL_CHECK(g_rccl.all_gather(send, recv, (size_t)count, 2 /* ncclInt32 */, c->comm, (hipStream_t)stream), "ncclAllGather");
  return NM_OK;
}

// neighbour-only exchange: one group call with a send and a receive per peer (the peers' buffers land behind one another)
extern "C" int nm_rccl_exchange_peers_f32(nm_rccl* c, const float* send, float* recv, int64_t count, uint32_t peers, void* stream) {
  NM_REQUIRE(c && count >= 0, "bad arguments");
  // (the caller leaves its own rank out of `peers` - nm_rollout.hip does; a self pair is legal in RCCL, a copy through the
  //  communicator, and is how the one-rank test exercises the four point-to-point entry points)
  if (c->world < 32) peers &= (1u << c->world) - 1u;
  if (count == 0 || peers == 0) return NM_OK;
  NM_REQUIRE(send && recv, "null buffer");
  NM_REQUIRE(g_rccl.send && g_rccl.recv && g_rccl.group_start && g_rccl.group_end, "this RCCL has no ncclSend / ncclRecv / ncclGroupStart / ncclGroupEnd");
  NM_RCCL_CHECK(g_rccl.group_start(), "ncclGroupStart");
  int k = 0, bad = 0;
  for (int q = 0; q < c->world && q < 32; ++q) {
    if (!((peers >> q) & 1u)) continue;
    int r = g_rccl.send(send, (size_t)count, 7 /* ncclFloat32 */, q, c->comm, (hipStream_t)stream);
    if (!r) r = g_rccl.recv(recv + (size_t)k * (size_t)count, (size_t)count, 7, q, c->comm, (hipStream_t)stream);
    if (r && !bad) bad = r;
    ++k;
  }
  const int e = g_rccl.group_end();
  NM_RCCL_CHECK(bad, "ncclSend / ncclRecv");
  NM_RCCL_CHECK(e, "ncclGroupEnd");
  return NM_OK;
}

// the nm_comm of the sharded roll-out, bound to this communicator: the roll-out loop then never leaves the library
static int cb_all_gather(void* user, const int32_t* send, int32_t* recv, int64_t count, void* stream) {
  return nm_rccl_all_gather_i32((nm_rccl*)user, send, recv, count, stream);
}
static int cb_all_reduce(void* user, float* buf, int64_t count, void* stream) {
  return nm_rccl_all_reduce_sum_f32((nm_rccl*)user, buf, count, stream);
}
static int cb_exchange_peers(void* user, const float* send, float* recv, int64_t count, uint32_t peers, void* stream) {
  return nm_rccl_exchange_peers_f32((nm_rccl*)user, send, recv, count, peers, stream);
}
extern "C" int nm_rccl_comm(nm_rccl* c, nm_comm* out) {
  NM_REQUIRE(c && out, "null pointer");
  out->world = c->world;
  out->rank = c->rank;
  out->all_gather_i32 = cb_all_gather;
  out->all_reduce_sum_f32 = cb_all_reduce;
  out->user = c;
  out->exchange_peers_f32 = (g_rccl.send && g_rccl.recv && g_rccl.group_start && g_rccl.group_end) ? cb_exchange_peers : nullptr;
  out->peers = NM_COMM_ALL_RANKS;      // (the caller narrows it: sim/shard.py)
  return NM_OK;
}

// start-up calibration of the shard cost model (sim/shard.py): mean duration, in microseconds, of `reps` back-to-back
// in-place all-reduces of `count` floats on `stream` (HIP events around the batch, after `warm` untimed ones).  Collective:
// every rank calls it with the same arguments.  Synchronises the stream.
extern "C" int nm_rccl_time_all_reduce(nm_rccl* c, float* buf, int64_t count, int32_t warm, int32_t reps, float* us_out, void* stream) {
  NM_REQUIRE(c && buf && us_out && count > 0 && reps > 0 && warm >= 0, "bad arguments");
  hipStream_t s = (hipStream_t)stream;
  for (int i = 0; i < warm; ++i) {
    int rc = nm_rccl_all_reduce_sum_f32(c, buf, count, stream);
    if (rc) return rc;
  }
  hipEvent_t a, b;
  NM_HIP_CHECK(hipEventCreate(&a));
  NM_HIP_CHECK(hipEventCreate(&b));
  NM_HIP_CHECK(hipEventRecord(a, s));
  for (int i = 0; i < reps; ++i) {
    int rc = nm_rccl_all_reduce_sum_f32(c, buf, count, stream);
    if (rc) { hipEventDestroy(a); hipEventDestroy(b); return rc; }
  }
  NM_HIP_CHECK(hipEventRecord(b, s));
  NM_HIP_CHECK(hipEventSynchronize(b));
  float ms = 0.f;
  NM_HIP_CHECK(hipEventElapsedTime(&ms, a, b));
  hipEventDestroy(a); hipEventDestroy(b);
  *us_out = 1e3f * ms / (float)reps;
  return NM_OK;
}

extern "C" int nm_rccl_time_exchange_peers(nm_rccl* c, const float* send, float* recv, int64_t count, uint32_t peers, int32_t warm,
                                           int32_t reps, float* us_out, void* stream) {
  NM_REQUIRE(c && us_out && count > 0 && reps > 0 && warm >= 0, "bad arguments");
  hipStream_t s = (hipStream_t)stream;
  for (int i = 0; i < warm; ++i) {
    int rc = nm_rccl_exchange_peers_f32(c, send, recv, count, peers, stream);
    if (rc) return rc;
  }
  hipEvent_t a, b;
  NM_HIP_CHECK(hipEventCreate(&a));
  NM_HIP_CHECK(hipEventCreate(&b));
  NM_HIP_CHECK(hipEventRecord(a, s));
  for (int i = 0; i < reps; ++i) {
    int rc = nm_rccl_exchange_peers_f32(c, send, recv, count, peers, stream);
    if (rc) { hipEventDestroy(a); hipEventDestroy(b); return rc; }
  }
  NM_HIP_CHECK(hipEventRecord(b, s));
  NM_HIP_CHECK(hipEventSynchronize(b));
  float ms = 0.f;
  NM_HIP_CHECK(hipEventElapsedTime(&ms, a, b));
  hipEventDestroy(a); hipEventDestroy(b);
  *us_out = 1e3f * ms / (float)reps;
  return NM_OK;
}
