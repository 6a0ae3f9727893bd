// Library-owned collectives of the row-sharded matcher (round 5).
//
// The reference has no collective (its only would-be sites are the dormant torch.distributed calls of
// codebook/models/bottleneck.py:45,75-77); the row-sharded DB is this build's own.  Until round 4
// every exchange of a sharded step was a torch.distributed call issued by the Python host: ~25 us of host time each,
// between two hipGraph segments.  Here the library talks to RCCL itself, on the stream it is handed:
//   * a communicator per (process, device) created from a 128-byte RCCL unique id the host passes around once
//     (qpg_comm_unique_id on rank 0, any side channel - the Python host broadcasts it through torch.distributed's store);
//   * byte-level all-gather / all-to-all of the exchange blocks, a MAX all-reduce of the trouble word, and the
//     ordered-key MIN all-reduce SURVEY.md offered (qpg_allreduce_min_u64: (distance key << 32 | candidate index) packed
//     tables reduce to the global first-wins winner in ONE collective);
//   * everything is stream-ordered and takes no host round trip, so a whole sharded step - kernels AND collectives - is
//     capturable as one hipGraph (RCCL records its kernels into the capturing stream).
// RCCL is bound lazily (dlopen of the librccl the process already holds - torch's - or the system's): the library keeps
// loading, and every other entry point keeps working, on a host without RCCL; only these entry points then fail, loudly.
#include "qpg_common.h"

#include <dlfcn.h>
#include <stdlib.h>

// The handful of RCCL / NCCL ABI names this file uses, declared HERE: the library is bound at run time (dlopen below), so
// the BUILD must not need RCCL's development headers either.  These are NCCL's public, ABI-stable definitions
// (nccl.h / rccl.h: a 128-byte opaque id, an opaque communicator handle, the enum values below); when the header is
// present the static_asserts compare them with it.
#if defined(__has_include)
#if __has_include(<rccl/rccl.h>) && defined(QPG_CHECK_RCCL_ABI)
#include <rccl/rccl.h>
#define QPG_HAVE_RCCL_H 1
#endif
#endif
#ifdef QPG_HAVE_RCCL_H
static_assert(sizeof(ncclUniqueId) == 128 && ncclSuccess == 0 && ncclUint8 == 1 && ncclInt32 == 2 && ncclUint64 == 5 &&
              ncclMax == 2 && ncclMin == 3, "RCCL ABI constants differ from the ones qpg_comm.hip declares");
#else
typedef struct { char internal[128]; } ncclUniqueId;
typedef struct ncclComm* ncclComm_t;
typedef enum { ncclSuccess = 0 } ncclResult_t;
typedef enum { ncclUint8 = 1, ncclInt32 = 2, ncclUint64 = 5 } ncclDataType_t;
typedef enum { ncclMax = 2, ncclMin = 3 } ncclRedOp_t;
#endif

struct qpg_comm {
  ncclComm_t comm;
  int rank, world, device;
};

namespace {
struct Rccl {
  void* h = nullptr;
  ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
  ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
  ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
  ncclResult_t (*AllGather)(const void*, void*, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*AllToAll)(const void*, void*, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*AllReduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
  const char* (*GetErrorString)(ncclResult_t) = nullptr;
  bool ok = false;
};
Rccl g_rccl;

bool rccl_load() {
  if (g_rccl.ok) return true;
  if (g_rccl.h) return false;
  const char* names[] = {getenv("QPG_RCCL_LIB"), "librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
  for (const char* n : names) {
    if (!n || !n[0]) continue;
    // RTLD_NOLOAD first: the copy the process already mapped (torch's) - two RCCL instances in one process would each
    // bring their own bootstrap threads and proxy state
    g_rccl.h = dlopen(n, RTLD_NOW | RTLD_NOLOAD | RTLD_GLOBAL);
    if (g_rccl.h) break;
  }
  for (const char* n : names) {
    if (g_rccl.h) break;
    if (!n || !n[0]) continue;
    g_rccl.h = dlopen(n, RTLD_NOW | RTLD_GLOBAL);
  }
  if (!g_rccl.h) {
    qpg_set_error("qpg_comm: librccl not found (%s)", dlerror());
    return false;
  }
#define QPG_SYM(field, name)                                                         \
  g_rccl.field = reinterpret_cast<decltype(g_rccl.field)>(dlsym(g_rccl.h, name));    \
  if (!g_rccl.field) {                                                               \
    qpg_set_error("qpg_comm: librccl lacks %s", name);                               \
    return false;                                                                    \
  }
  QPG_SYM(GetUniqueId, "ncclGetUniqueId")
  QPG_SYM(CommInitRank, "ncclCommInitRank")
  QPG_SYM(CommDestroy, "ncclCommDestroy")
  QPG_SYM(AllGather, "ncclAllGather")
  QPG_SYM(AllToAll, "ncclAllToAll")
  QPG_SYM(AllReduce, "ncclAllReduce")
  QPG_SYM(GetErrorString, "ncclGetErrorString")
#undef QPG_SYM
  g_rccl.ok = true;
  return true;
}
}  // namespace

#define QPG_RCCL(call, what)                                                                     \
  do {                                                                                           \
    ncclResult_t r_ = (call);                                                                    \
    if (r_ != ncclSuccess) {                                                                     \
      qpg_set_error("%s: RCCL error %d (%s)", what, (int)r_, g_rccl.GetErrorString(r_));         \
      return QPG_EHIP;                                                                           \
    }                                                                                            \
  } while (0)

extern "C" int qpg_comm_unique_id(void* id, int64_t id_bytes) {
  QPG_REQUIRE(id && id_bytes >= (int64_t)sizeof(ncclUniqueId), "qpg_comm_unique_id: needs a %d-byte buffer",
              (int)sizeof(ncclUniqueId));
  if (!rccl_load()) return QPG_EHIP;
  ncclUniqueId u;
  QPG_RCCL(g_rccl.GetUniqueId(&u), "qpg_comm_unique_id");
  memcpy(id, &u, sizeof(u));
  return QPG_OK;
}

extern "C" int qpg_comm_create(qpg_ctx* ctx, const void* id, int64_t id_bytes, int rank, int world, qpg_comm** out) {
  QPG_REQUIRE(ctx && id && out && id_bytes >= (int64_t)sizeof(ncclUniqueId) && world >= 1 && rank >= 0 && rank < world,
              "qpg_comm_create: bad argument (a %d-byte unique id, 0 <= rank < world)", (int)sizeof(ncclUniqueId));
  if (!rccl_load()) return QPG_EHIP;
  ncclUniqueId u;
  memcpy(&u, id, sizeof(u));
  int prev = 0;
  (void)hipGetDevice(&prev);
  if (hipSetDevice(ctx->device) != hipSuccess) {
    qpg_set_error("qpg_comm_create: cannot select device %d", ctx->device);
    return QPG_EHIP;
  }
  ncclComm_t c = nullptr;
  const ncclResult_t r = g_rccl.CommInitRank(&c, world, u, rank);
  (void)hipSetDevice(prev);
  if (r != ncclSuccess) {
    qpg_set_error("qpg_comm_create: ncclCommInitRank failed: %d (%s)", (int)r, g_rccl.GetErrorString(r));
    return QPG_EHIP;
  }
  qpg_comm* q = new qpg_comm;
  q->comm = c;
  q->rank = rank;
  q->world = world;
  q->device = ctx->device;
  *out = q;
  return QPG_OK;
}

extern "C" int qpg_comm_destroy(qpg_comm* c) {
  if (!c) return QPG_OK;
  if (g_rccl.ok && c->comm) (void)g_rccl.CommDestroy(c->comm);
  delete c;
  return QPG_OK;
}

// recv [dev] world x bytes: block w = rank w's `send` (bytes each).  In place allowed (send == recv + rank * bytes).
extern "C" int qpg_comm_allgather(qpg_ctx* ctx, void* stream, qpg_comm* c, const void* send, void* recv, int64_t bytes) {
  QPG_REQUIRE(ctx && c && send && recv && bytes >= 0, "qpg_comm_allgather: bad argument");
  if (bytes == 0) return QPG_OK;
  QPG_RCCL(g_rccl.AllGather(send, recv, (size_t)bytes, ncclUint8, c->comm, qpg_stream(stream)), "qpg_comm_allgather");
  return QPG_OK;
}

// send [dev] world blocks of `bytes` (block w goes to rank w); recv [dev] world blocks (block w came from rank w).
extern "C" int qpg_comm_alltoall(qpg_ctx* ctx, void* stream, qpg_comm* c, const void* send, void* recv, int64_t bytes) {
  QPG_REQUIRE(ctx && c && send && recv && bytes >= 0 && send != recv, "qpg_comm_alltoall: bad argument (out of place)");
  if (bytes == 0) return QPG_OK;
  QPG_RCCL(g_rccl.AllToAll(send, recv, (size_t)bytes, ncclUint8, c->comm, qpg_stream(stream)), "qpg_comm_alltoall");
  return QPG_OK;
}

// in place: buf[i] = max over ranks (the agreed trouble word of an all-to-all step)
extern "C" int qpg_comm_allreduce_max_i32(qpg_ctx* ctx, void* stream, qpg_comm* c, int32_t* buf, int64_t count) {
  QPG_REQUIRE(ctx && c && buf && count >= 0, "qpg_comm_allreduce_max_i32: bad argument");
  if (count == 0) return QPG_OK;
  QPG_RCCL(g_rccl.AllReduce(buf, buf, (size_t)count, ncclInt32, ncclMax, c->comm, qpg_stream(stream)),
           "qpg_comm_allreduce_max_i32");
  return QPG_OK;
}

// in place MIN of packed (order-preserving f32 distance key << 32 | global candidate index) tables: the global per-code
// winner with the reference's first-wins tie rule (lowest index among equal distances, GestureKNN.py:686-689) in one
// collective.  For tables whose values decide every comparison (text f32, Levenshtein); the f64 audio tables go through
// the byte exchanges above.
extern "C" int qpg_allreduce_min_u64(qpg_ctx* ctx, void* stream, qpg_comm* c, uint64_t* buf, int64_t count) {
  QPG_REQUIRE(ctx && c && buf && count >= 0, "qpg_allreduce_min_u64: bad argument");
  if (count == 0) return QPG_OK;
  QPG_RCCL(g_rccl.AllReduce(buf, buf, (size_t)count, ncclUint64, ncclMin, c->comm, qpg_stream(stream)),
           "qpg_allreduce_min_u64");
  return QPG_OK;
}

// ---- packed (key, index) tables for qpg_allreduce_min_u64 ------------------------------------------------------------
__global__ __launch_bounds__(256) void pack_min_u64_kernel(const float* __restrict__ d, const int32_t* __restrict__ idx,
                                                           int64_t n, unsigned long long* __restrict__ out) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const int32_t ix = idx[i];
  const unsigned int b = __float_as_uint(d[i]);
  const unsigned int key = (b >> 31) ? ~b : (b | 0x80000000u);
  out[i] = ix < 0 ? ~0ull : (((unsigned long long)key << 32) | (unsigned int)ix);
}

__global__ __launch_bounds__(256) void unpack_min_u64_kernel(const unsigned long long* __restrict__ in, int64_t n,
                                                             float absent, float* __restrict__ d,
                                                             int32_t* __restrict__ idx) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const unsigned long long v = in[i];
  if (v == ~0ull) {
    d[i] = absent;
    idx[i] = -1;
    return;
  }
  const unsigned int k = (unsigned int)(v >> 32);
  d[i] = __uint_as_float((k >> 31) ? (k & 0x7fffffffu) : ~k);
  idx[i] = (int32_t)(v & 0xffffffffu);
}

// dist f32 [n], idx i32 [n] (-1 = absent) -> packed u64 [n] (absent = all ones: never a minimum against a real entry)
extern "C" int qpg_pack_min_u64(qpg_ctx* ctx, void* stream, const float* dist, const int32_t* idx, int64_t n,
                                uint64_t* packed) {
  QPG_REQUIRE(ctx && dist && idx && packed && n >= 0, "qpg_pack_min_u64: bad argument");
  if (n == 0) return QPG_OK;
  hipLaunchKernelGGL(pack_min_u64_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, qpg_stream(stream), dist, idx, n,
                     reinterpret_cast<unsigned long long*>(packed));
  QPG_LAUNCH_CHECK("pack_min_u64_kernel");
  return QPG_OK;
}

extern "C" int qpg_unpack_min_u64(qpg_ctx* ctx, void* stream, const uint64_t* packed, int64_t n, float absent, float* dist,
                                  int32_t* idx) {
  QPG_REQUIRE(ctx && dist && idx && packed && n >= 0, "qpg_unpack_min_u64: bad argument");
  if (n == 0) return QPG_OK;
  hipLaunchKernelGGL(unpack_min_u64_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, qpg_stream(stream),
                     reinterpret_cast<const unsigned long long*>(packed), n, absent, dist, idx);
  QPG_LAUNCH_CHECK("unpack_min_u64_kernel");
  return QPG_OK;
}
