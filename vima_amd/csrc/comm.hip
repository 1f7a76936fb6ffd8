// Data-parallel exchange step of the policy path behind the C ABI (include/vima_hip.h): ONE all-gather of the
// [rows, 700] fp32 action logits per step over RCCL / xGMI (the reference has no distributed code).
// RCCL is bound lazily with dlopen so that loading libvima_hip.so never depends on it and a process that already has
// torch's copy loaded shares that copy (one RCCL per process).
#include "../../include/vima_hip.h"

#include <dlfcn.h>
#include <hip/hip_runtime.h>
#include <string.h>

#include <mutex>
#include <string>

namespace vima {
int api_fail(const std::string& m);   // vima_api.hip: sets the thread's vima_last_error() text, returns 1
}

namespace {

constexpr int kIdBytes = 128;   // NCCL_UNIQUE_ID_BYTES (rccl.h)
struct UniqueId { char internal[kIdBytes]; };
typedef void* comm_t;
enum { kFloat32 = 7 };          // ncclFloat32 (rccl.h ncclDataType_t)

struct Rccl {
  void* lib = nullptr;
  int (*GetUniqueId)(UniqueId*) = nullptr;
  int (*CommInitRank)(comm_t*, int, UniqueId, int) = nullptr;
  int (*AllGather)(const void*, void*, size_t, int, comm_t, hipStream_t) = nullptr;
  int (*CommDestroy)(comm_t) = nullptr;
  const char* (*GetErrorString)(int) = nullptr;
  std::string err;
};

Rccl g_rccl;
std::once_flag g_once;

void bind_rccl() {
  const char* names[] = {"librccl.so", "librccl.so.1"};
  for (const char* n : names)                       // a copy that is already in the process (torch's) wins
    if (!g_rccl.lib) g_rccl.lib = dlopen(n, RTLD_NOW | RTLD_NOLOAD);
  const char* paths[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
  for (const char* n : paths)
    if (!g_rccl.lib) g_rccl.lib = dlopen(n, RTLD_NOW | RTLD_GLOBAL);
  if (!g_rccl.lib) {
    const char* e = dlerror();   // ONE call: dlerror() clears the pending error, a second call returns NULL
    g_rccl.err = std::string("cannot load RCCL: ") + (e ? e : "?");
    return;
  }
#define BIND(field, sym)                                                         \
  *(void**)(&g_rccl.field) = dlsym(g_rccl.lib, sym);                             \
  if (!g_rccl.field) { g_rccl.err = std::string("RCCL symbol missing: ") + sym; return; }
  BIND(GetUniqueId, "ncclGetUniqueId")
  BIND(CommInitRank, "ncclCommInitRank")
  BIND(AllGather, "ncclAllGather")
  BIND(CommDestroy, "ncclCommDestroy")
  BIND(GetErrorString, "ncclGetErrorString")
#undef BIND
}

int cfail(const std::string& m) { return vima::api_fail(m); }   // text retrievable with vima_last_error()

int need_rccl() {
  std::call_once(g_once, bind_rccl);
  return g_rccl.err.empty() ? 0 : cfail(g_rccl.err);
}

#define NCK(expr)                                                                          \
  do {                                                                                     \
    int r__ = (expr);                                                                      \
    if (r__ != 0) return cfail(std::string(#expr) + ": " + g_rccl.GetErrorString(r__));    \
  } while (0)

}  // namespace

struct VimaComm {
  comm_t comm = nullptr;
  int world = 1, rank = 0, device = 0;
};

extern "C" {

int vima_comm_unique_id(uint8_t id[VIMA_COMM_ID_BYTES]) {
  if (!id) return cfail("vima_comm_unique_id: null id");
  if (need_rccl()) return 1;
  UniqueId u;
  NCK(g_rccl.GetUniqueId(&u));
  memcpy(id, u.internal, kIdBytes);
  return 0;
}

int vima_comm_create(const uint8_t id[VIMA_COMM_ID_BYTES], int world, int rank, int device, VimaComm** out) {
  if (!id || !out) return cfail("vima_comm_create: null argument");
  if (world < 1 || rank < 0 || rank >= world) return cfail("vima_comm_create: rank must be in [0, world)");
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0) return cfail("vima_comm_create: no HIP device available");
  if (device < 0 || device >= ndev) return cfail("vima_comm_create: device index out of range");
  if (need_rccl()) return 1;
  int prev_dev = -1;                                   // the caller's current device is restored on every exit path
  (void)hipGetDevice(&prev_dev);
  if (hipSetDevice(device) != hipSuccess) return cfail("vima_comm_create: hipSetDevice failed");
  UniqueId u;
  memcpy(u.internal, id, kIdBytes);
  VimaComm* c = new VimaComm();
  c->world = world;
  c->rank = rank;
  c->device = device;
  int r = g_rccl.CommInitRank(&c->comm, world, u, rank);
  if (prev_dev >= 0 && prev_dev != device) (void)hipSetDevice(prev_dev);
  if (r != 0) {
    delete c;
    return cfail(std::string("ncclCommInitRank: ") + g_rccl.GetErrorString(r));
  }
  *out = c;
  return 0;
}

int vima_comm_world(const VimaComm* c) { return c ? c->world : 0; }
int vima_comm_rank(const VimaComm* c) { return c ? c->rank : -1; }

int vima_allgather_logits(VimaComm* c, const float* local, float* global, int64_t rows_per_rank, int width,
                          vima_stream_t stream) {
  if (!c || !c->comm) return cfail("vima_allgather_logits: communicator not initialised");
  if (!local || !global || rows_per_rank < 0 || width <= 0) return cfail("vima_allgather_logits: bad argument");
  if (rows_per_rank == 0) return 0;
  NCK(g_rccl.AllGather(local, global, (size_t)rows_per_rank * (size_t)width, kFloat32, c->comm, (hipStream_t)stream));
  return 0;
}

void vima_comm_destroy(VimaComm* c) {
  if (!c) return;
  if (c->comm && g_rccl.CommDestroy) (void)g_rccl.CommDestroy(c->comm);
  delete c;
}

}  // extern "C"
