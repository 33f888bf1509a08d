// Native RCCL implementation of the two collectives of gsfm_rot_shard (include/gsfm_rot.h): in-place
// all-gather of per-camera slices and sum all-reduce of scalars, enqueued directly on the solver's HIP
// stream -- no Python in the PCG loop.  RCCL is resolved with dlopen so that the process uses ONE RCCL
// instance (the one PyTorch already loaded, when there is one).  Built into libgsfm_rccl.so.
#include <dlfcn.h>
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <cstdio>
#include <cstring>
#include <string>

namespace {
struct Api {
  void* handle = nullptr;
  ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
  ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
  ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
  ncclResult_t (*AllGather)(const void*, void*, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*AllReduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
  const char* (*GetErrorString)(ncclResult_t) = nullptr;
};
Api g_api;
thread_local std::string g_err;

struct Comm { ncclComm_t comm; int rank, world; };

template <typename T> bool sym(T& fn, const char* name) {
  fn = reinterpret_cast<T>(dlsym(g_api.handle, name));
  return fn != nullptr;
}
}  // namespace

extern "C" {

const char* gsfm_rccl_last_error(void) { return g_err.c_str(); }

// path may be NULL: tries the already-loaded library first, then librccl.so(.1) from the default search path.
int gsfm_rccl_init(const char* path) {
  if (g_api.handle) return 0;
  const char* candidates[4] = {path, "librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
  for (const char* c : candidates) {
    if (!c) continue;
    g_api.handle = dlopen(c, RTLD_NOW | RTLD_GLOBAL);
    if (g_api.handle) break;
  }
  if (!g_api.handle) { g_err = std::string("cannot dlopen RCCL: ") + dlerror(); return 1; }
  bool ok = sym(g_api.GetUniqueId, "ncclGetUniqueId") & sym(g_api.CommInitRank, "ncclCommInitRank") & sym(g_api.CommDestroy, "ncclCommDestroy") &
            sym(g_api.AllGather, "ncclAllGather") & sym(g_api.AllReduce, "ncclAllReduce") & sym(g_api.GetErrorString, "ncclGetErrorString");
  if (!ok) { g_err = "RCCL library lacks an expected symbol"; g_api.handle = nullptr; return 1; }
  return 0;
}

int gsfm_rccl_unique_id(char out[128]) {
  if (!g_api.handle && gsfm_rccl_init(nullptr)) return 1;
  static_assert(sizeof(ncclUniqueId) == 128, "ncclUniqueId size");
  ncclUniqueId id;
  const ncclResult_t r = g_api.GetUniqueId(&id);
  if (r != ncclSuccess) { g_err = g_api.GetErrorString(r); return 1; }
  std::memcpy(out, &id, 128);
  return 0;
}

// Collective over all ranks; uses the calling thread's current HIP device.
void* gsfm_rccl_create(const char id_bytes[128], int rank, int world) {
  if (!g_api.handle && gsfm_rccl_init(nullptr)) return nullptr;
  ncclUniqueId id;
  std::memcpy(&id, id_bytes, 128);
  Comm* c = new Comm{nullptr, rank, world};
  const ncclResult_t r = g_api.CommInitRank(&c->comm, world, id, rank);
  if (r != ncclSuccess) { g_err = std::string("ncclCommInitRank: ") + g_api.GetErrorString(r); delete c; return nullptr; }
  return c;
}
void gsfm_rccl_destroy(void* ctx) {
  Comm* c = static_cast<Comm*>(ctx);
  if (!c) return;
  if (c->comm) g_api.CommDestroy(c->comm);
  delete c;
}

// gsfm_rot_shard::all_gather : buf holds world*count doubles, rank r's slice already at buf + r*count
int gsfm_rccl_all_gather(void* ctx, double* buf, size_t count, void* stream) {
  Comm* c = static_cast<Comm*>(ctx);
  const ncclResult_t r = g_api.AllGather(buf + (size_t)c->rank * count, buf, count, ncclDouble, c->comm, (hipStream_t)stream);
  if (r != ncclSuccess) { g_err = std::string("ncclAllGather: ") + g_api.GetErrorString(r); fprintf(stderr, "gsfm_rccl: %s\n", g_err.c_str()); return 1; }
  return 0;
}
int gsfm_rccl_all_reduce_sum(void* ctx, double* buf, size_t count, void* stream) {
  Comm* c = static_cast<Comm*>(ctx);
  const ncclResult_t r = g_api.AllReduce(buf, buf, count, ncclDouble, ncclSum, c->comm, (hipStream_t)stream);
  if (r != ncclSuccess) { g_err = std::string("ncclAllReduce: ") + g_api.GetErrorString(r); fprintf(stderr, "gsfm_rccl: %s\n", g_err.c_str()); return 1; }
  return 0;
}

}  // extern "C"
