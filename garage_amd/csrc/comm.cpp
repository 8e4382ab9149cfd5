// RCCL communicator for the data-parallel gradient all-reduce (one per rank,
// one rank per GPU, xGMI inside the node).  RCCL is loaded lazily with dlopen so
// single-GPU runs and the CPU-side ABI checks do not need it.
#include <dlfcn.h>
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>
#include <stdint.h>
#include <string.h>

void ga_set_error(const char* fmt, ...);
typedef int (*ga_allreduce_fn)(void* comm, float* buf, int64_t n, void* stream);
extern "C" void ga_set_allreduce_hook(ga_allreduce_fn fn);

namespace {
struct Api {
  void* handle = nullptr;
  ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
  ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
  ncclResult_t (*AllReduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t,
                            ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
  ncclResult_t (*CommCount)(const ncclComm_t, int*) = nullptr;
  const char* (*GetErrorString)(ncclResult_t) = nullptr;
} g;

bool load_api() {
  if (g.handle) return true;
  g.handle = dlopen("librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
  if (!g.handle) g.handle = dlopen("librccl.so", RTLD_NOW | RTLD_GLOBAL);
  if (!g.handle) {
    ga_set_error("ga_comm: cannot load librccl: %s", dlerror());
    return false;
  }
  g.GetUniqueId = (decltype(g.GetUniqueId))dlsym(g.handle, "ncclGetUniqueId");
  g.CommInitRank = (decltype(g.CommInitRank))dlsym(g.handle, "ncclCommInitRank");
  g.AllReduce = (decltype(g.AllReduce))dlsym(g.handle, "ncclAllReduce");
  g.CommDestroy = (decltype(g.CommDestroy))dlsym(g.handle, "ncclCommDestroy");
  g.CommCount = (decltype(g.CommCount))dlsym(g.handle, "ncclCommCount");
  g.GetErrorString =
      (decltype(g.GetErrorString))dlsym(g.handle, "ncclGetErrorString");
  if (!g.GetUniqueId || !g.CommInitRank || !g.AllReduce || !g.CommDestroy) {
    ga_set_error("ga_comm: librccl lacks a required symbol");
    return false;
  }
  return true;
}

int allreduce_hook(void* comm, float* buf, int64_t n, void* stream) {
  const ncclResult_t r = g.AllReduce(buf, buf, (size_t)n, ncclFloat, ncclSum,
                                     (ncclComm_t)comm, (hipStream_t)stream);
  return r == ncclSuccess ? 0 : (int)r;
}
}  // namespace

static_assert(sizeof(ncclUniqueId) == 128, "ncclUniqueId is 128 bytes");

extern "C" int ga_comm_available(void) { return load_api() ? 1 : 0; }

extern "C" int ga_comm_unique_id(void* id128_host) {
  if (!id128_host || !load_api()) return -1;
  ncclUniqueId id;
  const ncclResult_t r = g.GetUniqueId(&id);
  if (r != ncclSuccess) {
    ga_set_error("ncclGetUniqueId failed: %d", (int)r);
    return -2;
  }
  memcpy(id128_host, &id, sizeof(id));
  return 0;
}

extern "C" void* ga_comm_init_rank(const void* id128_host, int rank, int world) {
  if (!id128_host || !load_api()) return nullptr;
  ncclUniqueId id;
  memcpy(&id, id128_host, sizeof(id));
  ncclComm_t comm = nullptr;
  const ncclResult_t r = g.CommInitRank(&comm, world, id, rank);
  if (r != ncclSuccess) {
    ga_set_error("ncclCommInitRank failed: %d", (int)r);
    return nullptr;
  }
  ga_set_allreduce_hook(allreduce_hook);
  return (void*)comm;
}

extern "C" int ga_comm_allreduce_sum_f32(void* comm, float* buf, int64_t n,
                                         hipStream_t stream) {
  if (!comm || !buf || n <= 0 || !load_api()) {
    ga_set_error("ga_comm_allreduce_sum_f32: bad arguments");
    return -1;
  }
  return allreduce_hook(comm, buf, n, (void*)stream);
}

extern "C" int ga_comm_count(void* comm) {
  if (!comm || !load_api() || !g.CommCount) {
    ga_set_error("ga_comm_count: bad arguments");
    return -1;
  }
  int n = 0;
  if (g.CommCount((ncclComm_t)comm, &n) != ncclSuccess) {
    ga_set_error("ncclCommCount failed");
    return -2;
  }
  return n;
}

extern "C" int ga_comm_destroy(void* comm) {
  if (!comm || !load_api()) return -1;
  return g.CommDestroy((ncclComm_t)comm) == ncclSuccess ? 0 : -2;
}
