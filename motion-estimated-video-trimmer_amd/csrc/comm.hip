// comm.hip — the path's single multi-GPU exchange as C ABI: an all-gather of fixed-size
// per-rank blocks (segment lists / timestamps) over RCCL (xGMI inside a node).
// RCCL is loaded lazily with dlopen so that libmtgpu.so has no hard dependency on it and
// shares the copy a host process (e.g. PyTorch) may already have loaded.
#include <hip/hip_runtime.h>

#include <dlfcn.h>
#include <rccl/rccl.h>

#include <mutex>
#include <new>

#include "api_internal.h"

namespace {

using mtgpu::fail;
using mtgpu::hip_fail;

// RCCL's own types (ncclUniqueId, ncclComm_t, ncclDataType_t, ncclResult_t) come from its header; only the
// LIBRARY is optional (dlopen), so nothing of <rccl/rccl.h> is called directly — every entry point goes through
// a pointer typed with decltype of the header's declaration: a signature that drifts breaks the build, not a run.
static_assert(sizeof(ncclUniqueId) == MTGPU_UNIQUE_ID_BYTES, "include/mtgpu.h: MTGPU_UNIQUE_ID_BYTES must be sizeof(ncclUniqueId)");

struct Rccl {
  void *h = nullptr;
  decltype(&ncclGetUniqueId) GetUniqueId = nullptr;
  decltype(&ncclCommInitRank) CommInitRank = nullptr;
  decltype(&ncclCommDestroy) CommDestroy = nullptr;
  decltype(&ncclAllGather) AllGather = nullptr;
  decltype(&ncclGetErrorString) GetErrorString = nullptr;
};

Rccl g_rccl;
std::once_flag g_once;

bool load_rccl() {
  std::call_once(g_once, [] {
    void *h = dlopen("librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
    if (!h) h = dlopen("librccl.so", RTLD_NOW | RTLD_GLOBAL);
    if (!h) return;
    Rccl r;
    r.h = h;
    r.GetUniqueId = reinterpret_cast<decltype(r.GetUniqueId)>(dlsym(h, "ncclGetUniqueId"));
    r.CommInitRank = reinterpret_cast<decltype(r.CommInitRank)>(dlsym(h, "ncclCommInitRank"));
    r.CommDestroy = reinterpret_cast<decltype(r.CommDestroy)>(dlsym(h, "ncclCommDestroy"));
    r.AllGather = reinterpret_cast<decltype(r.AllGather)>(dlsym(h, "ncclAllGather"));
    r.GetErrorString = reinterpret_cast<decltype(r.GetErrorString)>(dlsym(h, "ncclGetErrorString"));
    if (r.GetUniqueId && r.CommInitRank && r.CommDestroy && r.AllGather) g_rccl = r;
  });
  return g_rccl.h != nullptr;
}

int rccl_fail(ncclResult_t rc, const char *what) {
  return fail(MT_ERR_DEVICE, "%s: %s", what, g_rccl.GetErrorString ? g_rccl.GetErrorString(rc) : "RCCL error");
}

}  // namespace

struct mtgpu_comm {
  ncclComm_t comm = nullptr;
  int rank = 0, n_ranks = 1, device = 0;
};

extern "C" {

int mtgpu_comm_unique_id(void *id_out) {
  if (!id_out) return fail(MT_ERR_INVALID, "id_out is NULL");
  if (!load_rccl()) return fail(MT_ERR_DEVICE, "librccl.so.1 not available: %s", dlerror() ? dlerror() : "");
  ncclUniqueId id;
  ncclResult_t rc = g_rccl.GetUniqueId(&id);
  if (rc != ncclSuccess) return rccl_fail(rc, "ncclGetUniqueId");
  __builtin_memcpy(id_out, id.internal, MTGPU_UNIQUE_ID_BYTES);
  return MT_OK;
}

int mtgpu_comm_create(int rank, int n_ranks, const void *id, int device, mtgpu_comm **out) {
  if (!out || !id) return fail(MT_ERR_INVALID, "NULL argument");
  *out = nullptr;
  if (n_ranks < 1 || rank < 0 || rank >= n_ranks) return fail(MT_ERR_INVALID, "rank %d of %d", rank, n_ranks);
  if (!load_rccl()) return fail(MT_ERR_DEVICE, "librccl.so.1 not available");
  device = mtgpu::physical_device(device);
  hipError_t e = hipSetDevice(device);
  if (e != hipSuccess) return hip_fail(e, "hipSetDevice");
  mtgpu_comm *c = new (std::nothrow) mtgpu_comm();
  if (!c) return fail(MT_ERR_NOMEM, "out of host memory");
  ncclUniqueId uid;
  __builtin_memcpy(uid.internal, id, MTGPU_UNIQUE_ID_BYTES);
  ncclResult_t rc = g_rccl.CommInitRank(&c->comm, n_ranks, uid, rank);
  if (rc != ncclSuccess) { delete c; return rccl_fail(rc, "ncclCommInitRank"); }
  c->rank = rank; c->n_ranks = n_ranks; c->device = device;
  *out = c;
  return MT_OK;
}

void mtgpu_comm_destroy(mtgpu_comm *c) {
  if (!c) return;
  (void)hipSetDevice(c->device);
  if (c->comm && g_rccl.CommDestroy) (void)g_rccl.CommDestroy(c->comm);
  delete c;
}

int mtgpu_gather_segments(mtgpu_comm *c, const void *d_send, uint64_t bytes_per_rank, void *d_recv,
                          void *stream) {
  if (!c) return fail(MT_ERR_INVALID, "comm is NULL");
  if (bytes_per_rank == 0) return MT_OK;
  if (!d_send || !d_recv) return fail(MT_ERR_INVALID, "NULL device pointer");
  hipError_t e = hipSetDevice(c->device);
  if (e != hipSuccess) return hip_fail(e, "hipSetDevice");
  ncclResult_t rc = g_rccl.AllGather(d_send, d_recv, (size_t)bytes_per_rank, ncclUint8, c->comm,
                                     static_cast<hipStream_t>(stream));
  if (rc != ncclSuccess) return rccl_fail(rc, "ncclAllGather");
  return MT_OK;
}

}  // extern "C"
