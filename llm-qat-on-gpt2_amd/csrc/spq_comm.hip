// The one exchange step of the path: all ranks merge the running min/max of the input quantizers before
// finish_calibration, as a single in-place all-reduce(MAX) over the flat fp32 buffer [-min | max].  RCCL is bound at run
// time (dlopen of librccl.so.1: the copy the process already holds -- torch ships one -- or the ROCm one), so the
// kernels of libspq.so do not depend on it.
#include <dlfcn.h>
#include <string.h>
#include <mutex>

#include "spq_common.h"

namespace spq {
namespace {

// the slice of rccl.h this file uses (values from /opt/rocm/include/rccl/rccl.h: ncclFloat32 = 7, ncclMax = 2)
struct UniqueId { char internal[SPQ_COMM_ID_BYTES]; };
typedef void* Comm;
typedef int (*GetUniqueIdFn)(UniqueId*);
typedef int (*CommInitRankFn)(Comm*, int, UniqueId, int);
typedef int (*CommDestroyFn)(Comm);
typedef int (*AllReduceFn)(const void*, void*, size_t, int, int, Comm, hipStream_t);
typedef const char* (*ErrStrFn)(int);
constexpr int kFloat32 = 7, kMax = 2;

struct Rccl {
  void* h = nullptr;
  GetUniqueIdFn get_id = nullptr;
  CommInitRankFn init = nullptr;
  CommDestroyFn destroy = nullptr;
  AllReduceFn allreduce = nullptr;
  ErrStrFn errstr = nullptr;
};

Rccl g_rccl;
std::once_flag g_once;

void load_rccl() {
  void* h = dlopen("librccl.so.1", RTLD_NOW | RTLD_NOLOAD);   // already mapped (torch's copy)?
  if (!h) h = dlopen("librccl.so.1", RTLD_NOW | RTLD_LOCAL);
  if (!h) h = dlopen("/opt/rocm/lib/librccl.so.1", RTLD_NOW | RTLD_LOCAL);
  if (!h) return;
  Rccl r;
  r.h = h;
  r.get_id = (GetUniqueIdFn)dlsym(h, "ncclGetUniqueId");
  r.init = (CommInitRankFn)dlsym(h, "ncclCommInitRank");
  r.destroy = (CommDestroyFn)dlsym(h, "ncclCommDestroy");
  r.allreduce = (AllReduceFn)dlsym(h, "ncclAllReduce");
  r.errstr = (ErrStrFn)dlsym(h, "ncclGetErrorString");
  if (r.get_id && r.init && r.destroy && r.allreduce) g_rccl = r;
}

const Rccl* rccl() {
  std::call_once(g_once, load_rccl);
  if (!g_rccl.h) {
    set_error("RCCL not available: dlopen(librccl.so.1) failed (%s)", dlerror() ? dlerror() : "symbols missing");
    return nullptr;
  }
  return &g_rccl;
}

int rccl_fail(const Rccl* r, const char* what, int rc) {
  set_error("%s: RCCL error %d (%s)", what, rc, r->errstr ? r->errstr(rc) : "?");
  return SPQ_ERR_LAUNCH;
}

}  // namespace
}  // namespace spq

using namespace spq;

extern "C" int spq_comm_unique_id(void* id_out) {
  SPQ_REQUIRE(id_out, "spq_comm_unique_id: null buffer");
  const Rccl* r = rccl();
  if (!r) return SPQ_ERR_UNSUPPORTED;
  UniqueId id;
  int rc = r->get_id(&id);
  if (rc) return rccl_fail(r, "spq_comm_unique_id", rc);
  memcpy(id_out, &id, sizeof(id));
  return SPQ_OK;
}

extern "C" int spq_comm_init(int rank, int nranks, const void* unique_id, spq_comm_t* comm_out) {
  SPQ_REQUIRE(unique_id && comm_out, "spq_comm_init: null pointer");
  SPQ_REQUIRE(nranks >= 1 && rank >= 0 && rank < nranks, "spq_comm_init: rank %d outside [0,%d)", rank, nranks);
  const Rccl* r = rccl();
  if (!r) return SPQ_ERR_UNSUPPORTED;
  UniqueId id;
  memcpy(&id, unique_id, sizeof(id));
  Comm c = nullptr;
  int rc = r->init(&c, nranks, id, rank);     // collective over the ranks; binds the calling thread's current device
  if (rc) return rccl_fail(r, "spq_comm_init", rc);
  *comm_out = c;
  return SPQ_OK;
}

extern "C" int spq_comm_destroy(spq_comm_t comm) {
  if (!comm) return SPQ_OK;
  const Rccl* r = rccl();
  if (!r) return SPQ_ERR_UNSUPPORTED;
  int rc = r->destroy((Comm)comm);
  return rc ? rccl_fail(r, "spq_comm_destroy", rc) : SPQ_OK;
}

extern "C" int spq_allreduce_minmax(spq_comm_t comm, float* neg_min_and_max, size_t len, spq_stream_t stream) {
  SPQ_REQUIRE(comm && neg_min_and_max, "spq_allreduce_minmax: null pointer");
  if (len == 0) return SPQ_OK;
  const Rccl* r = rccl();
  if (!r) return SPQ_ERR_UNSUPPORTED;
  int rc = r->allreduce(neg_min_and_max, neg_min_and_max, len, kFloat32, kMax, (Comm)comm, (hipStream_t)stream);
  return rc ? rccl_fail(r, "spq_allreduce_minmax", rc) : SPQ_OK;
}
