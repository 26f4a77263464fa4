// comm.cpp — gradient exchange of the per-image data-parallel step over RCCL / xGMI, behind the C ABI.
//
// Replaces keras.utils.multi_gpu_model (utils.py:209-211): the reference builds in-graph towers and merges on the
// CPU; here every process owns one GPU and ONE all-reduce(sum) of the flat fp32 gradient arena crosses xGMI per step
// (SURVEY §8e).  RCCL is bound at run time (dlopen + dlsym) so that libdl3.so loads on a box without it and always
// shares the RCCL instance the process already holds (PyTorch ships one; two copies in one process would each keep
// their own topology / IPC state).  No allocation, no synchronisation: collectives are enqueued on the caller's stream.
#include <dlfcn.h>
#include <string.h>

#include <rccl/rccl.h>

#include "common.h"

namespace {

struct Rccl {
  void *h = nullptr;
  ncclResult_t (*GetUniqueId)(ncclUniqueId *) = nullptr;
  ncclResult_t (*CommInitRank)(ncclComm_t *, int, ncclUniqueId, int) = nullptr;
  ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
  ncclResult_t (*AllReduce)(const void *, void *, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*Broadcast)(const void *, void *, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
  const char *(*GetErrorString)(ncclResult_t) = nullptr;
  ncclResult_t (*CommCount)(const ncclComm_t, int *) = nullptr;
  bool ok = false;
};

Rccl &rccl() {
  static Rccl R;
  static bool tried = false;
  if (tried) return R;
  tried = true;
  // the copy already mapped into the process first (RTLD_NOLOAD), then the system one
  const char *names[] = {"librccl.so", "librccl.so.1", "/opt/rocm/lib/librccl.so.1"};
  for (const char *n : names) {
    R.h = dlopen(n, RTLD_NOW | RTLD_NOLOAD);
    if (R.h) break;
  }
  for (int i = 0; !R.h && i < 3; i++) R.h = dlopen(names[i], RTLD_NOW | RTLD_GLOBAL);
  if (!R.h) return R;
  R.GetUniqueId = (decltype(R.GetUniqueId))dlsym(R.h, "ncclGetUniqueId");
  R.CommInitRank = (decltype(R.CommInitRank))dlsym(R.h, "ncclCommInitRank");
  R.CommDestroy = (decltype(R.CommDestroy))dlsym(R.h, "ncclCommDestroy");
  R.AllReduce = (decltype(R.AllReduce))dlsym(R.h, "ncclAllReduce");
  R.Broadcast = (decltype(R.Broadcast))dlsym(R.h, "ncclBroadcast");
  R.GetErrorString = (decltype(R.GetErrorString))dlsym(R.h, "ncclGetErrorString");
  R.CommCount = (decltype(R.CommCount))dlsym(R.h, "ncclCommCount");
  R.ok = R.GetUniqueId && R.CommInitRank && R.CommDestroy && R.AllReduce && R.Broadcast && R.GetErrorString;
  return R;
}

struct Comm {
  ncclComm_t c;
  int rank, world;
};

int need_rccl(const char *who) {
  if (!rccl().ok) {
    const char *e = dlerror();  // one call: dlerror() clears the state it reports
    dl3_set_error("%s: librccl.so could not be loaded (%s)", who, e ? e : "symbols missing");
    return DL3_EUNSUPPORTED;
  }
  return DL3_OK;
}

#define DL3_NCCL(call, who)                                                  \
  do {                                                                       \
    ncclResult_t r__ = (call);                                               \
    if (r__ != ncclSuccess) {                                                \
      dl3_set_error("%s: RCCL error: %s", who, rccl().GetErrorString(r__));  \
      return DL3_EHIP;                                                       \
    }                                                                        \
  } while (0)

}  // namespace

extern "C" int dl3_comm_unique_id(void *id128) {
  DL3_CHECK_ARG(id128, "comm_unique_id: null pointer");
  static_assert(sizeof(ncclUniqueId) == DL3_COMM_ID_BYTES, "ncclUniqueId is 128 bytes");
  if (int rc = need_rccl("comm_unique_id")) return rc;
  ncclUniqueId id;
  DL3_NCCL(rccl().GetUniqueId(&id), "comm_unique_id");
  memcpy(id128, &id, sizeof(id));
  return DL3_OK;
}

extern "C" int dl3_comm_init(void **comm, const void *id128, int rank, int world) {
  DL3_CHECK_ARG(comm && id128 && world >= 1 && rank >= 0 && rank < world, "comm_init: bad argument");
  if (int rc = need_rccl("comm_init")) return rc;
  ncclUniqueId id;
  memcpy(&id, id128, sizeof(id));
  Comm *c = new Comm{nullptr, rank, world};
  ncclResult_t r = rccl().CommInitRank(&c->c, world, id, rank);  // binds to the calling thread's current HIP device
  if (r != ncclSuccess) {
    dl3_set_error("comm_init: RCCL error: %s", rccl().GetErrorString(r));
    delete c;
    return DL3_EHIP;
  }
  *comm = c;
  return DL3_OK;
}

extern "C" int dl3_comm_allreduce_f32(void *comm, const float *send, float *recv, size_t n, void *stream) {
  DL3_CHECK_ARG(comm && send && recv && n > 0, "comm_allreduce_f32: bad argument");
  Comm *c = (Comm *)comm;
  DL3_NCCL(rccl().AllReduce(send, recv, n, ncclFloat, ncclSum, c->c, (hipStream_t)stream), "comm_allreduce_f32");
  return DL3_OK;
}

extern "C" int dl3_comm_broadcast_f32(void *comm, float *buf, size_t n, int root, void *stream) {
  DL3_CHECK_ARG(comm && buf && n > 0, "comm_broadcast_f32: bad argument");
  Comm *c = (Comm *)comm;
  DL3_CHECK_ARG(root >= 0 && root < c->world, "comm_broadcast_f32: bad root");
  DL3_NCCL(rccl().Broadcast(buf, buf, n, ncclFloat, root, c->c, (hipStream_t)stream), "comm_broadcast_f32");
  return DL3_OK;
}

extern "C" int dl3_comm_count(void *comm, int *count) {
  DL3_CHECK_ARG(comm && count, "comm_count: null pointer");
  Comm *c = (Comm *)comm;
  DL3_UNSUPPORTED(!rccl().CommCount, "comm_count: this librccl does not export ncclCommCount");
  DL3_NCCL(rccl().CommCount(c->c, count), "comm_count");
  return DL3_OK;
}

extern "C" int dl3_comm_destroy(void *comm) {
  if (!comm) return DL3_OK;
  Comm *c = (Comm *)comm;
  ncclResult_t r = rccl().CommDestroy(c->c);
  delete c;
  if (r != ncclSuccess) {
    dl3_set_error("comm_destroy: RCCL error: %s", rccl().GetErrorString(r));
    return DL3_EHIP;
  }
  return DL3_OK;
}
