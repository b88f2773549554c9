// Data-parallel learners' collective behind the C ABI: one RCCL communicator per process, one rank per
// GPU, the flat fp32 gradient bucket averaged in place with ONE ncclAllReduce(ncclAvg) per minibatch on the learner's
// stream (capturable into the learn() hipGraph).  The reference has a single learner and no collective at all.
//
// RCCL is bound at run time (dlopen) so that single-GPU users of libjorldy_hip.so carry no dependency on it; inside a
// PyTorch process "librccl.so.1" resolves to the copy torch already loaded (same soname), otherwise to /opt/rocm/lib's.
#include <dlfcn.h>
#include <rccl/rccl.h>  // types and enums only: every call goes through the table below

#include "jh_common.h"

namespace {
struct Rccl {
  void* so = nullptr;
  ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
  ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
  ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
  ncclResult_t (*AllReduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*Broadcast)(const void*, void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*AllGather)(const void*, void*, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
  const char* (*GetErrorString)(ncclResult_t) = nullptr;
  std::string why;
};

Rccl* rccl() {
  static Rccl r;
  static std::once_flag once;
  std::call_once(once, [] {
    for (const char* name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
      r.so = dlopen(name, RTLD_NOW | RTLD_LOCAL);
      if (r.so) break;
    }
    if (!r.so) {
      r.why = std::string("RCCL not found (dlopen librccl.so.1): ") + (dlerror() ? dlerror() : "?");
      return;
    }
    auto sym = [&](const char* n) {
      void* p = dlsym(r.so, n);
      if (!p && r.why.empty()) r.why = std::string("RCCL symbol missing: ") + n;
      return p;
    };
    r.GetUniqueId = (decltype(r.GetUniqueId))sym("ncclGetUniqueId");
    r.CommInitRank = (decltype(r.CommInitRank))sym("ncclCommInitRank");
    r.CommDestroy = (decltype(r.CommDestroy))sym("ncclCommDestroy");
    r.AllReduce = (decltype(r.AllReduce))sym("ncclAllReduce");
    r.Broadcast = (decltype(r.Broadcast))sym("ncclBroadcast");
    r.AllGather = (decltype(r.AllGather))sym("ncclAllGather");
    r.GetErrorString = (decltype(r.GetErrorString))sym("ncclGetErrorString");
  });
  return &r;
}
}  // namespace

struct jh_comm {
  jh_ctx* ctx = nullptr;
  ncclComm_t comm = nullptr;
  int nranks = 0, rank = 0;
};

#define JH_NCCL(expr)                                                                                         \
  do {                                                                                                        \
    ncclResult_t _r = (expr);                                                                                 \
    if (_r != ncclSuccess)                                                                                    \
      return jh_fail(JH_ERR_HIP, "%s:%d %s -> %s", __FILE__, __LINE__, #expr, R->GetErrorString ? R->GetErrorString(_r) : "rccl error"); \
  } while (0)

static int need_rccl(Rccl** out) {
  Rccl* R = rccl();
  if (!R->so || !R->why.empty()) return jh_fail(JH_ERR_STATE, "%s", R->why.c_str());
  *out = R;
  return JH_OK;
}

JH_EXPORT int jh_comm_unique_id(void* h_id128) {
  JH_ARG(h_id128 != nullptr);
  Rccl* R;
  if (int rc = need_rccl(&R)) return rc;
  ncclUniqueId id;
  JH_NCCL(R->GetUniqueId(&id));
  static_assert(sizeof(id) == JH_COMM_ID_BYTES, "ncclUniqueId is 128 bytes");
  memcpy(h_id128, &id, sizeof(id));
  return JH_OK;
}

JH_EXPORT int jh_comm_create(jh_ctx* ctx, int32_t nranks, int32_t rank, const void* h_id128, jh_comm** out) {
  JH_ARG(ctx && h_id128 && out);
  JH_ARG(nranks >= 1 && rank >= 0 && rank < nranks);
  Rccl* R;
  if (int rc = need_rccl(&R)) return rc;
  JH_HIP(hipSetDevice(ctx->device));
  ncclUniqueId id;
  memcpy(&id, h_id128, sizeof(id));
  ncclComm_t c = nullptr;
  JH_NCCL(R->CommInitRank(&c, nranks, id, rank));
  jh_comm* m = new jh_comm();
  m->ctx = ctx;
  m->comm = c;
  m->nranks = nranks;
  m->rank = rank;
  *out = m;
  return JH_OK;
}

JH_EXPORT void jh_comm_destroy(jh_comm* m) {
  if (!m) return;
  Rccl* R = rccl();
  if (m->comm && R->CommDestroy) (void)R->CommDestroy(m->comm);
  delete m;
}

JH_EXPORT int jh_comm_info(const jh_comm* m, int32_t* nranks, int32_t* rank) {
  JH_ARG(m != nullptr);
  if (nranks) *nranks = m->nranks;
  if (rank) *rank = m->rank;
  return JH_OK;
}

JH_EXPORT int jh_comm_allreduce_mean_f32(jh_comm* m, float* d_bucket, int64_t n, jh_stream stream) {
  JH_ARG(m && d_bucket && n > 0);
  Rccl* R;
  if (int rc = need_rccl(&R)) return rc;
  JH_NCCL(R->AllReduce(d_bucket, d_bucket, (size_t)n, ncclFloat32, ncclAvg, m->comm, jh_s(stream)));
  return JH_OK;
}

JH_EXPORT int jh_comm_broadcast(jh_comm* m, void* d_buf, int64_t bytes, int32_t root, jh_stream stream) {
  JH_ARG(m && d_buf && bytes > 0 && root >= 0 && root < m->nranks);
  Rccl* R;
  if (int rc = need_rccl(&R)) return rc;
  JH_NCCL(R->Broadcast(d_buf, d_buf, (size_t)bytes, ncclUint8, root, m->comm, jh_s(stream)));
  return JH_OK;
}

JH_EXPORT int jh_comm_allgather_f64(jh_comm* m, const double* d_in, double* d_out, int64_t n, jh_stream stream) {
  JH_ARG(m && d_in && d_out && n > 0);
  Rccl* R;
  if (int rc = need_rccl(&R)) return rc;
  JH_NCCL(R->AllGather(d_in, d_out, (size_t)n, ncclFloat64, m->comm, jh_s(stream)));
  return JH_OK;
}
