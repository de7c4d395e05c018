// Collectives under the C ABI: RCCL over xGMI, one process per GPU.
//
// The reference's CleanRL path has no collective; its skrl front-end broadcasts the parameters at start-up
// (skrl/ppo.py:126-131), averages the gradients after backward (:534-537) and all-reduces the KL estimate
// (:562-564) through torch.distributed/NCCL.  Here the same exchange points call RCCL directly, on the caller's
// stream, so that (a) a consumer of libcatppo.so needs no torch.distributed, and (b) the all-reduce of an optimiser
// step is an ordinary stream operation that a hipGraph capture records together with the kernels around it.
//
// librccl is loaded with dlopen on first use: single-GPU users never map it (it is a 570 MB library), and a
// process that already carries an RCCL (PyTorch bundles one under the same soname) shares that instance.
#include "common.h"

#include <dlfcn.h>
#include <rccl/rccl.h>

namespace {

struct RcclApi {
  void* handle = nullptr;
  ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
  ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
  ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
  ncclResult_t (*AllReduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*Broadcast)(const void*, void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*AllGather)(const void*, void*, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*GroupStart)() = nullptr;
  ncclResult_t (*GroupEnd)() = nullptr;
  const char* (*GetErrorString)(ncclResult_t) = nullptr;
  char why[256] = {0};
};

RcclApi& rccl_state() {
  static RcclApi api;
  return api;
}

RcclApi* rccl() {
  RcclApi& api = rccl_state();
  static bool tried = false;
  if (tried) return api.handle ? &api : nullptr;
  tried = true;
  const char* names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
  void* h = nullptr;
  for (const char* n : names) {
    h = dlopen(n, RTLD_NOW | RTLD_GLOBAL);
    if (h) break;
  }
  if (!h) {
    snprintf(api.why, sizeof(api.why), "dlopen(librccl.so.1) failed: %s", dlerror());
    return nullptr;
  }
  auto sym = [&](const char* n) { return dlsym(h, n); };
  api.GetUniqueId = reinterpret_cast<decltype(api.GetUniqueId)>(sym("ncclGetUniqueId"));
  api.CommInitRank = reinterpret_cast<decltype(api.CommInitRank)>(sym("ncclCommInitRank"));
  api.CommDestroy = reinterpret_cast<decltype(api.CommDestroy)>(sym("ncclCommDestroy"));
  api.AllReduce = reinterpret_cast<decltype(api.AllReduce)>(sym("ncclAllReduce"));
  api.Broadcast = reinterpret_cast<decltype(api.Broadcast)>(sym("ncclBroadcast"));
  api.AllGather = reinterpret_cast<decltype(api.AllGather)>(sym("ncclAllGather"));
  api.GroupStart = reinterpret_cast<decltype(api.GroupStart)>(sym("ncclGroupStart"));
  api.GroupEnd = reinterpret_cast<decltype(api.GroupEnd)>(sym("ncclGroupEnd"));
  api.GetErrorString = reinterpret_cast<decltype(api.GetErrorString)>(sym("ncclGetErrorString"));
  if (!api.GetUniqueId || !api.CommInitRank || !api.CommDestroy || !api.AllReduce || !api.Broadcast || !api.AllGather ||
      !api.GroupStart || !api.GroupEnd) {
    snprintf(api.why, sizeof(api.why), "librccl is missing an entry point");
    dlclose(h);
    return nullptr;
  }
  api.handle = h;
  return &api;
}

// why rccl() returned nullptr
const char* why_not() {
  const RcclApi& api = rccl_state();
  return api.why[0] ? api.why : "librccl could not be loaded";
}

int comm_fail(catppo_ctx* ctx, RcclApi* r, const char* what, ncclResult_t rc) {
  return catppo_fail(ctx, CATPPO_E_COMM, "%s: %s", what, (r && r->GetErrorString) ? r->GetErrorString(rc) : "rccl error");
}

bool dtype_of(int dtype, ncclDataType_t* out) {
  if (dtype == CATPPO_F32) *out = ncclFloat32;
  else if (dtype == CATPPO_F64) *out = ncclFloat64;
  else if (dtype == CATPPO_F16) *out = ncclFloat16;
  else return false;
  return true;
}

static_assert(sizeof(ncclUniqueId) == CATPPO_UNIQUE_ID_BYTES, "ncclUniqueId is 128 bytes");

}  // namespace

// ctx == NULL: can librccl be loaded (CATPPO_OK / CATPPO_E_COMM)?  ctx != NULL: the world size of its communicator (>= 1), 0 when it
// has none and librccl is loadable, CATPPO_E_COMM when it is not.  (ABI 0.6: replaces catppo_comm_probe(void) + catppo_comm_world)
extern "C" int catppo_comm_probe(catppo_ctx* ctx) {
  if (ctx && ctx->comm) return ctx->comm_world;
  return rccl() ? CATPPO_OK : CATPPO_E_COMM;
}

extern "C" int catppo_comm_unique_id(uint8_t* out128) {
  if (!out128) return CATPPO_E_ARG;
  RcclApi* r = rccl();
  if (!r) return CATPPO_E_COMM;
  ncclUniqueId id;
  if (r->GetUniqueId(&id) != ncclSuccess) return CATPPO_E_COMM;
  memcpy(out128, &id, sizeof(id));
  return CATPPO_OK;
}

extern "C" int catppo_comm_init(catppo_ctx* ctx, int rank, int world, const uint8_t* unique_id128) {
  CATPPO_CHECK_ARG(ctx, ctx != nullptr);
  CATPPO_CHECK_ARG(ctx, unique_id128 != nullptr && world >= 1 && rank >= 0 && rank < world);
  CATPPO_CHECK_ARG(ctx, ctx->comm == nullptr);
  RcclApi* r = rccl();
  if (!r) return catppo_fail(ctx, CATPPO_E_COMM, "catppo_comm_init: %s", why_not());
  ncclUniqueId id;
  memcpy(&id, unique_id128, sizeof(id));
  int cur = 0;
  (void)hipGetDevice(&cur);
  (void)hipSetDevice(ctx->device);
  ncclComm_t c = nullptr;
  const ncclResult_t rc = r->CommInitRank(&c, world, id, rank);
  (void)hipSetDevice(cur);
  if (rc != ncclSuccess) return comm_fail(ctx, r, "ncclCommInitRank", rc);
  ctx->comm = c;
  ctx->comm_rank = rank;
  ctx->comm_world = world;
  return CATPPO_OK;
}

extern "C" int catppo_comm_destroy(catppo_ctx* ctx) {
  if (!ctx) return CATPPO_E_ARG;
  if (ctx->comm) {
    if (RcclApi* r = rccl()) (void)r->CommDestroy(static_cast<ncclComm_t>(ctx->comm));
    ctx->comm = nullptr;
    ctx->comm_world = 0;
  }
  return CATPPO_OK;
}

extern "C" int catppo_allreduce(catppo_ctx* ctx, void* buf, int64_t count, int dtype, int op, void* stream) {
  CATPPO_CHECK_ARG(ctx, ctx != nullptr);
  CATPPO_CHECK_ARG(ctx, buf != nullptr && count >= 1 && (op == CATPPO_SUM || op == CATPPO_MAX));
  if (!ctx->comm) return catppo_fail(ctx, CATPPO_E_COMM, "catppo_allreduce: no communicator (catppo_comm_init)");
  ncclDataType_t dt;
  CATPPO_CHECK_ARG(ctx, dtype_of(dtype, &dt));
  RcclApi* r = rccl();
  const ncclResult_t rc = r->AllReduce(buf, buf, (size_t)count, dt, op == CATPPO_SUM ? ncclSum : ncclMax,
                                       static_cast<ncclComm_t>(ctx->comm), static_cast<hipStream_t>(stream));
  if (rc != ncclSuccess) return comm_fail(ctx, r, "ncclAllReduce", rc);
  return CATPPO_OK;
}

extern "C" int catppo_broadcast(catppo_ctx* ctx, void* buf, int64_t count, int dtype, int root, void* stream) {
  CATPPO_CHECK_ARG(ctx, ctx != nullptr);
  CATPPO_CHECK_ARG(ctx, buf != nullptr && count >= 1);
  if (!ctx->comm) return catppo_fail(ctx, CATPPO_E_COMM, "catppo_broadcast: no communicator (catppo_comm_init)");
  CATPPO_CHECK_ARG(ctx, root >= 0 && root < ctx->comm_world);
  ncclDataType_t dt;
  CATPPO_CHECK_ARG(ctx, dtype_of(dtype, &dt));
  RcclApi* r = rccl();
  const ncclResult_t rc = r->Broadcast(buf, buf, (size_t)count, dt, root, static_cast<ncclComm_t>(ctx->comm),
                                       static_cast<hipStream_t>(stream));
  if (rc != ncclSuccess) return comm_fail(ctx, r, "ncclBroadcast", rc);
  return CATPPO_OK;
}

extern "C" int catppo_allgather(catppo_ctx* ctx, const void* send, void* recv, int64_t bytes, void* stream) {
  CATPPO_CHECK_ARG(ctx, ctx != nullptr);
  CATPPO_CHECK_ARG(ctx, send != nullptr && recv != nullptr && bytes >= 1);
  if (!ctx->comm) return catppo_fail(ctx, CATPPO_E_COMM, "catppo_allgather: no communicator (catppo_comm_init)");
  RcclApi* r = rccl();
  const ncclResult_t rc = r->AllGather(send, recv, (size_t)bytes, ncclUint8, static_cast<ncclComm_t>(ctx->comm),
                                       static_cast<hipStream_t>(stream));
  if (rc != ncclSuccess) return comm_fail(ctx, r, "ncclAllGather", rc);
  return CATPPO_OK;
}

// ---- gradient buckets (ABI 0.4) ------------------------------------------------------------------------------------
// The reference's distributed front end reduces every gradient after backward() (skrl/ppo.py:534-537).  Here the flat
// gradient is reduced in per-layer buckets on the context's side stream while the backward launches of the layers
// below still run (catppo_ppo_minibatch_grad*, see mlp.hip): the ranges of one bucket go out as ONE grouped operation.
int catppo_internal_allreduce_ranges(catppo_ctx* ctx, float* base, const int64_t* off, const int64_t* cnt, int n,
                                     hipStream_t stream) {
  if (!ctx->comm) return catppo_fail(ctx, CATPPO_E_COMM, "gradient bucket all-reduce: no communicator");
  RcclApi* r = rccl();
  ncclResult_t rc = n > 1 ? r->GroupStart() : ncclSuccess;
  if (rc != ncclSuccess) return comm_fail(ctx, r, "ncclGroupStart", rc);
  ncclResult_t first_bad = ncclSuccess;
  for (int i = 0; i < n; ++i) {
    if (cnt[i] <= 0) continue;
    rc = r->AllReduce(base + off[i], base + off[i], (size_t)cnt[i], ncclFloat32, ncclSum,
                      static_cast<ncclComm_t>(ctx->comm), stream);
    if (rc != ncclSuccess && first_bad == ncclSuccess) first_bad = rc;
  }
  if (n > 1) {
    rc = r->GroupEnd();      // always closed, also after a failed member call
    if (rc != ncclSuccess && first_bad == ncclSuccess) first_bad = rc;
  }
  if (first_bad != ncclSuccess) return comm_fail(ctx, r, "ncclAllReduce (gradient bucket)", first_bad);
  return CATPPO_OK;
}

// on = 0 | 1 | 2: set the mode (returns CATPPO_OK).  on = -1: query - returns 1 when a mode is set AND a communicator exists (the
// all-reduce really runs inside the gradient call), else 0.  (ABI 0.6: the query replaces catppo_grad_overlap_active)
extern "C" int catppo_set_grad_overlap(catppo_ctx* ctx, int on) {
  CATPPO_CHECK_ARG(ctx, ctx != nullptr);
  if (on == -1) return (ctx->grad_overlap && ctx->comm != nullptr) ? 1 : 0;
  CATPPO_CHECK_ARG(ctx, on == 0 || on == 1 || on == 2);
  if (on && ctx->use_side)
    return catppo_fail(ctx, CATPPO_E_ARG, "catppo_set_grad_overlap: the side stream is taken by CATPPO_SIDE_STREAM=1");
  ctx->grad_overlap = on;
  return CATPPO_OK;
}

