// libcatppo lifecycle: context, workspace, error text.
#include "common.h"

#include <cstdlib>

extern "C" int catppo_version(void) { return CATPPO_VERSION; }

extern "C" const char* catppo_plan_log(catppo_ctx* ctx, int enable) {
  if (!ctx) return "";
  if (enable > 0) {            // start recording (clears the log)
    ctx->plan_on = true, ctx->plan_len = 0, ctx->plan[0] = 0;
  } else if (enable == 0) {    // stop recording, keep the text
    ctx->plan_on = false;
  }                            // enable < 0: just read
  return ctx->plan;
}

extern "C" int catppo_create(int device, catppo_ctx** out) {
  if (!out) return CATPPO_E_ARG;
  *out = nullptr;
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess || n <= 0 || device < 0 || device >= n) return CATPPO_E_NODEV;
  hipDeviceProp_t prop;
  if (hipGetDeviceProperties(&prop, device) != hipSuccess) return CATPPO_E_NODEV;
  // gfx950 only: the kernels are written for CDNA4 (wave64, fp32 MFMA shapes, 160 KiB LDS)
  if (strncmp(prop.gcnArchName, "gfx950", 6) != 0) return CATPPO_E_NODEV;
  catppo_ctx* ctx = new catppo_ctx();
  ctx->device = device;
  ctx->n_cu = prop.multiProcessorCount;
  // measured on MI355X: forking the weight-gradient GEMMs to a side stream is 5% SLOWER for the whole
  // iteration (both chains are MFMA bound; they only steal each other's CUs) - off unless asked for
  if (const char* e = getenv("CATPPO_SIDE_STREAM")) ctx->use_side = (e[0] == '1');
  *out = ctx;
  int cur = 0;
  (void)hipGetDevice(&cur);
  (void)hipSetDevice(device);
  bool ok = hipStreamCreateWithFlags(&ctx->side, hipStreamNonBlocking) == hipSuccess;
  for (auto& e : ctx->ev_fork) ok = ok && hipEventCreateWithFlags(&e, hipEventDisableTiming) == hipSuccess;
  ok = ok && hipEventCreateWithFlags(&ctx->ev_join, hipEventDisableTiming) == hipSuccess;
  ok = ok && hipMalloc(reinterpret_cast<void**>(&ctx->tickets), sizeof(unsigned int) * catppo_ctx::kTickets) == hipSuccess;
  ok = ok && hipMemset(ctx->tickets, 0, sizeof(unsigned int) * catppo_ctx::kTickets) == hipSuccess;
  (void)hipSetDevice(cur);
  if (!ok) {
    delete ctx;
    *out = nullptr;
    return CATPPO_E_HIP;
  }
  // default workspace: enough for the reduction partials of every non-MLP call
  if (int rc = catppo_reserve(ctx, 8ull << 20)) {
    delete ctx;
    *out = nullptr;
    return rc;
  }
  return CATPPO_OK;
}

extern "C" void catppo_destroy(catppo_ctx* ctx) {
  if (!ctx) return;
  int cur = 0;
  (void)hipGetDevice(&cur);
  (void)hipSetDevice(ctx->device);
  (void)catppo_comm_destroy(ctx);
  for (auto& g : ctx->graphs)
    if (g) (void)hipGraphExecDestroy(g);
  if (ctx->tickets) (void)hipFree(ctx->tickets);
  if (ctx->post_rpart) (void)hipFree(ctx->post_rpart);
  if (ctx->ws) (void)hipFree(ctx->ws);
  if (ctx->side) (void)hipStreamDestroy(ctx->side);
  for (auto& e : ctx->ev_fork)
    if (e) (void)hipEventDestroy(e);
  if (ctx->ev_join) (void)hipEventDestroy(ctx->ev_join);
  if (ctx->ev_tail) (void)hipEventDestroy(ctx->ev_tail);
  (void)hipSetDevice(cur);
  delete ctx;
}

extern "C" const char* catppo_last_error(catppo_ctx* ctx) { return ctx ? ctx->err : "null ctx"; }

extern "C" int catppo_reserve(catppo_ctx* ctx, uint64_t bytes) {
  if (!ctx) return CATPPO_E_ARG;
  if (bytes <= ctx->ws_bytes) return CATPPO_OK;
  if (ctx->capturing)
    return catppo_fail(ctx, CATPPO_E_ARG, "catppo_reserve: cannot grow the workspace while a graph capture is active "
                                          "(reserve %llu B before catppo_graph_begin)", (unsigned long long)bytes);
  int cur = 0;
  (void)hipGetDevice(&cur);
  (void)hipSetDevice(ctx->device);
  // the old block may still be referenced by enqueued kernels: drain before freeing
  (void)hipDeviceSynchronize();
  // captured graphs hold pointers into the old block: drop them BEFORE it is freed (a graph that outlived its
  // workspace - e.g. when the hipMalloc below fails - would replay through dangling pointers; without a graph
  // catppo_graph_launch fails and the caller captures again)
  for (auto& g : ctx->graphs)
    if (g) {
      (void)hipGraphExecDestroy(g);
      g = nullptr;
    }
  // release the old block BEFORE asking for the bigger one: the allocator can then extend / reuse its address range
  // instead of placing the new block wherever a hole of that size is left (a workspace that grew after other
  // allocations was measured 1.7x slower for the GEMMs that live in it - fragmented placement)
  const uint64_t old_bytes = ctx->ws_bytes;
  if (ctx->ws) (void)hipFree(ctx->ws);
  ctx->ws = nullptr;
  ctx->ws_bytes = 0;
  void* p = nullptr;
  hipError_t e = hipMalloc(&p, bytes);
  if (e != hipSuccess) {
    (void)hipGetLastError();
    // keep the context usable at its previous size (the calls that fitted before still fit)
    void* q = nullptr;
    if (old_bytes && hipMalloc(&q, old_bytes) == hipSuccess) {
      ctx->ws = q;
      ctx->ws_bytes = old_bytes;
    }
    (void)hipSetDevice(cur);
    return catppo_fail(ctx, CATPPO_E_HIP, "catppo_reserve: hipMalloc(%llu) failed: %s", (unsigned long long)bytes,
                       hipGetErrorString(e));
  }
  ctx->ws = p;
  ctx->ws_bytes = bytes;
  (void)hipSetDevice(cur);
  return CATPPO_OK;
}

// ---------------------------------------------------------------------------------------------- hipGraphs
extern "C" int catppo_graph_begin(catppo_ctx* ctx, void* stream) {
  CATPPO_CHECK_ARG(ctx, ctx != nullptr);
  CATPPO_CHECK_ARG(ctx, stream != nullptr);      // the legacy default stream cannot be captured
  CATPPO_CHECK_ARG(ctx, !ctx->capturing);
  // relaxed mode: host threads of the process (allocator, data loaders) may keep calling the runtime
  hipError_t e = hipStreamBeginCapture(static_cast<hipStream_t>(stream), hipStreamCaptureModeRelaxed);
  if (e != hipSuccess)
    return catppo_fail(ctx, CATPPO_E_HIP, "catppo_graph_begin: hipStreamBeginCapture: %s", hipGetErrorString(e));
  ctx->capturing = true;
  return CATPPO_OK;
}

extern "C" int catppo_graph_end(catppo_ctx* ctx, void* stream, int* graph_id, int* n_nodes) {
  CATPPO_CHECK_ARG(ctx, ctx != nullptr);
  if (graph_id == nullptr) {
    // abort (ABI 0.5: catppo_graph_abort): a failure between begin and end leaves a PARTIAL capture - end it, drop the graph
    if (!ctx->capturing) return CATPPO_OK;
    ctx->capturing = false;
    hipGraph_t ga = nullptr;
    (void)hipStreamEndCapture(static_cast<hipStream_t>(stream), &ga);   // may itself fail if the capture was invalidated
    if (ga) (void)hipGraphDestroy(ga);
    (void)hipGetLastError();
    return CATPPO_OK;
  }
  CATPPO_CHECK_ARG(ctx, ctx->capturing);
  ctx->capturing = false;
  hipGraph_t g = nullptr;
  hipError_t e = hipStreamEndCapture(static_cast<hipStream_t>(stream), &g);
  if (e != hipSuccess || g == nullptr)
    return catppo_fail(ctx, CATPPO_E_HIP, "catppo_graph_end: hipStreamEndCapture: %s", hipGetErrorString(e));
  size_t nn = 0;
  (void)hipGraphGetNodes(g, nullptr, &nn);
  if (n_nodes) *n_nodes = (int)nn;
  int slot = -1;
  for (int i = 0; i < catppo_ctx::kMaxGraphs; ++i)
    if (ctx->graphs[i] == nullptr) {
      slot = i;
      break;
    }
  if (slot < 0) {
    (void)hipGraphDestroy(g);
    return catppo_fail(ctx, CATPPO_E_ARG, "catppo_graph_end: all %d graph slots are in use", catppo_ctx::kMaxGraphs);
  }
  hipGraphExec_t ex = nullptr;
  e = hipGraphInstantiate(&ex, g, nullptr, nullptr, 0);
  (void)hipGraphDestroy(g);
  if (e != hipSuccess)
    return catppo_fail(ctx, CATPPO_E_HIP, "catppo_graph_end: hipGraphInstantiate: %s", hipGetErrorString(e));
  ctx->graphs[slot] = ex;
  *graph_id = slot;
  return CATPPO_OK;
}

extern "C" int catppo_graph_launch(catppo_ctx* ctx, int graph_id, void* stream) {
  CATPPO_CHECK_ARG(ctx, ctx != nullptr);
  CATPPO_CHECK_ARG(ctx, graph_id >= 0 && graph_id < catppo_ctx::kMaxGraphs && ctx->graphs[graph_id] != nullptr);
  hipError_t e = hipGraphLaunch(ctx->graphs[graph_id], static_cast<hipStream_t>(stream));
  if (e != hipSuccess)
    return catppo_fail(ctx, CATPPO_E_HIP, "catppo_graph_launch: %s", hipGetErrorString(e));
  return CATPPO_OK;
}

extern "C" int catppo_graph_destroy(catppo_ctx* ctx, int graph_id) {
  CATPPO_CHECK_ARG(ctx, ctx != nullptr);
  CATPPO_CHECK_ARG(ctx, graph_id >= 0 && graph_id < catppo_ctx::kMaxGraphs);
  if (ctx->graphs[graph_id]) (void)hipGraphExecDestroy(ctx->graphs[graph_id]);
  ctx->graphs[graph_id] = nullptr;
  return CATPPO_OK;
}
