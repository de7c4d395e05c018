// libcatppo lifecycle: context, workspace, error text.
#include "common.h"

#include <cstdlib>

extern "C" int catppo_version(void) { return CATPPO_VERSION; }

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
  if (ctx->ws) (void)hipFree(ctx->ws);
  if (ctx->side) (void)hipStreamDestroy(ctx->side);
  for (auto& e : ctx->ev_fork)
    if (e) (void)hipEventDestroy(e);
  if (ctx->ev_join) (void)hipEventDestroy(ctx->ev_join);
  (void)hipSetDevice(cur);
  delete ctx;
}

extern "C" const char* catppo_last_error(catppo_ctx* ctx) { return ctx ? ctx->err : "null ctx"; }

extern "C" int catppo_reserve(catppo_ctx* ctx, uint64_t bytes) {
  if (!ctx) return CATPPO_E_ARG;
  if (bytes <= ctx->ws_bytes) return CATPPO_OK;
  int cur = 0;
  (void)hipGetDevice(&cur);
  (void)hipSetDevice(ctx->device);
  // the old block may still be referenced by enqueued kernels: drain before freeing
  (void)hipDeviceSynchronize();
  void* p = nullptr;
  hipError_t e = hipMalloc(&p, bytes);
  if (e != hipSuccess) {
    (void)hipSetDevice(cur);
    return catppo_fail(ctx, CATPPO_E_HIP, "catppo_reserve: hipMalloc(%llu) failed: %s", (unsigned long long)bytes,
                       hipGetErrorString(e));
  }
  if (ctx->ws) (void)hipFree(ctx->ws);
  ctx->ws = p;
  ctx->ws_bytes = bytes;
  (void)hipSetDevice(cur);
  return CATPPO_OK;
}
