// libcatppo lifecycle: context, workspace, error text.
#include "common.h"

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
  *out = ctx;
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
  if (ctx->ws) {
    int cur = 0;
    (void)hipGetDevice(&cur);
    (void)hipSetDevice(ctx->device);
    (void)hipFree(ctx->ws);
    (void)hipSetDevice(cur);
  }
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
