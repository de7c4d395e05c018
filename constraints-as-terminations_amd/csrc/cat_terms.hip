// Solo12 constraint terms C1..C15 evaluated from sim-state tensors straight into the
// row-major constraint matrix cstr[N,K] that catppo_cat_step consumes.
//
// Replaces cat/constraints.py:23-235 (2-8 eager launches per term).  One block owns a tile of
// 32 envs: every term writes its columns of the tile in LDS, then the tile is flushed with one
// contiguous, coalesced store (a tile of consecutive envs is a contiguous span of cstr).
// Unfused fp32 (-ffp-contract=off); the abs/limit families are bit-identical to the torch ops,
// the norm based ones agree to 1 ulp of the norm (torch does not fix its summation order).
#include "common.h"

namespace {

constexpr int kThreads = 256;
constexpr int kRows = 16;
constexpr int kMaxBlocks = 256;   // = partial column-maximum rows handed to the CaT step
constexpr int kMaxTerms = 16;

struct TermTable {
  int n;
  int off[kMaxTerms + 1];
  catppo_term_desc d[kMaxTerms];
};

__device__ __forceinline__ float norm3(const float* p) {
  float s = p[0] * p[0];
  s = s + p[1] * p[1];
  s = s + p[2] * p[2];
  return sqrtf(s);
}

// max over history of |F[e,h,b,:]|
__device__ __forceinline__ float force_peak(const float* forces, int64_t fstride, int64_t env, int H, int B, int b) {
  const float* base = forces + env * fstride + (int64_t)b * 3;
  float m = norm3(base);
  for (int h = 1; h < H; ++h) m = nanmax(m, norm3(base + (int64_t)h * B * 3));
  return m;
}

// Work split: ONE WAVE PER TERM (terms wave, wave+4, ...): the term kind is wave-uniform (no divergence) and the
// four waves walk different terms at the same time, so the dependent chain "descriptor -> index -> state load"
// is paid once per ~n_terms/4 instead of once per term.  A block walks tiles b, b+grid, ...; besides the flushed
// tile it keeps the running column maxima of everything it produced (-> one partial row per block for the CaT
// step, which then skips its own pass over cstr).
__global__ __launch_bounds__(kThreads) void cat_terms_kernel(TermTable tab, int64_t N, const float* __restrict__ forces,
                                                             int64_t fstride, int H, int B,
                                                             const float* __restrict__ command, int cld,
                                                             float* __restrict__ cstr, int K,
                                                             float* __restrict__ colmax_partial) {
  extern __shared__ float tile[];  // [kRows*K] + [K] running column maxima
  float* cmax = tile + kRows * K;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (int c = threadIdx.x; c < K; c += kThreads) cmax[c] = -__builtin_inff();
  const int64_t n_tiles = (N + kRows - 1) / kRows;
  for (int64_t tl = blockIdx.x; tl < n_tiles; tl += gridDim.x) {
    const int64_t r0 = tl * kRows;
    const int rows = (int)((N - r0) < kRows ? (N - r0) : kRows);

    for (int t = wave; t < tab.n; t += kThreads / 64) {
      const catppo_term_desc& d = tab.d[t];
      const int W = d.width;
      const int col0 = tab.off[t];
      for (int w = lane; w < rows * W; w += 64) {
        const int e = w / W, j = w - e * W;
        const int64_t env = r0 + e;
        float out = 0.0f;
        switch (d.kind) {
          case CATPPO_TERM_ABS_LIMIT: {
            out = fabsf(d.x[env * d.x_ld + d.ids[j]]) - d.limit;
          } break;
          case CATPPO_TERM_ABS_DIFF_LIMIT: {
            const float df = d.x[env * d.x_ld + d.ids[j]] - d.y[env * d.y_ld + d.ids[j]];
            out = fabsf(df) - d.limit;
          } break;
          case CATPPO_TERM_ABS_DIFF_LIMIT_GATE_CMDY: {
            const float df = d.x[env * d.x_ld + d.ids[j]] - d.y[env * d.y_ld + d.ids[j]];
            const float c = fabsf(df) - d.limit;
            const float gate = fabsf(command[env * cld + 1]) < d.aux ? 1.0f : 0.0f;
            out = c * gate;
          } break;
          case CATPPO_TERM_GREATER: {
            out = d.x[env * d.x_ld + d.ids[0]] > d.limit ? 1.0f : 0.0f;
          } break;
          case CATPPO_TERM_CONTACT_ANY: {
            bool any = false;
            for (int b = 0; b < d.n_ids; ++b) any = any || (force_peak(forces, fstride, env, H, B, d.ids[b]) > d.limit);
            out = any ? 1.0f : 0.0f;
          } break;
          case CATPPO_TERM_NORM2_LIMIT: {
            const float a = d.x[env * d.x_ld + 0], b = d.x[env * d.x_ld + 1];
            float s = a * a;
            s = s + b * b;
            out = sqrtf(s) - d.limit;
          } break;
          case CATPPO_TERM_AIR_TIME: {
            const float gate = norm3(command + env * cld) > d.aux ? 1.0f : 0.0f;
            float c = d.limit - d.x[env * d.x_ld + d.ids[j]];
            c = c * d.y[env * d.y_ld + d.ids[j]];
            out = c * gate;
          } break;
          case CATPPO_TERM_N_FOOT_CONTACT: {
            int n = 0;
            for (int b = 0; b < d.n_ids; ++b) n += force_peak(forces, fstride, env, H, B, d.ids[b]) > 1.0f ? 1 : 0;
            int diff = n - (int)d.limit;
            diff = diff < 0 ? -diff : diff;
            const float gate = norm3(command + env * cld) > d.aux ? 1.0f : 0.0f;
            out = (float)diff * gate;
          } break;
          case CATPPO_TERM_ACTION_RATE: {
            const float df = fabsf(d.x[env * d.x_ld + d.ids[j]] - d.y[env * d.y_ld + d.ids[j]]);
            out = df / d.aux - d.limit;
          } break;
          case CATPPO_TERM_FORCE_LIMIT: {
            out = force_peak(forces, fstride, env, H, B, d.ids[j]) - d.limit;
          } break;
          case CATPPO_TERM_LIMIT_MINUS: {
            out = d.limit - d.x[env * d.x_ld + d.ids[0]];
          } break;
          case CATPPO_TERM_ABS_LIMIT_GATE_CMDNORM_LT: {
            const float c = fabsf(d.x[env * d.x_ld + d.ids[j]]) - d.limit;
            const float gate = norm3(command + env * cld) < d.aux ? 1.0f : 0.0f;
            out = c * gate;
          } break;
          default:
            break;
        }
        tile[e * K + col0 + j] = out;
      }
    }
    __syncthreads();
    float* dst = cstr + r0 * K;
    for (int e = threadIdx.x; e < rows * K; e += kThreads) dst[e] = tile[e];
    if (colmax_partial != nullptr) {
      for (int c = threadIdx.x; c < K; c += kThreads) {
        float m = cmax[c];
        for (int r = 0; r < rows; ++r) m = nanmax(m, tile[r * K + c]);
        cmax[c] = m;
      }
    }
    __syncthreads();
  }
  if (colmax_partial != nullptr)
    for (int c = threadIdx.x; c < K; c += kThreads) colmax_partial[(int64_t)blockIdx.x * K + c] = cmax[c];
}

}  // namespace

// shared by catppo_cat_terms and catppo_cat_terms_step (cat_step.hip); *nblk_out = partial rows written
int catppo_internal_launch_terms(catppo_ctx* ctx, const catppo_term_desc* desc, int n_terms, int64_t N,
                                 const float* forces, int64_t forces_env_stride, int H, int B, const float* command,
                                 int command_ld, float* cstr, int K, float* colmax_partial, int* nblk_out,
                                 hipStream_t stream) {
  CATPPO_CHECK_ARG(ctx, ctx != nullptr);
  CATPPO_CHECK_ARG(ctx, desc && cstr && N >= 1 && n_terms >= 1 && n_terms <= kMaxTerms);
  TermTable tab;
  tab.n = n_terms;
  int off = 0;
  for (int t = 0; t < n_terms; ++t) {
    const catppo_term_desc& d = desc[t];
    CATPPO_CHECK_ARG(ctx, d.width >= 1 && d.n_ids >= 0 && d.n_ids <= 16);
    const bool needs_forces = d.kind == CATPPO_TERM_CONTACT_ANY || d.kind == CATPPO_TERM_N_FOOT_CONTACT ||
                              d.kind == CATPPO_TERM_FORCE_LIMIT;
    const bool needs_cmd = d.kind == CATPPO_TERM_ABS_DIFF_LIMIT_GATE_CMDY || d.kind == CATPPO_TERM_AIR_TIME ||
                           d.kind == CATPPO_TERM_N_FOOT_CONTACT || d.kind == CATPPO_TERM_ABS_LIMIT_GATE_CMDNORM_LT;
    const bool needs_y = d.kind == CATPPO_TERM_ABS_DIFF_LIMIT || d.kind == CATPPO_TERM_ABS_DIFF_LIMIT_GATE_CMDY ||
                         d.kind == CATPPO_TERM_AIR_TIME || d.kind == CATPPO_TERM_ACTION_RATE;
    CATPPO_CHECK_ARG(ctx, !needs_forces || (forces != nullptr && H >= 1 && B >= 1 &&
                                            forces_env_stride >= (int64_t)H * B * 3));
    CATPPO_CHECK_ARG(ctx, !needs_cmd || (command != nullptr && command_ld >= 3));
    CATPPO_CHECK_ARG(ctx, needs_forces || d.x != nullptr);
    CATPPO_CHECK_ARG(ctx, !needs_y || d.y != nullptr);
    const bool per_id = d.kind != CATPPO_TERM_GREATER && d.kind != CATPPO_TERM_CONTACT_ANY &&
                        d.kind != CATPPO_TERM_NORM2_LIMIT && d.kind != CATPPO_TERM_N_FOOT_CONTACT &&
                        d.kind != CATPPO_TERM_LIMIT_MINUS;
    CATPPO_CHECK_ARG(ctx, per_id ? d.width == d.n_ids : d.width == 1);
    tab.off[t] = off;
    tab.d[t] = d;
    off += d.width;
  }
  tab.off[n_terms] = off;
  CATPPO_CHECK_ARG(ctx, off == K);
  const size_t lds = sizeof(float) * ((size_t)kRows * K + K);
  CATPPO_CHECK_ARG(ctx, lds <= 150 * 1024);
  int64_t nblk = cdiv64(N, kRows);
  if (nblk > kMaxBlocks) nblk = kMaxBlocks;
  hipLaunchKernelGGL(cat_terms_kernel, dim3((unsigned)nblk), dim3(kThreads), lds, stream, tab, N, forces,
                     forces_env_stride, H, B, command, command_ld, cstr, K, colmax_partial);
  CATPPO_CHECK_LAUNCH(ctx);
  if (nblk_out) *nblk_out = (int)nblk;
  return CATPPO_OK;
}

extern "C" int catppo_cat_terms(catppo_ctx* ctx, const catppo_term_desc* desc, int n_terms, int64_t N,
                                const float* forces, int64_t forces_env_stride, int H, int B,
                                const float* command, int command_ld, float* cstr, int K, void* stream) {
  return catppo_internal_launch_terms(ctx, desc, n_terms, N, forces, forces_env_stride, H, B, command, command_ld,
                                      cstr, K, nullptr, nullptr, static_cast<hipStream_t>(stream));
}
