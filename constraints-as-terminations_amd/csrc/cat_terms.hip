// Solo12 constraint terms C1..C15 evaluated from sim-state tensors straight into the
// row-major constraint matrix cstr[N,K] that catppo_cat_step consumes.
//
// Replaces cat/constraints.py:23-235 (2-8 eager launches per term).  One block owns a tile of
// 16 envs: every term writes its columns of the tile in LDS, then the tile is flushed with one
// contiguous, coalesced store (a tile of consecutive envs is a contiguous span of cstr).
// Unfused fp32 (-ffp-contract=off); the abs/limit families are bit-identical to the torch ops,
// the norm based ones agree to 1 ulp of the norm (torch does not fix its summation order).
#include "terms_eval.h"

namespace {

using namespace terms;
constexpr int kThreads = 256;

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
        const float out = eval_term(d, d.ids, env, j, forces, fstride, H, B, command, cld);
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
  if (const char* why = build_table(desc, n_terms, forces, forces_env_stride, H, B, command, command_ld, K, &tab))
    return catppo_fail(ctx, CATPPO_E_ARG, "catppo_cat_terms: %s", why);
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
