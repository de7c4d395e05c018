// CaT step: constraint matrix -> running-max EMA -> per-column termination probability ->
// per-env max, per-term episode statistics, reward scaling and float dones.
//
// Replaces the reference's per-term eager op chains (cat/constraint_manager.py:39-82,213-229
// and cat/cat_env.py:102-107,118-121; ~15 launches + one host sync per term per env step)
// with three launches and no host sync:
//   colmax_partial : per-block column maxima over a slab of envs (coalesced row reads)
//   reduce_ema     : fold the partials (order independent => bit-exact), floor at 1e-6,
//                    EMA update of the running maxima in the reference's unfused fp32 order
//   finish         : probabilities for a tile of envs staged in LDS, per-term / per-env max,
//                    statistics RMW, reward/dones epilogue
// The file is compiled with -ffp-contract=off: every product and sum below is rounded
// separately, exactly like the separate torch kernels of the reference.
#include "common.h"

namespace {

constexpr int kThreads = 256;
constexpr int kFinishRows = 32;   // envs per finish block
constexpr int kMaxTerms = 64;

// per-term metadata travels by value in the kernel arguments: the curriculum rewrites max_p on
// the host at every reset, so there is no device copy to keep coherent
struct TermMeta {
  int32_t off[kMaxTerms + 1];
  float dp[kMaxTerms];
};

// ---------------------------------------------------------------------------------------------
// per-block column maxima.  thread (c, g): column c, row group g; consecutive lanes read
// consecutive columns of one row => each wave touches ~1 contiguous row segment per load.
__global__ __launch_bounds__(kThreads) void cat_colmax_partial(const float* __restrict__ cstr, int64_t N,
                                                               int K, int rows_per_block,
                                                               float* __restrict__ partial) {
  extern __shared__ float lds[];
  const int Kc = K < kThreads ? K : kThreads;
  const int G = kThreads / Kc;
  const int c0 = threadIdx.x % Kc;
  const int g = threadIdx.x / Kc;
  for (int cb = 0; cb < K; cb += Kc) {   // uniform trip count: barriers inside
    const int c = cb + c0;
    const bool active = g < G && c < K;
    float m = -__builtin_inff();
    if (active) {
      for (int64_t r0 = (int64_t)blockIdx.x * rows_per_block; r0 < N; r0 += (int64_t)gridDim.x * rows_per_block) {
        int64_t r1 = r0 + rows_per_block < N ? r0 + rows_per_block : N;
        for (int64_t r = r0 + g; r < r1; r += G) m = nanmax(m, cstr[r * K + c]);
      }
      lds[g * Kc + c0] = m;
    }
    __syncthreads();
    if (g == 0 && c < K) {
      for (int gg = 1; gg < G; ++gg) m = nanmax(m, lds[gg * Kc + c0]);
      partial[(int64_t)blockIdx.x * K + c] = m;
    }
    __syncthreads();
  }
}

// fold [nblk,K] partial maxima (or a single row = an already reduced / all-reduced colmax),
// floor at 1e-6 (:55), optionally publish the column maxima, optionally EMA-update rm (:58-61).
// Thread (c,g): column c, partial-row group g; max is order independent => bit-exact.
__global__ __launch_bounds__(kThreads) void cat_reduce_ema(const float* __restrict__ partial, int nblk, int K,
                                                           float* __restrict__ colmax_out, float* __restrict__ rm,
                                                           int do_ema, int first_call, float tau,
                                                           float one_minus_tau) {
  __shared__ float sm[kThreads];
  const int Kc = K < kThreads ? K : kThreads;
  const int G = kThreads / Kc;
  const int c0 = threadIdx.x % Kc, g = threadIdx.x / Kc;
  for (int cb = 0; cb < K; cb += Kc) {
    const int c = cb + c0;
    float m = -__builtin_inff();
    if (g < G && c < K) {
#pragma unroll 8
      for (int b = g; b < nblk; b += G) m = nanmax(m, partial[(int64_t)b * K + c]);
    }
    sm[threadIdx.x] = m;
    __syncthreads();
    if (g == 0 && c < K) {
      for (int gg = 1; gg < G; ++gg) m = nanmax(m, sm[gg * Kc + c0]);
      m = (m < 1e-6f) ? 1e-6f : m;  // clamp(min=1e-6); NaN stays NaN like torch
      if (colmax_out) colmax_out[c] = m;
      if (do_ema) {
        float r;
        if (first_call) {
          r = m;
        } else {
          float a = rm[c] * tau;          // rm.mul_(tau)
          float b = one_minus_tau * m;    // (1-tau) * cmax
          r = a + b;                      // .add_()
        }
        rm[c] = r;
      }
    }
    __syncthreads();
  }
}

// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(kThreads) void cat_finish(
    const float* __restrict__ cstr, int64_t N, int K, const TermMeta meta, int n_terms, float min_p,
    const float* __restrict__ rm, float* __restrict__ reward,
    const uint8_t* __restrict__ reset_mask, float* __restrict__ cstr_prob, float* __restrict__ dones,
    float* __restrict__ ep_viol, float* __restrict__ ep_prob, float* __restrict__ probs) {
  extern __shared__ float lds[];
  float* col_rm = lds;                               // [K]
  float* col_dp = col_rm + K;                        // [K]
  float* tile = col_dp + K;                          // [kFinishRows*K]
  float* tmax = tile + kFinishRows * K;              // [n_terms*kFinishRows]
  __shared__ int s_off[kMaxTerms + 1];

  if (threadIdx.x <= n_terms) s_off[threadIdx.x] = meta.off[threadIdx.x];
  __syncthreads();
  for (int c = threadIdx.x; c < K; c += kThreads) {
    int t = 0;
    while (t + 1 < n_terms && c >= s_off[t + 1]) ++t;
    col_rm[c] = rm[c];
    col_dp[c] = meta.dp[t];
  }
  __syncthreads();

  const int64_t r0 = (int64_t)blockIdx.x * kFinishRows;
  const int rows = (int)((N - r0) < kFinishRows ? (N - r0) : kFinishRows);
  const int n_el = rows * K;
  const float* src = cstr + r0 * K;
  float* pdst = probs ? probs + r0 * K : nullptr;
  // the tile of `rows` consecutive envs is one contiguous span of the row-major matrix
  for (int e = threadIdx.x; e < n_el; e += kThreads) {
    const int c = e % K;
    const float x = src[e];
    float p = 0.0f;
    if (x > 0.0f) {
      float q = x / col_rm[c];                       // normalized = constraint / running_max
      q = q < 0.0f ? 0.0f : (q > 1.0f ? 1.0f : q);   // clamp(0,1)
      const float s = q * col_dp[c];                 // * (max_p - min_p)
      p = min_p + s;                                 // min_p + ...
    }
    tile[e] = p;
    if (pdst) pdst[e] = p;
  }
  __syncthreads();

  // per (term, env) max; consecutive threads -> consecutive envs of one term (coalesced RMW)
  for (int w = threadIdx.x; w < n_terms * rows; w += kThreads) {
    const int t = w / rows, e = w - t * rows;
    const float* row = tile + e * K;
    float m = row[s_off[t]];
    for (int c = s_off[t] + 1; c < s_off[t + 1]; ++c) m = nanmax(m, row[c]);
    tmax[t * kFinishRows + e] = m;
    const int64_t gi = (int64_t)t * N + r0 + e;
    ep_viol[gi] = ep_viol[gi] + (m > 0.0f ? 1.0f : 0.0f);
    ep_prob[gi] = ep_prob[gi] + m;
  }
  __syncthreads();

  if (threadIdx.x < rows) {
    const int e = threadIdx.x;
    float p = tmax[e];
    for (int t = 1; t < n_terms; ++t) p = nanmax(p, tmax[t * kFinishRows + e]);
    const int64_t i = r0 + e;
    cstr_prob[i] = p;
    if (reward) {
      const float omp = 1.0f - p;
      const float r = reward[i] * omp;
      reward[i] = (r < 0.0f) ? 0.0f : r;             // clip(min=0)
    }
    if (dones) dones[i] = (reset_mask && reset_mask[i]) ? 1.0f : p;
  }
}

// one 1024-thread block per term: masked means of sums/len (fp64 accumulation, fixed order), then zero
// the rows.  N = 4096 is four elements per thread: the kernel is a single round of loads.
constexpr int kResetThreads = 1024;
__global__ __launch_bounds__(kResetThreads) void cat_reset_stats(float* __restrict__ ep_viol, float* __restrict__ ep_prob,
                                                                 const int64_t* __restrict__ ep_len,
                                                                 const uint8_t* __restrict__ mask, int64_t N,
                                                                 const float* __restrict__ prev, float* __restrict__ out) {
  __shared__ double s_a[kResetThreads / 64], s_b[kResetThreads / 64], s_n[kResetThreads / 64];
  const int t = blockIdx.x;
  float* v = ep_viol + (int64_t)t * N;
  float* p = ep_prob + (int64_t)t * N;
  double a = 0.0, b = 0.0, n = 0.0;
  for (int64_t i = threadIdx.x; i < N; i += kResetThreads) {
    if (mask == nullptr || mask[i]) {
      const float L = (float)ep_len[i];
      a += (double)(v[i] / L);
      b += (double)(p[i] / L);
      n += 1.0;
      v[i] = 0.0f;
      p[i] = 0.0f;
    }
  }
#pragma unroll
  for (int m = 32; m >= 1; m >>= 1) {
    a += __shfl_xor(a, m, 64);
    b += __shfl_xor(b, m, 64);
    n += __shfl_xor(n, m, 64);
  }
  const int wave = threadIdx.x >> 6;
  if ((threadIdx.x & 63) == 0) s_a[wave] = a, s_b[wave] = b, s_n[wave] = n;
  __syncthreads();
  if (threadIdx.x == 0) {
    for (int w = 1; w < kResetThreads / 64; ++w) a += s_a[w], b += s_b[w], n += s_n[w];
    if (n > 0.0) {
      out[2 * t] = (float)(a / n) * 100.0f;
      out[2 * t + 1] = (float)(b / n);
    } else if (prev != nullptr) {   // nobody reset: keep the previous log values (reference keeps extras["log"])
      out[2 * t] = prev[2 * t];
      out[2 * t + 1] = prev[2 * t + 1];
    }
  }
}

int launch_colmax(catppo_ctx* ctx, const float* cstr, int64_t N, int K, float** partial_out, int* nblk_out,
                  hipStream_t s) {
  const int Kc = K < kThreads ? K : kThreads;
  const int G = kThreads / Kc;
  const int rows_per_block = G * 8;
  int nblk = (int)cdiv64(N, rows_per_block);
  if (nblk > 128) nblk = 128;   // the fold kernel walks the partial rows: keep them few
  WsCarver ws(ctx);
  float* partial = ws.take<float>((uint64_t)nblk * K);
  CATPPO_NEED_WS(ctx, partial);
  hipLaunchKernelGGL(cat_colmax_partial, dim3(nblk), dim3(kThreads), sizeof(float) * kThreads, s, cstr, N, K,
                     rows_per_block, partial);
  CATPPO_CHECK_LAUNCH(ctx);
  *partial_out = partial;
  *nblk_out = nblk;
  return CATPPO_OK;
}

int launch_finish(catppo_ctx* ctx, const float* cstr, int64_t N, int K, const int32_t* term_off, int n_terms,
                  const float* term_dp, float min_p, const float* rm, float* reward, const uint8_t* reset_mask,
                  float* cstr_prob, float* dones, float* ep_viol, float* ep_prob, float* probs, hipStream_t s) {
  const size_t lds = sizeof(float) * ((size_t)2 * K + (size_t)kFinishRows * K + (size_t)n_terms * kFinishRows);
  if (lds > 150 * 1024) return catppo_fail(ctx, CATPPO_E_ARG, "cat_finish: K=%d too wide for one LDS tile", K);
  const int nblk = (int)cdiv64(N, kFinishRows);
  TermMeta meta;
  int prev = 0;
  for (int t = 0; t <= n_terms; ++t) {
    if (term_off[t] < prev || term_off[t] > K) return catppo_fail(ctx, CATPPO_E_ARG, "cat: term_off not monotone");
    prev = meta.off[t] = term_off[t];
  }
  if (term_off[0] != 0 || term_off[n_terms] != K) return catppo_fail(ctx, CATPPO_E_ARG, "cat: term_off must span [0,K]");
  for (int t = 0; t < n_terms; ++t) meta.dp[t] = term_dp[t];
  hipLaunchKernelGGL(cat_finish, dim3(nblk), dim3(kThreads), lds, s, cstr, N, K, meta, n_terms, min_p, rm, reward,
                     reset_mask, cstr_prob, dones, ep_viol, ep_prob, probs);
  CATPPO_CHECK_LAUNCH(ctx);
  return CATPPO_OK;
}

int check_common(catppo_ctx* ctx, const float* cstr, int64_t N, int K, int n_terms) {
  CATPPO_CHECK_ARG(ctx, ctx != nullptr);
  CATPPO_CHECK_ARG(ctx, cstr != nullptr);
  CATPPO_CHECK_ARG(ctx, N > 0 && N < (int64_t(1) << 31) / (K > 0 ? K : 1));
  CATPPO_CHECK_ARG(ctx, K > 0 && K <= 4096);
  CATPPO_CHECK_ARG(ctx, n_terms > 0 && n_terms <= kMaxTerms && n_terms <= K);
  return CATPPO_OK;
}

}  // namespace

extern "C" int catppo_cat_colmax(catppo_ctx* ctx, const float* cstr, int64_t N, int K, float* colmax,
                                 void* stream) {
  if (int rc = check_common(ctx, cstr, N, K, 1)) return rc;
  CATPPO_CHECK_ARG(ctx, colmax != nullptr);
  hipStream_t s = static_cast<hipStream_t>(stream);
  float* partial;
  int nblk;
  if (int rc = launch_colmax(ctx, cstr, N, K, &partial, &nblk, s)) return rc;
  hipLaunchKernelGGL(cat_reduce_ema, dim3(1), dim3(kThreads), 0, s, partial, nblk, K, colmax, (float*)nullptr, 0, 0,
                     0.0f, 0.0f);
  CATPPO_CHECK_LAUNCH(ctx);
  return CATPPO_OK;
}

extern "C" int catppo_cat_terms_colmax(catppo_ctx* ctx, const catppo_term_desc* desc, int n_terms, int64_t N,
                                       const float* forces, int64_t forces_env_stride, int H, int B,
                                       const float* command, int command_ld, float* cstr, int K, float* colmax,
                                       void* stream) {
  if (int rc = check_common(ctx, cstr, N, K, n_terms)) return rc;
  CATPPO_CHECK_ARG(ctx, desc && colmax);
  hipStream_t s = static_cast<hipStream_t>(stream);
  WsCarver ws(ctx);
  float* partial = ws.take<float>((uint64_t)256 * K);
  CATPPO_NEED_WS(ctx, partial);
  int nblk = 0;
  if (int rc = catppo_internal_launch_terms(ctx, desc, n_terms, N, forces, forces_env_stride, H, B, command,
                                            command_ld, cstr, K, partial, &nblk, s))
    return rc;
  hipLaunchKernelGGL(cat_reduce_ema, dim3(1), dim3(kThreads), 0, s, partial, nblk, K, colmax, (float*)nullptr, 0, 0,
                     0.0f, 0.0f);
  CATPPO_CHECK_LAUNCH(ctx);
  return CATPPO_OK;
}

extern "C" int catppo_cat_apply(catppo_ctx* ctx, const float* cstr, int64_t N, int K, const int32_t* term_off,
                                int n_terms, const float* term_dp, float min_p, float tau, float one_minus_tau,
                                int first_call, const float* colmax, float* rm, float* reward,
                                const uint8_t* reset_mask, float* cstr_prob, float* dones, float* ep_viol,
                                float* ep_prob, float* probs, void* stream) {
  if (int rc = check_common(ctx, cstr, N, K, n_terms)) return rc;
  CATPPO_CHECK_ARG(ctx, term_off && term_dp && colmax && rm && cstr_prob && ep_viol && ep_prob);
  hipStream_t s = static_cast<hipStream_t>(stream);
  hipLaunchKernelGGL(cat_reduce_ema, dim3(1), dim3(kThreads), 0, s, colmax, 1, K, (float*)nullptr, rm, 1,
                     first_call, tau, one_minus_tau);
  CATPPO_CHECK_LAUNCH(ctx);
  return launch_finish(ctx, cstr, N, K, term_off, n_terms, term_dp, min_p, rm, reward, reset_mask, cstr_prob,
                       dones, ep_viol, ep_prob, probs, s);
}

extern "C" int catppo_cat_step(catppo_ctx* ctx, const float* cstr, int64_t N, int K, const int32_t* term_off,
                               int n_terms, const float* term_dp, float min_p, float tau, float one_minus_tau,
                               int first_call, float* rm, float* reward, const uint8_t* reset_mask,
                               float* cstr_prob, float* dones, float* ep_viol, float* ep_prob, float* probs,
                               void* stream) {
  if (int rc = check_common(ctx, cstr, N, K, n_terms)) return rc;
  CATPPO_CHECK_ARG(ctx, term_off && term_dp && rm && cstr_prob && ep_viol && ep_prob);
  hipStream_t s = static_cast<hipStream_t>(stream);
  float* partial;
  int nblk;
  if (int rc = launch_colmax(ctx, cstr, N, K, &partial, &nblk, s)) return rc;
  hipLaunchKernelGGL(cat_reduce_ema, dim3(1), dim3(kThreads), 0, s, partial, nblk, K, (float*)nullptr, rm, 1,
                     first_call, tau, one_minus_tau);
  CATPPO_CHECK_LAUNCH(ctx);
  return launch_finish(ctx, cstr, N, K, term_off, n_terms, term_dp, min_p, rm, reward, reset_mask, cstr_prob,
                       dones, ep_viol, ep_prob, probs, s);
}

extern "C" int catppo_cat_terms_step(catppo_ctx* ctx, const catppo_term_desc* desc, int n_terms, int64_t N,
                                     const float* forces, int64_t forces_env_stride, int H, int B,
                                     const float* command, int command_ld, float* cstr, int K,
                                     const int32_t* term_off, const float* term_dp, float min_p, float tau,
                                     float one_minus_tau, int first_call, float* rm, float* reward,
                                     const uint8_t* reset_mask, float* cstr_prob, float* dones, float* ep_viol,
                                     float* ep_prob, float* probs, void* stream) {
  if (int rc = check_common(ctx, cstr, N, K, n_terms)) return rc;
  CATPPO_CHECK_ARG(ctx, desc && term_off && term_dp && rm && cstr_prob && ep_viol && ep_prob);
  hipStream_t s = static_cast<hipStream_t>(stream);
  WsCarver ws(ctx);
  float* partial = ws.take<float>((uint64_t)256 * K);
  CATPPO_NEED_WS(ctx, partial);
  int nblk = 0;
  if (int rc = catppo_internal_launch_terms(ctx, desc, n_terms, N, forces, forces_env_stride, H, B, command,
                                            command_ld, cstr, K, partial, &nblk, s))
    return rc;
  hipLaunchKernelGGL(cat_reduce_ema, dim3(1), dim3(kThreads), 0, s, partial, nblk, K, (float*)nullptr, rm, 1,
                     first_call, tau, one_minus_tau);
  CATPPO_CHECK_LAUNCH(ctx);
  return launch_finish(ctx, cstr, N, K, term_off, n_terms, term_dp, min_p, rm, reward, reset_mask, cstr_prob,
                       dones, ep_viol, ep_prob, probs, s);
}

extern "C" int catppo_cat_reset(catppo_ctx* ctx, float* ep_viol, float* ep_prob, const int64_t* episode_length,
                                const uint8_t* mask, int n_terms, int64_t N, const float* prev, float* out,
                                void* stream) {
  CATPPO_CHECK_ARG(ctx, ctx != nullptr);
  CATPPO_CHECK_ARG(ctx, ep_viol && ep_prob && episode_length && out && n_terms >= 1 && N >= 1);
  hipLaunchKernelGGL(cat_reset_stats, dim3(n_terms), dim3(kResetThreads), 0, static_cast<hipStream_t>(stream), ep_viol,
                     ep_prob, episode_length, mask, N, prev, out);
  CATPPO_CHECK_LAUNCH(ctx);
  return CATPPO_OK;
}
