// Per-env bookkeeping around the CaT step, fused into single launches.
//   env_pre_step   : counters, terminations and the raw reward of one env step
//                    (reference cat/cat_env.py:62,92-97: process_action, episode_length_buf += 1,
//                    termination_manager.compute(): time_outs / terminated / reset_buf; plus the
//                    reward_manager output copied into reward_buf)   - replaces 7 tiny torch launches
//   rollout_store  : rewards[step] / dones[step+1] / true_dones[step+1] of the PPO rollout buffer
//                    (reference cleanrl/ppo.py:215-216,226)            - replaces 3 tiny torch launches
#include "common.h"
#include "rng.h"

namespace {

__global__ __launch_bounds__(256) void env_pre_step_kernel(const float* __restrict__ action_in, float* __restrict__ action,
                                                           float* __restrict__ prev_action, int A,
                                                           int64_t* __restrict__ episode_length, int64_t max_len,
                                                           const float* __restrict__ hard_reset, int64_t hr_stride,
                                                           const float* __restrict__ reward_src, int64_t rw_stride,
                                                           uint8_t* __restrict__ time_outs, uint8_t* __restrict__ terminated,
                                                           uint8_t* __restrict__ reset, float* __restrict__ reward_out,
                                                           int64_t N) {
  const int64_t tid = (int64_t)blockIdx.x * 256 + threadIdx.x;
  // action manager: prev <- current, current <- new   (N*A elements)
  for (int64_t e = tid; e < N * A; e += (int64_t)gridDim.x * 256) {
    prev_action[e] = action[e];
    action[e] = action_in[e];
  }
  for (int64_t i = tid; i < N; i += (int64_t)gridDim.x * 256) {
    const int64_t len = episode_length[i] + 1;          // episode_length_buf += 1
    episode_length[i] = len;
    const bool to = len >= max_len;                      // time_out termination term
    const bool term = hard_reset[i * hr_stride] > 0.5f;  // simulator-side hard terminations
    time_outs[i] = to;
    terminated[i] = term;
    reset[i] = to || term;
    reward_out[i] = reward_src[i * rw_stride];
  }
}

template <typename ST>
__global__ __launch_bounds__(256) void rollout_store_kernel(const float* __restrict__ reward, const float* __restrict__ dones,
                                                            const uint8_t* __restrict__ time_outs,
                                                            ST* __restrict__ rewards_t, ST* __restrict__ dones_t1,
                                                            ST* __restrict__ true_dones_t1, int64_t N) {
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < N; i += (int64_t)gridDim.x * 256) {
    rewards_t[i] = (ST)reward[i];          // fp16 planes: round to nearest even
    dones_t1[i] = (ST)dones[i];
    true_dones_t1[i] = time_outs[i] ? (ST)1.0f : (ST)0.0f;
  }
}

}  // namespace

extern "C" int catppo_env_pre_step(catppo_ctx* ctx, const float* action_in, float* action, float* prev_action, int A,
                                   int64_t* episode_length, int64_t max_episode_length, const float* hard_reset,
                                   int64_t hard_reset_stride, const float* reward_src, int64_t reward_stride,
                                   uint8_t* time_outs, uint8_t* terminated, uint8_t* reset, float* reward_out,
                                   int64_t N, void* stream) {
  CATPPO_CHECK_ARG(ctx, ctx != nullptr);
  CATPPO_CHECK_ARG(ctx, action_in && action && prev_action && A >= 1 && episode_length && hard_reset && reward_src);
  CATPPO_CHECK_ARG(ctx, time_outs && terminated && reset && reward_out && N >= 1);
  int64_t nblk = cdiv64(N * A, 256);
  if (nblk > 1024) nblk = 1024;
  hipLaunchKernelGGL(env_pre_step_kernel, dim3((unsigned)nblk), dim3(256), 0, static_cast<hipStream_t>(stream), action_in,
                     action, prev_action, A, episode_length, max_episode_length, hard_reset, hard_reset_stride,
                     reward_src, reward_stride, time_outs, terminated, reset, reward_out, N);
  CATPPO_CHECK_LAUNCH(ctx);
  return CATPPO_OK;
}

static int rollout_store_f32(catppo_ctx* ctx, const float* reward, const float* dones, const uint8_t* time_outs,
                                    float* rewards_t, float* dones_t1, float* true_dones_t1, int64_t N, void* stream) {
  CATPPO_CHECK_ARG(ctx, ctx != nullptr);
  CATPPO_CHECK_ARG(ctx, reward && dones && time_outs && rewards_t && dones_t1 && true_dones_t1 && N >= 1);
  int64_t nblk = cdiv64(N, 256);
  if (nblk > 1024) nblk = 1024;
  hipLaunchKernelGGL(rollout_store_kernel<float>, dim3((unsigned)nblk), dim3(256), 0, static_cast<hipStream_t>(stream),
                     reward, dones, time_outs, rewards_t, dones_t1, true_dones_t1, N);
  CATPPO_CHECK_LAUNCH(ctx);
  return CATPPO_OK;
}

extern "C" int catppo_rollout_store_ex(catppo_ctx* ctx, const float* reward, const float* dones,
                                       const uint8_t* time_outs, void* rewards_t, void* dones_t1, void* true_dones_t1,
                                       int dtype, int64_t N, void* stream) {
  if (dtype == CATPPO_F32)
    return rollout_store_f32(ctx, reward, dones, time_outs, static_cast<float*>(rewards_t),
                                static_cast<float*>(dones_t1), static_cast<float*>(true_dones_t1), N, stream);
  CATPPO_CHECK_ARG(ctx, ctx != nullptr);
  CATPPO_CHECK_ARG(ctx, dtype == CATPPO_F16);
  CATPPO_CHECK_ARG(ctx, reward && dones && time_outs && rewards_t && dones_t1 && true_dones_t1 && N >= 1);
  int64_t nblk = cdiv64(N, 256);
  if (nblk > 1024) nblk = 1024;
  using h = _Float16;
  hipLaunchKernelGGL(rollout_store_kernel<h>, dim3((unsigned)nblk), dim3(256), 0, static_cast<hipStream_t>(stream),
                     reward, dones, time_outs, static_cast<h*>(rewards_t), static_cast<h*>(dones_t1),
                     static_cast<h*>(true_dones_t1), N);
  CATPPO_CHECK_LAUNCH(ctx);
  return CATPPO_OK;
}

// ---------------------------------------------------------------------------------------------
// Advantage moments of EVERY minibatch of an iteration in one launch (env-sharded exact mode):
// out[m] = { sum_i adv[inds[m*mb + i]], sum_i adv[...]^2 } in fp64, fixed order.  The ranks then
// exchange all minibatches' moments with ONE all-reduce per iteration instead of one per minibatch.
namespace {
__global__ __launch_bounds__(256) void adv_moments_kernel(const float* __restrict__ adv, const int64_t* __restrict__ inds,
                                                          int64_t total, int64_t mb, double* __restrict__ out) {
  __shared__ double s1[256], s2[256];
  const int64_t lo = (int64_t)blockIdx.x * mb;
  const int64_t hi = lo + mb < total ? lo + mb : total;
  double a = 0.0, b = 0.0;
  for (int64_t i = lo + threadIdx.x; i < hi; i += 256) {
    const double v = (double)adv[inds[i]];
    a += v;
    b += v * v;
  }
  s1[threadIdx.x] = a, s2[threadIdx.x] = b;
  __syncthreads();
  for (int w = 128; w >= 1; w >>= 1) {
    if (threadIdx.x < w) {
      s1[threadIdx.x] += s1[threadIdx.x + w];
      s2[threadIdx.x] += s2[threadIdx.x + w];
    }
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    out[3 * blockIdx.x] = s1[0];
    out[3 * blockIdx.x + 1] = s2[0];
    out[3 * blockIdx.x + 2] = (double)(hi - lo);
  }
}

// the same moments from the 64-row chunk sums the epoch gather wrote: one wave per minibatch, lane-strided partial
// sums in fixed order, then a butterfly (order depends on the chunk count only => run-to-run reproducible)
__global__ __launch_bounds__(64) void adv_moments_parts_kernel(const double* __restrict__ parts, int parts_per_mb,
                                                               int64_t total, int64_t mb, double* __restrict__ out) {
  const double* p = parts + 2 * (int64_t)blockIdx.x * parts_per_mb;
  double a = 0.0, b = 0.0;
  for (int c = threadIdx.x; c < parts_per_mb; c += 64) a += p[2 * c], b += p[2 * c + 1];
#pragma unroll
  for (int m = 32; m >= 1; m >>= 1) a += __shfl_xor(a, m, 64), b += __shfl_xor(b, m, 64);
  if (threadIdx.x == 0) {
    const int64_t lo = (int64_t)blockIdx.x * mb, hi = lo + mb < total ? lo + mb : total;
    out[3 * blockIdx.x] = a;
    out[3 * blockIdx.x + 1] = b;
    out[3 * blockIdx.x + 2] = (double)(hi - lo);
  }
}

// Round 5: the chunk sums of EVERY epoch of an iteration without the gather (grid = chunks x minibatches x epochs): the
// keyed permutation of epoch e is a function of (seed, iteration, e) alone (rng.h), so the advantage moments of all
// E x n_mb minibatches can be formed - and exchanged between ranks in ONE all-reduce - before the first epoch starts.
// Per 64-row chunk the statement sequence is ppo_gather_kernel's (widen, square in fp64, wave butterfly): the chunk
// sums, and with them the statistics, are bit-identical to the per-epoch route through the gather's own partials.
template <typename AT>
__global__ __launch_bounds__(64) void adv_parts_keyed_kernel(const AT* __restrict__ adv,
                                                             const catppo_iter_state* __restrict__ st, int64_t total,
                                                             int64_t M, int parts_per_mb, double* __restrict__ parts) {
  const int mbk = blockIdx.y, epoch = blockIdx.z;
  const int64_t m0 = (int64_t)mbk * M;
  const int64_t Mm = (total - m0) < M ? (total - m0) : M;
  const int64_t r0 = (int64_t)blockIdx.x * 64;
  double a1 = 0.0, a2 = 0.0;
  if (r0 + threadIdx.x < Mm) {
    rng::FeistelPerm perm;
    perm.init(st->seed, st->iteration, epoch, total);
    const int64_t src = perm(m0 + r0 + threadIdx.x);
    a1 = (double)(float)adv[src];
    a2 = a1 * a1;
  }
#pragma unroll
  for (int m = 32; m >= 1; m >>= 1) a1 += __shfl_xor(a1, m, 64), a2 += __shfl_xor(a2, m, 64);
  if (threadIdx.x == 0) {
    double* p = parts + 2 * (((int64_t)epoch * gridDim.y + mbk) * parts_per_mb + blockIdx.x);
    p[0] = a1, p[1] = a2;
  }
}

// parts [E * n_mb][parts_per_mb][2] -> moments [E * n_mb][3]: adv_moments_parts_kernel with the epoch folded into the grid
__global__ __launch_bounds__(64) void adv_moments_parts_epochs_kernel(const double* __restrict__ parts, int parts_per_mb,
                                                                      int n_mb, int64_t total, int64_t mb,
                                                                      double* __restrict__ out) {
  const double* p = parts + 2 * (int64_t)blockIdx.x * parts_per_mb;
  double a = 0.0, b = 0.0;
  for (int c = threadIdx.x; c < parts_per_mb; c += 64) a += p[2 * c], b += p[2 * c + 1];
#pragma unroll
  for (int m = 32; m >= 1; m >>= 1) a += __shfl_xor(a, m, 64), b += __shfl_xor(b, m, 64);
  if (threadIdx.x == 0) {
    const int64_t lo = (int64_t)(blockIdx.x % n_mb) * mb, hi = lo + mb < total ? lo + mb : total;
    out[3 * blockIdx.x] = a;
    out[3 * blockIdx.x + 1] = b;
    out[3 * blockIdx.x + 2] = (double)(hi - lo);
  }
}

// moments (after the SUM all-reduce) -> {mean, unbiased std + 1e-8} per minibatch (ppo.py:316-318)
__global__ void adv_stats_kernel(const double* __restrict__ mom, int n, float* __restrict__ stats) {
  const int m = blockIdx.x * blockDim.x + threadIdx.x;
  if (m >= n) return;
  const double cnt = mom[3 * m + 2];
  const double mean = mom[3 * m] / cnt;
  double var = (mom[3 * m + 1] - cnt * mean * mean) / (cnt - 1.0);
  if (var < 0.0) var = 0.0;
  stats[2 * m] = (float)mean;
  stats[2 * m + 1] = (float)sqrt(var) + 1e-8f;
}
}  // namespace

extern "C" int catppo_adv_moments_parts(catppo_ctx* ctx, const double* adv_part_g, int parts_per_mb, int64_t total,
                                        int64_t minibatch, double* moments, void* stream) {
  CATPPO_CHECK_ARG(ctx, ctx != nullptr);
  CATPPO_CHECK_ARG(ctx, adv_part_g && moments && parts_per_mb >= 1 && total >= 1 && minibatch >= 1);
  CATPPO_CHECK_ARG(ctx, (int64_t)parts_per_mb * 64 >= minibatch);
  const int64_t n_mb = cdiv64(total, minibatch);
  hipLaunchKernelGGL(adv_moments_parts_kernel, dim3((unsigned)n_mb), dim3(64), 0, static_cast<hipStream_t>(stream),
                     adv_part_g, parts_per_mb, total, minibatch, moments);
  CATPPO_CHECK_LAUNCH(ctx);
  return CATPPO_OK;
}

extern "C" int catppo_adv_moments_keyed(catppo_ctx* ctx, const void* advantages, int adv_dtype,
                                        const catppo_iter_state* state, int32_t n_epochs, int64_t total,
                                        int64_t minibatch, double* parts_scratch, double* moments, void* stream) {
  CATPPO_CHECK_ARG(ctx, ctx != nullptr);
  CATPPO_CHECK_ARG(ctx, advantages && state && parts_scratch && moments);
  CATPPO_CHECK_ARG(ctx, adv_dtype == CATPPO_F32 || adv_dtype == CATPPO_F16);
  CATPPO_CHECK_ARG(ctx, n_epochs >= 1 && n_epochs <= 65535 && total >= 1 && total < (int64_t(1) << 31) && minibatch >= 1);
  const int64_t n_mb = cdiv64(total, minibatch);
  const int64_t parts = cdiv64(minibatch, 64);
  CATPPO_CHECK_ARG(ctx, n_mb <= 65535 && parts <= (int64_t(1) << 30));
  hipStream_t s = static_cast<hipStream_t>(stream);
  const dim3 grid((unsigned)parts, (unsigned)n_mb, (unsigned)n_epochs);
  if (adv_dtype == CATPPO_F16)
    hipLaunchKernelGGL(adv_parts_keyed_kernel<_Float16>, grid, dim3(64), 0, s, static_cast<const _Float16*>(advantages),
                       state, total, minibatch, (int)parts, parts_scratch);
  else
    hipLaunchKernelGGL(adv_parts_keyed_kernel<float>, grid, dim3(64), 0, s, static_cast<const float*>(advantages), state,
                       total, minibatch, (int)parts, parts_scratch);
  CATPPO_CHECK_LAUNCH(ctx);
  hipLaunchKernelGGL(adv_moments_parts_epochs_kernel, dim3((unsigned)(n_mb * n_epochs)), dim3(64), 0, s,
                     (const double*)parts_scratch, (int)parts, (int)n_mb, total, minibatch, moments);
  CATPPO_CHECK_LAUNCH(ctx);
  return CATPPO_OK;
}

extern "C" int catppo_adv_stats(catppo_ctx* ctx, const double* moments, int n_minibatches, float* stats, void* stream) {
  CATPPO_CHECK_ARG(ctx, ctx != nullptr);
  CATPPO_CHECK_ARG(ctx, moments && stats && n_minibatches >= 1);
  hipLaunchKernelGGL(adv_stats_kernel, dim3((n_minibatches + 63) / 64), dim3(64), 0, static_cast<hipStream_t>(stream),
                     moments, n_minibatches, stats);
  CATPPO_CHECK_LAUNCH(ctx);
  return CATPPO_OK;
}
