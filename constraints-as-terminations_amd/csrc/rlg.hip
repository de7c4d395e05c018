// rl_games front end: the per-step episode bookkeeping of CaTA2CAgent.play_steps with FLOAT dones
// (reference rl_games/cat_common.py:71-92) as one launch, no host synchronisation:
//
//     current_rewards += rewards; current_shaped_rewards += shaped_rewards; current_lengths += 1
//     done_i = dones_i >= 1.0                       ( dones.ge(1.0): only a CERTAIN termination ends an episode )
//     game_rewards / game_shaped_rewards / game_lengths .update( current_*[done] )      (rl_games AverageMeter)
//     current_rewards *= (1 - dones); current_shaped_rewards *= (1 - dones)             ( float not_dones: a
//     current_lengths[done] = 0                       termination PROBABILITY scales the running return down )
//
// The reference does this with ~12 eager launches and a nonzero() host sync per env step.  The three AverageMeters
// (running means over the last `max_size` finished episodes; rl_games is not vendored - its published update rule is
// restated in catppo.h, PARITY UNPINNED against rl_games itself) live in device memory (catppo_rlg_meters).
#include "common.h"
#include "xwg.h"

namespace {

constexpr int kMaxV = CATPPO_RLG_MAX_VALUE_SIZE;
constexpr int kTicketRlg = 60;      // catppo_ctx::tickets slot (rollout_pre uses 0..32, rollout_post 40)

// torch_ext.AverageMeter.update(values) with  n = number of rows of `values`, sum = their column sum
__device__ __forceinline__ void meter_update(float& mean, int32_t& cur, int max_size, double sum, int64_t n) {
  if (n == 0) return;
  const float new_mean = (float)(sum / (double)n);
  const int size = n < max_size ? (int)n : max_size;                 // np.clip(size, 0, max_size)
  const int old_size = (max_size - size) < cur ? (max_size - size) : cur;
  const int size_sum = old_size + size;
  cur = size_sum;
  mean = (mean * (float)old_size + new_mean * (float)size) / (float)size_sum;
}

__global__ __launch_bounds__(256) void rlg_episode_step_kernel(const float* __restrict__ rewards,
                                                               const float* __restrict__ shaped,
                                                               const float* __restrict__ dones, int V,
                                                               float* __restrict__ cur_rew, float* __restrict__ cur_shaped,
                                                               float* __restrict__ cur_len, int64_t N,
                                                               double* __restrict__ part, unsigned int* __restrict__ ticket,
                                                               catppo_rlg_meters* __restrict__ meters,
                                                               uint8_t* __restrict__ done_mask) {
  constexpr int NS = 2 + 2 * kMaxV;                  // count, sum len, sum rew[V], sum shaped[V]
  __shared__ double sm[4][NS];
  double acc[NS];
#pragma unroll
  for (int q = 0; q < NS; ++q) acc[q] = 0.0;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < N; i += (int64_t)gridDim.x * 256) {
    const float d = dones[i];
    const bool done = d >= 1.0f;
    const float nd = 1.0f - d;
    const float len = cur_len[i] + 1.0f;
    if (done) acc[0] += 1.0, acc[1] += (double)len;
    for (int v = 0; v < V; ++v) {
      const float r = cur_rew[i * V + v] + rewards[i * V + v];
      const float s = cur_shaped[i * V + v] + shaped[i * V + v];
      if (done) acc[2 + v] += (double)r, acc[2 + kMaxV + v] += (double)s;
      cur_rew[i * V + v] = r * nd;
      cur_shaped[i * V + v] = s * nd;
    }
    cur_len[i] = done ? 0.0f : len;
    if (done_mask) done_mask[i] = done;
  }
#pragma unroll
  for (int q = 0; q < NS; ++q) {
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) acc[q] += __shfl_xor(acc[q], m, 64);
  }
  if ((threadIdx.x & 63) == 0)
    for (int q = 0; q < NS; ++q) sm[threadIdx.x >> 6][q] = acc[q];
  __syncthreads();
  if (threadIdx.x < NS) {
    const int q = threadIdx.x;
    xwg_store(part + (int64_t)blockIdx.x * NS + q, (sm[0][q] + sm[1][q]) + (sm[2][q] + sm[3][q]));
  }
  // last workgroup to arrive folds the partials in block order (result independent of arrival order).  The hand-shake
  // is xwg.h's (shared with rollout.hip): device-scope atomic stores / loads ordered by the stores' completion instead
  // of device-scope fences (an L2 write-back / invalidate per workgroup on gfx950); -DROLLOUT_FENCES=1 builds the fence
  // version here too.  (last_block_arrives re-arms the ticket for the next launch.)
  if (!last_block_arrives(ticket, gridDim.x)) return;
  if (threadIdx.x < NS) {
    const int q = threadIdx.x;
    double s = 0.0;
    for (unsigned b = 0; b < gridDim.x; ++b) s += xwg_load(part + (int64_t)b * NS + q);
    sm[0][q] = s;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    const int64_t n = (int64_t)(sm[0][0] + 0.5);
    const int max_size = meters->max_size;
    meters->last_done_count = (int32_t)n;
    meter_update(meters->mean_lengths, meters->size_lengths, max_size, sm[0][1], n);
    int32_t c1 = meters->size_rewards, c2 = meters->size_shaped;
    for (int v = 0; v < V; ++v) {
      int32_t a = meters->size_rewards, b = meters->size_shaped;
      meter_update(meters->mean_rewards[v], a, max_size, sm[0][2 + v], n);
      meter_update(meters->mean_shaped_rewards[v], b, max_size, sm[0][2 + kMaxV + v], n);
      c1 = a, c2 = b;
    }
    meters->size_rewards = c1, meters->size_shaped = c2;
  }
}

__global__ void rlg_meters_init_kernel(catppo_rlg_meters* m, int max_size) {
  if (threadIdx.x == 0 && blockIdx.x == 0) {
    for (int v = 0; v < kMaxV; ++v) m->mean_rewards[v] = 0.0f, m->mean_shaped_rewards[v] = 0.0f;
    m->mean_lengths = 0.0f;
    m->size_rewards = m->size_shaped = m->size_lengths = 0;
    m->max_size = max_size;
    m->last_done_count = 0;
  }
}

}  // namespace

extern "C" int catppo_rlg_meters_init(catppo_ctx* ctx, catppo_rlg_meters* meters, int max_size, void* stream) {
  CATPPO_CHECK_ARG(ctx, ctx != nullptr);
  CATPPO_CHECK_ARG(ctx, meters != nullptr && max_size >= 1);
  hipLaunchKernelGGL(rlg_meters_init_kernel, dim3(1), dim3(64), 0, static_cast<hipStream_t>(stream), meters, max_size);
  CATPPO_CHECK_LAUNCH(ctx);
  return CATPPO_OK;
}

extern "C" int catppo_rlg_episode_step(catppo_ctx* ctx, const float* rewards, const float* shaped_rewards,
                                       const float* dones, int value_size, float* current_rewards,
                                       float* current_shaped_rewards, float* current_lengths, int64_t N,
                                       catppo_rlg_meters* meters, uint8_t* done_mask_out, void* stream) {
  CATPPO_CHECK_ARG(ctx, ctx != nullptr);
  CATPPO_CHECK_ARG(ctx, rewards && shaped_rewards && dones && current_rewards && current_shaped_rewards &&
                            current_lengths && meters && N >= 1);
  CATPPO_CHECK_ARG(ctx, value_size >= 1 && value_size <= kMaxV);
  int64_t nblk = cdiv64(N, 256);
  if (nblk > 256) nblk = 256;
  WsCarver ws(ctx);
  double* part = ws.take<double>((uint64_t)nblk * (2 + 2 * kMaxV));
  CATPPO_NEED_WS(ctx, part);
  hipLaunchKernelGGL(rlg_episode_step_kernel, dim3((unsigned)nblk), dim3(256), 0, static_cast<hipStream_t>(stream),
                     rewards, shaped_rewards, dones, value_size, current_rewards, current_shaped_rewards,
                     current_lengths, N, part, ctx->tickets + kTicketRlg, meters, done_mask_out);
  CATPPO_CHECK_LAUNCH(ctx);
  return CATPPO_OK;
}
