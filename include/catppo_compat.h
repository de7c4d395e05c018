/* catppo_compat.h - the entry-point NAMES of ABI <= 0.5 as static inline wrappers over ABI 0.6 (include/catppo.h).
 *
 * ABI 0.6 collapsed the accreted `_ex` / `_f16` / `_rng` / `_mode` families into one exported entry each (77 -> 57
 * exports).  Source written against the old names keeps compiling with this header; nothing here is
 * exported by libcatppo.so, and a binding (ctypes, cgo ...) should bind the 0.6 names directly. */
#ifndef CATPPO_COMPAT_H
#define CATPPO_COMPAT_H

#include <stddef.h>

#include "catppo.h"

#ifdef __cplusplus
extern "C" {
#endif

/* ---- GAE (cleanrl/ppo.py:251-277; rl_games / skrl variants) -> catppo_gae_planes */
static inline int catppo_gae_ex(catppo_ctx* ctx, int kind, const float* rewards, const float* values, const float* dones,
                                const float* true_dones, const float* next_value, const float* next_done,
                                const float* next_true_done, float gamma, float gamma_lambda, float* advantages,
                                float* returns, int T, int64_t N, void* stream) {
  return catppo_gae_planes(ctx, kind, CATPPO_GAE_SERIAL, CATPPO_F32, rewards, values, dones, true_dones, next_value, next_done,
                           next_true_done, gamma, gamma_lambda, advantages, returns, T, N, stream);
}
static inline int catppo_gae(catppo_ctx* ctx, const float* rewards, const float* values, const float* dones,
                             const float* true_dones, const float* next_value, const float* next_done,
                             const float* next_true_done, float gamma, float gamma_lambda, float* advantages,
                             float* returns, int T, int64_t N, void* stream) {
  return catppo_gae_ex(ctx, CATPPO_GAE_CLEANRL, rewards, values, dones, true_dones, next_value, next_done, next_true_done,
                       gamma, gamma_lambda, advantages, returns, T, N, stream);
}
static inline int catppo_gae_f16(catppo_ctx* ctx, int kind, const void* rewards, const void* values, const void* dones,
                                 const void* true_dones, const void* next_value, const void* next_done,
                                 const void* next_true_done, float gamma, float gamma_lambda, void* advantages,
                                 void* returns, int T, int64_t N, void* stream) {
  return catppo_gae_planes(ctx, kind, CATPPO_GAE_SERIAL, CATPPO_F16, rewards, values, dones, true_dones, next_value, next_done,
                           next_true_done, gamma, gamma_lambda, advantages, returns, T, N, stream);
}
static inline int catppo_gae_mode(catppo_ctx* ctx, int mode, const float* rewards, const float* values, const float* dones,
                                  const float* true_dones, const float* next_value, const float* next_done,
                                  const float* next_true_done, float gamma, float gamma_lambda, float* advantages,
                                  float* returns, int T, int64_t N, void* stream) {
  return catppo_gae_planes(ctx, CATPPO_GAE_CLEANRL, mode, CATPPO_F32, rewards, values, dones, true_dones, next_value, next_done,
                           next_true_done, gamma, gamma_lambda, advantages, returns, T, N, stream);
}

/* ---- rollout forward (cleanrl/ppo.py:104-119,186-189,251) -> catppo_policy_step */
static inline int catppo_policy_act_ex(catppo_ctx* ctx, const catppo_mlp_shape* shape, const float* params, const float* x,
                                       int64_t N, const float* eps, const float* given_action, float* action,
                                       float* logprob, void* value, int value_dtype, void* stream) {
  return catppo_policy_step(ctx, shape, params, x, N, eps, given_action, NULL, 0, NULL, action, logprob, value, value_dtype, stream);
}
static inline int catppo_policy_act(catppo_ctx* ctx, const catppo_mlp_shape* shape, const float* params, const float* x,
                                    int64_t N, const float* eps, const float* given_action, float* action, float* logprob,
                                    float* value, void* stream) {
  return catppo_policy_act_ex(ctx, shape, params, x, N, eps, given_action, action, logprob, value, CATPPO_F32, stream);
}
static inline int catppo_policy_act_rng(catppo_ctx* ctx, const catppo_mlp_shape* shape, const float* params, const float* x,
                                        int64_t N, const catppo_iter_state* state, int32_t step, float* eps_out,
                                        float* action, float* logprob, void* value, int value_dtype, void* stream) {
  if (state == NULL) return CATPPO_E_ARG;
  return catppo_policy_step(ctx, shape, params, x, N, NULL, NULL, state, step, eps_out, action, logprob, value, value_dtype, stream);
}
static inline int catppo_value_ex(catppo_ctx* ctx, const catppo_mlp_shape* shape, const float* params, const float* x,
                                  int64_t N, void* value, int value_dtype, void* stream) {
  return catppo_policy_step(ctx, shape, params, x, N, NULL, NULL, NULL, 0, NULL, NULL, NULL, value, value_dtype, stream);
}
static inline int catppo_value(catppo_ctx* ctx, const catppo_mlp_shape* shape, const float* params, const float* x, int64_t N,
                               float* value, void* stream) {
  return catppo_value_ex(ctx, shape, params, x, N, value, CATPPO_F32, stream);
}

/* ---- epoch gather (cleanrl/ppo.py:295-302) -> catppo_ppo_gather_ex */
static inline int catppo_ppo_gather(catppo_ctx* ctx, const catppo_mlp_shape* shape, const float* b_obs, const float* b_actions,
                                    const float* b_logprobs, const float* b_advantages, const float* b_returns_n,
                                    const float* b_values_n, const int64_t* inds, int64_t total, int64_t M, float* x_g,
                                    float* act_g, float* scal_g, double* adv_part_g, void* stream) {
  return catppo_ppo_gather_ex(ctx, shape, b_obs, b_actions, b_logprobs, b_advantages, CATPPO_F32, b_returns_n, b_values_n, inds,
                              NULL, 0, total, M, x_g, act_g, scal_g, adv_part_g, NULL, stream);
}

/* ---- RunningMeanStd (cleanrl/ppo.py:12-62), fp32 input -> the _ex entries */
static inline int catppo_rms_moments(catppo_ctx* ctx, const float* x, int64_t N, int D, int64_t ldx, double* sums, void* stream) {
  return catppo_rms_moments_ex(ctx, x, CATPPO_F32, N, D, ldx, sums, stream);
}
static inline int catppo_rms_update(catppo_ctx* ctx, const float* x, int64_t N, int D, int64_t ldx, float* mean, float* var,
                                    float* count, void* stream) {
  return catppo_rms_update_ex(ctx, x, CATPPO_F32, N, D, ldx, mean, var, count, stream);
}
static inline int catppo_rms_normalize(catppo_ctx* ctx, const float* x, int64_t N, int D, int64_t ldx, const float* mean,
                                       const float* var, float eps, float* out, int64_t ldo, void* stream) {
  return catppo_rms_normalize_ex(ctx, x, CATPPO_F32, N, D, ldx, mean, var, eps, out, ldo, stream);
}

/* ---- rollout buffer rows (cleanrl/ppo.py:215-226), fp32 planes -> catppo_rollout_store_ex */
static inline int catppo_rollout_store(catppo_ctx* ctx, const float* reward, const float* dones, const uint8_t* time_outs,
                                       float* rewards_t, float* dones_t1, float* true_dones_t1, int64_t N, void* stream) {
  return catppo_rollout_store_ex(ctx, reward, dones, time_outs, rewards_t, dones_t1, true_dones_t1, CATPPO_F32, N, stream);
}

/* ---- small merges */
static inline int catppo_rollout_flush(catppo_ctx* ctx, void* stream) { return catppo_rollout_defer_tail(ctx, -1, stream); }
static inline int catppo_graph_abort(catppo_ctx* ctx, void* stream) { return catppo_graph_end(ctx, stream, NULL, NULL); }
static inline int catppo_comm_world(catppo_ctx* ctx) {
  const int w = ctx ? catppo_comm_probe(ctx) : 0;
  return w > 0 ? w : 0;
}
static inline int catppo_grad_overlap_active(catppo_ctx* ctx) { return ctx ? catppo_set_grad_overlap(ctx, -1) : 0; }
static inline uint64_t catppo_rollout_xchg_bytes(int K, int D) {
  uint64_t b = 0, o = 0;
  return catppo_rollout_xchg_layout(K, D, &b, &o) == CATPPO_OK ? b : 0;
}
static inline uint64_t catppo_rollout_xchg_sum_offset(int K) {
  uint64_t b = 0, o = 0;
  return catppo_rollout_xchg_layout(K, 1, &b, &o) == CATPPO_OK ? o : 0;
}

#ifdef __cplusplus
}
#endif
#endif /* CATPPO_COMPAT_H */
