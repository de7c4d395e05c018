// Device-resident iteration state: learning-rate schedules, Adam step counter, RNG counters.
//
// The reference keeps these in Python (cleanrl/ppo.py:196-199 linear anneal; torch.optim.Adam's step count; the
// torch generator) and, for its skrl / rl_games front-ends, adapts the learning rate to the measured KL on the
// host (skrl/ppo.py:558-567 incl. the KL all-reduce; rl_games_cat_solo.yaml:64-66).  Here they live in one small
// device struct (catppo_iter_state) written only by kernels: launches become identical from iteration to
// iteration (=> replayable hipGraphs) and the KL schedule needs no device->host read.
#include "common.h"

namespace {

__global__ void iter_init_kernel(catppo_iter_state* st, uint64_t seed, double lr) {
  st->seed = seed;
  st->iteration = 0;
  st->adam_step = 0;
  st->lr = lr;
  st->kl_mark = 0.0;
  st->n_mark = 0.0;
  st->last_kl = 0.0;
  st->adam_step_size = 0.0f, st->adam_bc2_sqrt = 0.0f;
}

__global__ void iter_begin_kernel(catppo_iter_state* st, double lr0, double num_iterations, int schedule) {
  const int64_t it = st->iteration + 1;
  st->iteration = it;
  st->kl_mark = 0.0;   // the per-iteration diagnostics (diag[8]) restart from zero
  st->n_mark = 0.0;
  if (schedule == CATPPO_LR_FIXED) {
    st->lr = lr0;
  } else if (schedule == CATPPO_LR_LINEAR) {
    // frac = 1.0 - (iteration - 1.0) / NUM_ITERATIONS; lrnow = frac * LEARNING_RATE   (Python doubles)
    const double frac = 1.0 - ((double)it - 1.0) / num_iterations;
    st->lr = frac * lr0;
  }
}

__global__ void kl_mean_kernel(catppo_iter_state* st, const float* __restrict__ diag, float* __restrict__ kl_out) {
  const double s = (double)diag[4], n = (double)diag[7];
  const double dn = n - st->n_mark;
  kl_out[0] = dn > 0.0 ? (float)((s - st->kl_mark) / dn) : 0.0f;
  st->kl_mark = s;
  st->n_mark = n;
}

__global__ void kl_adaptive_lr_kernel(catppo_iter_state* st, const float* __restrict__ kl, double thr, double kf,
                                      double lf, double min_lr, double max_lr) {
  const double k = (double)kl[0];
  double lr = st->lr;
  if (k > thr * kf) {
    lr = lr / lf;
    lr = lr < min_lr ? min_lr : lr;
  } else if (k < thr / kf) {
    lr = lr * lf;
    lr = lr > max_lr ? max_lr : lr;
  }
  st->lr = lr;
  st->last_kl = k;
}

}  // namespace

extern "C" int catppo_iter_init(catppo_ctx* ctx, catppo_iter_state* state, uint64_t seed, double lr, void* stream) {
  CATPPO_CHECK_ARG(ctx, ctx != nullptr);
  CATPPO_CHECK_ARG(ctx, state != nullptr);
  hipLaunchKernelGGL(iter_init_kernel, dim3(1), dim3(1), 0, static_cast<hipStream_t>(stream), state, seed, lr);
  CATPPO_CHECK_LAUNCH(ctx);
  return CATPPO_OK;
}

extern "C" int catppo_iter_begin(catppo_ctx* ctx, catppo_iter_state* state, double lr0, int64_t num_iterations,
                                 int schedule, void* stream) {
  CATPPO_CHECK_ARG(ctx, ctx != nullptr);
  CATPPO_CHECK_ARG(ctx, state != nullptr && schedule >= CATPPO_LR_FIXED && schedule <= CATPPO_LR_KEEP);
  CATPPO_CHECK_ARG(ctx, schedule != CATPPO_LR_LINEAR || num_iterations >= 1);
  hipLaunchKernelGGL(iter_begin_kernel, dim3(1), dim3(1), 0, static_cast<hipStream_t>(stream), state, lr0,
                     (double)num_iterations, schedule);
  CATPPO_CHECK_LAUNCH(ctx);
  return CATPPO_OK;
}

extern "C" int catppo_kl_mean(catppo_ctx* ctx, catppo_iter_state* state, const float* diag, float* kl_out,
                              void* stream) {
  CATPPO_CHECK_ARG(ctx, ctx != nullptr);
  CATPPO_CHECK_ARG(ctx, state && diag && kl_out);
  hipLaunchKernelGGL(kl_mean_kernel, dim3(1), dim3(1), 0, static_cast<hipStream_t>(stream), state, diag, kl_out);
  CATPPO_CHECK_LAUNCH(ctx);
  return CATPPO_OK;
}

extern "C" int catppo_kl_adaptive_lr(catppo_ctx* ctx, catppo_iter_state* state, const float* kl, double kl_threshold,
                                     double kl_factor, double lr_factor, double min_lr, double max_lr, void* stream) {
  CATPPO_CHECK_ARG(ctx, ctx != nullptr);
  CATPPO_CHECK_ARG(ctx, state && kl && kl_threshold > 0.0 && kl_factor >= 1.0 && lr_factor >= 1.0 && min_lr > 0.0 &&
                            max_lr >= min_lr);
  hipLaunchKernelGGL(kl_adaptive_lr_kernel, dim3(1), dim3(1), 0, static_cast<hipStream_t>(stream), state, kl,
                     kl_threshold, kl_factor, lr_factor, min_lr, max_lr);
  CATPPO_CHECK_LAUNCH(ctx);
  return CATPPO_OK;
}
