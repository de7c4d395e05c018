/*
 * catppo.h - C ABI of libcatppo.so: the MI355X (gfx950) kernels underneath the
 * CaT + CleanRL-PPO hot path.
 *
 * The reference (Gepetto/constraints-as-terminations) has no FFI: the path sits behind
 * Python interfaces.  Each entry point below replaces a sequence of eager
 * torch ops of the reference; the citation after "replaces:" is the reference location,
 * relative to /root/reference/exts/cat_envs/cat_envs/tasks/utils/.
 *
 * Conventions
 *   - extern "C", plain pointers and sizes, no torch types.  All tensor memory is
 *     caller-owned DEVICE memory (fp32 unless stated), row-major, densely packed unless a
 *     leading dimension `ld*` is given.  The library owns only its workspace.
 *   - every call ENQUEUES on `stream` (a hipStream_t passed as void*) and never
 *     synchronises; no allocation after catppo_create / catppo_reserve.
 *   - return value: 0 = ok, <0 = error (CATPPO_E_*); text via catppo_last_error(ctx).
 *   - a ctx is bound to one device and is not thread-safe (one host thread per ctx,
 *     like the reference's single-threaded loop).
 *   - scalars that the reference computes in Python double and then applies to an fp32
 *     tensor (tau, 1-tau, max_p-min_p, gamma, gamma*lambda) are passed ALREADY ROUNDED to
 *     fp32 by the caller; the kernels use unfused IEEE fp32 (no FMA contraction, correctly
 *     rounded division / sqrt) so that termination masks are bit-exact.
 */
#ifndef CATPPO_H
#define CATPPO_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define CATPPO_VERSION 600 /* 0.6.0: the _ex families collapsed into one entry each (58 exports, 77 in 0.5); the
                              names of ABI <= 0.5 are static inline wrappers in catppo_compat.h */

#define CATPPO_OK 0
#define CATPPO_E_ARG (-1)     /* bad argument */
#define CATPPO_E_HIP (-2)     /* HIP runtime error (launch, alloc) */
#define CATPPO_E_NODEV (-3)   /* no gfx950 device / device index out of range */
#define CATPPO_E_WORKSPACE (-4) /* workspace too small: call catppo_reserve first */
#define CATPPO_E_COMM (-5)    /* RCCL error / communicator not initialised */

/* element types of the buffers that exist in more than one precision (rollout planes, collectives) */
#define CATPPO_F32 0
#define CATPPO_F16 1
#define CATPPO_F64 2

typedef struct catppo_ctx catppo_ctx;

/* ---- lifecycle ------------------------------------------------------------------- */
int catppo_version(void);
/* ABI 0.5: which kernels does a shape get?  enable > 0 starts recording (clears the log), 0 stops, < 0 only reads; returns
 * the text recorded so far: one line per launch decision of catppo_policy_act* / catppo_value* /
 * catppo_ppo_minibatch_grad* / _step_packed (kernel, grid shape, the rule that selected it), written at the decision
 * sites themselves.  tools/explain_plan.py prints it for a (obs_dim, hidden, rows) triple. */
const char* catppo_plan_log(catppo_ctx* ctx, int enable);
int catppo_create(int device, catppo_ctx** out);
void catppo_destroy(catppo_ctx* ctx);
const char* catppo_last_error(catppo_ctx* ctx);
/* make sure the internal workspace holds at least `bytes` (never called implicitly while
 * a stream capture is active; returns CATPPO_E_WORKSPACE from compute calls otherwise). */
int catppo_reserve(catppo_ctx* ctx, uint64_t bytes);

/* ---- CaT: constraints -> termination probability ------------------------------------
 * cstr        [N,K]   constraint values of ALL terms side by side (positive = violated;
 *                     bool terms already converted to 0/1), term t owns the columns
 *                     [term_off[t], term_off[t+1]).
 * term_off    [n_terms+1] int32, HOST array: column offsets, term_off[0]==0, term_off[n_terms]==K
 * term_dp     [n_terms]  fp32, HOST array: fl32(max_p_t - min_p); the curriculum rewrites max_p
 *                        on the host at every reset, so both arrays travel in the kernel arguments
 *                        (n_terms <= 64)
 * rm          [K]     running maxima (state, in/out)
 * reward      [N]     in/out, may be NULL      r <- max(fl(r * fl(1-p)), 0)
 * reset_mask  [N]     uint8/bool, may be NULL  dones[i] = reset ? 1 : p
 * cstr_prob   [N]     out    p_i = max_j prob_ij
 * dones       [N]     out, may be NULL
 * ep_viol     [n_terms,N] in/out   += (max_{j in t} prob_ij > 0)
 * ep_prob     [n_terms,N] in/out   += max_{j in t} prob_ij
 * probs       [N,K]   out, may be NULL (per column probabilities, CaT.probs)
 *
 * replaces: cat/constraint_manager.py:39-82 (CaT.add per term, CaT.get_probs),
 *           :213-229 (ConstraintManager.compute statistics),
 *           cat/cat_env.py:102-107,118-121 (reward scaling, float dones, hard resets).
 */
int catppo_cat_step(catppo_ctx* ctx, const float* cstr, int64_t N, int K,
                    const int32_t* term_off, int n_terms, const float* term_dp,
                    float min_p, float tau, float one_minus_tau, int first_call,
                    float* rm, float* reward, const uint8_t* reset_mask,
                    float* cstr_prob, float* dones, float* ep_viol, float* ep_prob,
                    float* probs, void* stream);

/* Two-phase form for env-sharded runs (a MAX all-reduce of `colmax` goes in between):
 *   catppo_cat_colmax : colmax[j] = max(max_i cstr[i,j], 1e-6)        (:55)
 *   catppo_cat_apply  : running-max EMA from (all-reduced) colmax, then everything else. */
int catppo_cat_colmax(catppo_ctx* ctx, const float* cstr, int64_t N, int K, float* colmax,
                      void* stream);
int catppo_cat_apply(catppo_ctx* ctx, const float* cstr, int64_t N, int K,
                     const int32_t* term_off, int n_terms, const float* term_dp,
                     float min_p, float tau, float one_minus_tau, int first_call,
                     const float* colmax, float* rm, float* reward,
                     const uint8_t* reset_mask, float* cstr_prob, float* dones,
                     float* ep_viol, float* ep_prob, float* probs, void* stream);

/* ConstraintManager.reset: per-term episode statistics of the envs selected by `mask`
 * (uint8/bool [N]; NULL = all envs), then zero their accumulators:
 *   out[2t]   = mean_i(ep_viol[t,i] / len_i) * 100      "Episode_Constraint_violation/<term>"
 *   out[2t+1] = mean_i(ep_prob[t,i] / len_i)            "Episode_Constraint_probability/<term>"
 * If no env is selected `out` receives `prev` (may be NULL: `out` untouched) - the reference keeps the
 * last log dict in env.extras until the next reset.  episode_length: int64 [N] (IsaacLab's
 * episode_length_buf); a zero length gives NaN/inf exactly like the reference's first reset.
 * replaces: cat/constraint_manager.py:190-211. */
int catppo_cat_reset(catppo_ctx* ctx, float* ep_viol, float* ep_prob, const int64_t* episode_length,
                     const uint8_t* mask, int n_terms, int64_t N, const float* prev, float* out, void* stream);

/* Solo12 constraint terms evaluated straight from sim-state tensors into the cstr matrix.
 * `desc` is a host array of n_terms descriptors (see catppo_term_desc).
 * replaces: cat/constraints.py:23-235 (C1..C15). */
enum catppo_term_kind {
  CATPPO_TERM_ABS_LIMIT = 0,      /* |x[:,ids]| - limit                    C1,C3,C4,C5 */
  CATPPO_TERM_ABS_DIFF_LIMIT = 1, /* |x[:,ids]-y[:,ids]| - limit           C11 */
  CATPPO_TERM_ABS_DIFF_LIMIT_GATE_CMDY = 2, /* (|x-y|-limit)*[|cmd_y|<dz]  C2 */
  CATPPO_TERM_GREATER = 3,        /* x[:,col] > limit  (bool -> 0/1)       C6 */
  CATPPO_TERM_CONTACT_ANY = 4,    /* any_b(max_h |F_hb| > thr)             C7 */
  CATPPO_TERM_NORM2_LIMIT = 5,    /* ||x[:, :2]|| - limit                  C8 */
  CATPPO_TERM_AIR_TIME = 6,       /* (limit-last_air)*touchdown*[|cmd|>dz] C9 */
  CATPPO_TERM_N_FOOT_CONTACT = 7, /* |#contacts - n| * [|cmd|>minc]        C10 */
  CATPPO_TERM_ACTION_RATE = 8,    /* |a-a_prev|/dt - limit                 C12 */
  CATPPO_TERM_FORCE_LIMIT = 9,    /* max_h |F_hb| - limit                  C13 */
  CATPPO_TERM_LIMIT_MINUS = 10,   /* limit - x[:,col]                      C14 */
  CATPPO_TERM_ABS_LIMIT_GATE_CMDNORM_LT = 11 /* (|x|-limit)*[|cmd|<dz]     C15 */
};

#define CATPPO_TERM_MAX_IDS 32
typedef struct catppo_term_desc {
  int32_t kind;          /* catppo_term_kind */
  int32_t width;         /* output columns */
  int32_t n_ids;         /* number of joint/body ids used (<= CATPPO_TERM_MAX_IDS) */
  int32_t ids[CATPPO_TERM_MAX_IDS]; /* joint ids, body ids, or ids[0] = column (Solo12: 12 joints, 17 bodies) */
  float limit;           /* limit / threshold */
  float aux;             /* dead-zone / min command / desired feet / step_dt */
  const float* x;        /* primary state tensor (N, x_ld) */
  const float* y;        /* secondary (default_joint_pos, prev_action, first_contact as 0/1) */
  int32_t x_ld, y_ld;
} catppo_term_desc;

/* forces: (N,H,B,3) net_forces_w_history, env i at forces + i*forces_env_stride (= H*B*3 when
 * dense); command: (N,3) with leading dimension command_ld.  Either may be NULL if no term uses
 * it.  cstr [N,K] out, K = sum of widths. */
int catppo_cat_terms(catppo_ctx* ctx, const catppo_term_desc* desc, int n_terms, int64_t N,
                     const float* forces, int64_t forces_env_stride, int H, int B,
                     const float* command, int command_ld, float* cstr, int K, void* stream);

/* catppo_cat_terms + catppo_cat_step in three launches: the term kernel also emits the per-workgroup column
 * maxima, so the CaT step does not re-read cstr for them.  Same arguments as the two calls (n_terms is both
 * the descriptor count and the term count of term_off / term_dp).  Single-process runs only: env-sharded runs
 * need the all-reduce point between catppo_cat_colmax and catppo_cat_apply. */
int catppo_cat_terms_step(catppo_ctx* ctx, const catppo_term_desc* desc, int n_terms, int64_t N,
                          const float* forces, int64_t forces_env_stride, int H, int B, const float* command,
                          int command_ld, float* cstr, int K, const int32_t* term_off, const float* term_dp,
                          float min_p, float tau, float one_minus_tau, int first_call, float* rm, float* reward,
                          const uint8_t* reset_mask, float* cstr_prob, float* dones, float* ep_viol,
                          float* ep_prob, float* probs, void* stream);

/* Env-sharded variant of the same fusion: terms -> cstr plus the local (floored) column maxima in two launches;
 * all-reduce `colmax` with MAX, then catppo_cat_apply. */
int catppo_cat_terms_colmax(catppo_ctx* ctx, const catppo_term_desc* desc, int n_terms, int64_t N,
                            const float* forces, int64_t forces_env_stride, int H, int B, const float* command,
                            int command_ld, float* cstr, int K, float* colmax, void* stream);

/* ---- env-step bookkeeping, fused ------------------------------------------------------------
 * catppo_env_pre_step: process_action (prev <- action, action <- action_in, [N,A]); episode_length += 1;
 *   time_outs = episode_length >= max_episode_length; terminated = hard_reset > 0.5; reset = either;
 *   reward_out = reward_src (the reward manager's output, strided view allowed).  time_outs /
 *   terminated / reset are uint8 (torch.bool) [N].   replaces: cat/cat_env.py:62,92-97.
 * catppo_rollout_store: rewards[step] = reward, dones[step+1] = dones, true_dones[step+1] =
 *   float(time_outs).   replaces: cleanrl/ppo.py:215-216,226. */
int catppo_env_pre_step(catppo_ctx* ctx, const float* action_in, float* action, float* prev_action, int A,
                        int64_t* episode_length, int64_t max_episode_length, const float* hard_reset,
                        int64_t hard_reset_stride, const float* reward_src, int64_t reward_stride,
                        uint8_t* time_outs, uint8_t* terminated, uint8_t* reset, float* reward_out,
                        int64_t N, void* stream);

/* env-sharded exact mode: advantage moments of every minibatch of a (multi-)epoch permutation in one
 * launch, moments[m] = {sum, sum of squares, count} (fp64) over adv[inds[m*mb : (m+1)*mb]]; after ONE
 * SUM all-reduce catppo_adv_stats turns them into {mean, unbiased std + 1e-8} per minibatch, the
 * `adv_stats` input of catppo_ppo_minibatch_grad.   replaces: cleanrl/ppo.py:314-318 across ranks. */
int catppo_adv_stats(catppo_ctx* ctx, const double* moments, int n_minibatches, float* stats, void* stream);
/* the same moments from the per-chunk partial sums catppo_ppo_gather[_ex] already wrote (`adv_part_g`:
 * [n_mb][parts_per_mb][2] fp64, parts_per_mb = ceil(minibatch / 64)) - no index array, any advantage plane
 * precision (the gather widened fp16 exactly).  moments[m] = {sum, sum of squares, rows of minibatch m}. */
int catppo_adv_moments_parts(catppo_ctx* ctx, const double* adv_part_g, int parts_per_mb, int64_t total,
                             int64_t minibatch, double* moments, void* stream);

/* ---- GAE -------------------------------------------------------------------------------
 * time-major (T,N) buffers; float dones in [0,1]; separate time-out mask.
 *   nn = 1-d_{t+1}, tn = 1-td_{t+1}
 *   delta = fl(fl(r_t + fl(fl(fl(gamma*v_{t+1})*nn)*tn)) - v_t)
 *   A_t   = fl(delta + fl(fl(fl(gl*nn)*tn)*A_{t+1})),  ret_t = fl(A_t + v_t)
 * replaces: cleanrl/ppo.py:251-277.  Algorithmic HBM traffic: 24 B per env-step. */

/* The float-done recurrences of the reference's other two trainers, same buffers and launch shape:
 *   CATPPO_GAE_RL_GAMES  rl_games/cat_common.py:96-103 -> rl_games A2CBase.discount_values(fdones float):
 *       the CleanRL recurrence without the time-out channel (true_dones / next_true_done ignored, may
 *       be NULL); gamma_lambda = fl32(gamma*tau).  18 B per env-step.
 *   CATPPO_GAE_SKRL      skrl/ppo.py:397-442 compute_gae with `not_dones = 1 - dones`:
 *       A_t = fl(fl(r_t - v_t) + fl(fl(gamma*(1-d_t)) * fl(v_{t+1} + fl(lambda*A_{t+1}))));
 *       `dones` is indexed at t (not t+1), next_done / true_dones unused (may be NULL) and the
 *       `gamma_lambda` argument carries lambda itself.  18 B per env-step.
 * ONE entry since ABI 0.6: catppo_gae_planes(kind, mode, dtype, ...) - kind as above, mode CATPPO_GAE_SERIAL | _SCAN (below),
 * dtype CATPPO_F32 | CATPPO_F16 = element type of EVERY plane and row (fp16 rollout planes, BASELINE config 5: values are
 * widened to fp32, the recurrence runs in the same fp32 op order, advantages and returns = fl32(A + v) are rounded to half
 * (RNE) when stored; 12 B (CleanRL) / 10 B per env-step).  The scan mode exists for the CleanRL recurrence on fp32 planes. */
typedef enum { CATPPO_GAE_CLEANRL = 0, CATPPO_GAE_RL_GAMES = 1, CATPPO_GAE_SKRL = 2 } catppo_gae_kind;
enum { CATPPO_GAE_SERIAL = 0, CATPPO_GAE_SCAN = 1 };
int catppo_gae_planes(catppo_ctx* ctx, int kind, int mode, int dtype, const void* rewards, const void* values,
                      const void* dones, const void* true_dones, const void* next_value, const void* next_done,
                      const void* next_true_done, float gamma, float gamma_lambda, void* advantages, void* returns,
                      int T, int64_t N, void* stream);

/* skrl whole-batch advantage normalisation (skrl/ppo.py:436): out = (A - mean(A)) / (std_unbiased(A) + 1e-8),
 * moments in fp64 (deterministic); out may alias advantages; stats (NULL ok) receives {mean, std + 1e-8}. */
int catppo_adv_normalize(catppo_ctx* ctx, const float* advantages, int64_t n, float* out, float* stats,
                         void* stream);

/* rl_games `value_bootstrap` (rl_games/cat_common.py:59-64): rewards_i += fl(fl(gamma*values_i) * time_out_i),
 * in place, time_outs as uint8 {0,1}. */
int catppo_value_bootstrap(catppo_ctx* ctx, float* rewards, const float* values, const uint8_t* time_outs,
                           float gamma, int64_t N, void* stream);

/* ---- RunningMeanStd ----------------------------------------------------------------------
 * x [N,D] with leading dimension ldx.  mean/var [D], count [1] fp32 state.
 *   catppo_rms_moments : sums[0:D] = sum_i x, sums[D:2D] = sum_i x^2   (fp64, deterministic)
 *   catppo_rms_merge   : batch mean / biased var from (all-reduced) sums and total row
 *                        count n, then the Chan merge in the reference's fp32 op order
 *   catppo_rms_update  : moments + merge (single GPU)
 *   catppo_rms_normalize: out = (x-mean)/sqrt(var+eps)   (out may alias x)
 * replaces: cleanrl/ppo.py:12-62. */
int catppo_rms_merge(catppo_ctx* ctx, const double* sums, double n, int D, float* mean,
                     float* var, float* count, void* stream);

/* ---- actor-critic MLP ---------------------------------------------------------------------
 * Two independent MLPs (critic then actor, the reference's registration order) with L
 * hidden layers, ELU(alpha=1).  All parameters, gradients and Adam moments live in flat
 * fp32 buffers laid out by catppo_mlp_layout (segments 16-byte aligned, weight matrices
 * 128-byte aligned, first-layer rows padded from D to Dp = round_up(D,16) with zeros; the
 * padding between segments is never read as a parameter and never written by a gradient).
 * replaces: cleanrl/ppo.py:71-123 (Agent), :300-354 (minibatch update). */
#define CATPPO_MAX_HIDDEN 4

typedef struct catppo_mlp_shape {
  int32_t obs_dim;                  /* D  */
  int32_t act_dim;                  /* A  (<= 15) */
  int32_t n_hidden;                 /* L  (1..CATPPO_MAX_HIDDEN) */
  int32_t hidden[CATPPO_MAX_HIDDEN]; /* widths, each a multiple of 64 */
  int32_t mfma_bf16;                /* 0: fp32-input MFMA (reference numerics, default).  1: the hidden-layer GEMMs
                                       (forward, data gradient, weight gradient) round their operands to bf16
                                       (RNE) and use v_mfma_f32_32x32x16_bf16 with fp32 accumulation; parameters,
                                       gradients and the optimiser stay fp32 (BASELINE config 5).  Round 6: from 4096 rows
                                       up the hidden activations and dZ live in the workspace AS bf16 (rounded where they
                                       are produced instead of where a GEMM consumes them: same operands; elu' and the bias
                                       gradients see the rounded values) - CATPPO_ACT16=0 keeps them fp32-stored.
                                       2: split-bf16 ("bf16x3"): every operand is split x = hi + lo (two bf16 values,
                                       16 mantissa bits) and a.b = a_hi.b_hi + a_hi.b_lo + a_lo.b_hi on the bf16
                                       matrix pipe - meets the fp32 parity tolerances at 5x less matrix-pipe time;
                                       finer than the TF32 the reference enables (scripts/clean_rl/train.py:86-87). */
} catppo_mlp_shape;

typedef struct catppo_mlp_layout {
  int32_t obs_pad;                  /* Dp */
  int64_t n_flat;                   /* total floats in the flat buffer (incl. padding) */
  int64_t n_params;                 /* true parameter count (377,241 for the reference) */
  int64_t off_logstd;               /* [A] */
  /* per net (0 = critic, 1 = actor), per layer l = 0..L (L = output layer) */
  int64_t off_w[2][CATPPO_MAX_HIDDEN + 1]; /* [out_l, in_pad_l] row-major */
  int64_t off_b[2][CATPPO_MAX_HIDDEN + 1]; /* [out_l] */
  int32_t in_dim[CATPPO_MAX_HIDDEN + 1];   /* padded input width of layer l */
  int32_t out_dim[2][CATPPO_MAX_HIDDEN + 1];
} catppo_mlp_layout;

int catppo_mlp_layout_of(const catppo_mlp_shape* shape, catppo_mlp_layout* out);

/* workspace bytes needed by the MLP calls for a batch of `rows` samples */
uint64_t catppo_mlp_workspace_bytes(const catppo_mlp_shape* shape, int64_t rows);

/* Rollout policy step (no grad):  x [N,Dp] normalised obs ->
 *   action = mu + exp(logstd)*eps  (eps [N,A] supplied N(0,1) noise; NULL -> action = mu;
 *            given_action [N,A] non-NULL -> that action is scored instead, ppo.py:109-113)
 *   logprob [N], value [N].   replaces: ppo.py:104-119,208-212. */

/* critic only (bootstrap value, ppo.py:252) */

typedef struct catppo_ppo_hparams {
  float clip_coef, ent_coef, vf_coef;
  int32_t norm_adv, clip_vloss;
  float inv_global_batch;   /* 1 / (minibatch size summed over all ranks) */
  /* advantage normalisation statistics: if adv_stats_external != 0 the kernel reads
   * {mean, 1/(std+1e-8)} from adv_stats (device, 2 floats; env-sharded exact mode),
   * otherwise it computes them over this minibatch. */
  int32_t adv_stats_external;
} catppo_ppo_hparams;

/* One minibatch: gather rows `mb_inds` of the flattened rollout buffers, forward both nets,
 * PPO losses, backward.  Writes the flat gradient (same layout as params) and 8 diagnostics
 * {pg_loss, v_loss, entropy, loss, approx_kl, old_approx_kl, clipfrac, 0} which are
 * ACCUMULATED into diag[8] (zeroed by the caller once per iteration).
 *   b_obs [B,Dp], b_actions [B,A], b_logprobs/b_advantages/b_returns_n/b_values_n [B]
 *   mb_inds [M] int64;   value_rms {mean,var} device scalars for newvalue normalisation.
 * replaces: ppo.py:298-352. */
int catppo_ppo_minibatch_grad(catppo_ctx* ctx, const catppo_mlp_shape* shape,
                              const catppo_ppo_hparams* hp, const float* params,
                              const float* b_obs, const float* b_actions,
                              const float* b_logprobs, const float* b_advantages,
                              const float* b_returns_n, const float* b_values_n,
                              const int64_t* mb_inds, int64_t M, const float* vrms_mean,
                              const float* vrms_var, const float* adv_stats, float* grad,
                              float* diag, void* stream);

/* Epoch-level gather: ONE launch copies the samples of a whole permutation into packed buffers so that minibatch
 * m (M samples, the last one possibly shorter) is the contiguous slice
 *   x_g[m*M*Dp ..], act_g[m*M*A ..], scal_g[4*m*M ..] = {old log-prob, advantage, returns_n, values_n}[M_m],
 *   adv_part_g[m*CATPPO_GATHER_PARTS(M)*2 ..] = fp64 {sum, sum of squares} of the advantages per 64-row chunk,
 * and catppo_ppo_minibatch_grad_packed runs forward / losses / backward on such a slice without gathering again
 * (5 gathers per iteration instead of 30).  Same arithmetic and results as catppo_ppo_minibatch_grad. */
#define CATPPO_GATHER_ROWS 64
#define CATPPO_GATHER_PARTS(M) (((M) + CATPPO_GATHER_ROWS - 1) / CATPPO_GATHER_ROWS)
int catppo_ppo_minibatch_grad_packed(catppo_ctx* ctx, const catppo_mlp_shape* shape, const catppo_ppo_hparams* hp,
                                     const float* params, const float* x_mb, const float* act_mb,
                                     const float* scal_mb, const double* adv_part_mb, int64_t M,
                                     const float* vrms_mean, const float* vrms_var, const float* adv_stats,
                                     float* grad, float* diag, void* stream);

/* Global-norm clip + Adam on the flat buffers (after the gradient all-reduce if sharded).
 *   g <- g * min(1, max_norm/(||g||+1e-6));  Adam(beta1,beta2,eps), bias-corrected, `step`
 *   counted from 1.   replaces: ppo.py:353-354 (clip_grad_norm_, optim.Adam.step). */
int catppo_clip_adam(catppo_ctx* ctx, float* params, float* grad, float* exp_avg,
                     float* exp_avg_sq, int64_t n_flat, float max_grad_norm, double lr,
                     double beta1, double beta2, double eps, int64_t step, void* stream);

/* ---- device-resident iteration state --------------------------------------------------------------------
 * Everything that changes from one optimiser step / rollout step / iteration to the next and that a kernel needs
 * (learning rate, Adam step count, RNG counters) lives in ONE small device struct instead of kernel arguments, so
 * that the launches of an iteration are identical from one iteration to the next and can be replayed from a
 * hipGraph (catppo_graph_*), and so that a KL-adaptive schedule can change the learning rate without a host
 * round trip.  The struct is caller-owned device memory (sizeof(catppo_iter_state) bytes); only the library's
 * kernels write it. */
typedef struct catppo_iter_state {
  uint64_t seed;      /* key of the counter-based RNG (Philox4x32-10 action noise, minibatch permutation) */
  int64_t iteration;  /* 1-based PPO iteration, incremented by catppo_iter_begin */
  int64_t adam_step;  /* optimiser steps taken, incremented by catppo_clip_adam_dev */
  double lr;          /* learning rate used by catppo_clip_adam_dev */
  double kl_mark;     /* diag[4] (approx-KL sum) at the last catppo_kl_adaptive_lr of this iteration */
  double n_mark;      /* diag[7] (minibatch count) at that point */
  double last_kl;     /* KL the schedule last saw (diagnostics) */
  float adam_step_size;  /* lr / (1 - beta1^step) of the step catppo_clip_adam_dev is taking (written by its first launch, */
  float adam_bc2_sqrt;   /* sqrt(1 - beta2^step)                      read by its second; scratch, not an input)     */
} catppo_iter_state;

/* state <- {seed, iteration 0, adam_step 0, lr} */
int catppo_iter_init(catppo_ctx* ctx, catppo_iter_state* state, uint64_t seed, double lr, void* stream);
/* start of an iteration: iteration += 1, KL marks <- 0, and the learning-rate schedule
 *   CATPPO_LR_FIXED   lr = lr0
 *   CATPPO_LR_LINEAR  lr = (1 - (iteration-1)/num_iterations) * lr0   in double, the reference's expression
 *                     (cleanrl/ppo.py:196-199)
 *   CATPPO_LR_KEEP    lr unchanged (it is driven by catppo_kl_adaptive_lr) */
enum { CATPPO_LR_FIXED = 0, CATPPO_LR_LINEAR = 1, CATPPO_LR_KEEP = 2 };
int catppo_iter_begin(catppo_ctx* ctx, catppo_iter_state* state, double lr0, int64_t num_iterations, int schedule,
                      void* stream);

/* catppo_clip_adam with lr and the step count taken from (and the step count advanced in) the device state.
 * Same arithmetic: the bias corrections 1-beta^step are evaluated in double on the device. */
int catppo_clip_adam_dev(catppo_ctx* ctx, float* params, float* grad, float* exp_avg, float* exp_avg_sq,
                         int64_t n_flat, float max_grad_norm, double beta1, double beta2, double eps,
                         catppo_iter_state* state, void* stream);

/* ABI 0.4: one optimiser step of a single process in ONE call = catppo_ppo_minibatch_grad_packed followed by
 * catppo_clip_adam_dev (same arguments, same results up to the summation order of the fp64 squared norm), for callers
 * with nothing to do between the two - no gradient all-reduce.  The workgroups that fold the split-K partials into the
 * flat gradient also emit the sum of squares of what they write, so the clip needs no launch that re-reads the
 * gradient: one launch fewer per step (7 instead of 8 at 16384 x 3 x 256).
 * replaces: cleanrl/ppo.py:298-356 (minibatch losses, backward, clip_grad_norm_, optimizer.step) of a non-distributed run. */
int catppo_ppo_minibatch_step_packed(catppo_ctx* ctx, const catppo_mlp_shape* shape, const catppo_ppo_hparams* hp,
                                     float* params, const float* x_mb, const float* act_mb, const float* scal_mb,
                                     const double* adv_part_mb, int64_t M, const float* vrms_mean,
                                     const float* vrms_var, const float* adv_stats, float* grad, float* diag,
                                     float* exp_avg, float* exp_avg_sq, float max_grad_norm, double beta1, double beta2,
                                     double eps, catppo_iter_state* state, void* stream);

/* KL-adaptive learning rate, device side (no host sync).  Two calls so that an env-sharded run can put its
 * all-reduce between them:
 *   catppo_kl_mean        kl_out[0] = (diag[4] - kl_mark) / (diag[7] - n_mark): the mean approx-KL of the minibatches
 *                         processed since the previous call of this iteration (the marks then advance).  With
 *                         inv_global_batch = 1/(M*world) in the minibatch calls the per-rank values SUM to the
 *                         global mean, which is what skrl's all_reduce(kl, SUM)/world_size computes.
 *   catppo_kl_adaptive_lr kl > threshold*kl_factor : lr = max(lr / lr_factor, min_lr)
 *                         kl < threshold/kl_factor : lr = min(lr * lr_factor, max_lr)
 * replaces: skrl/ppo.py:558-567 (scheduler.step(kl) after the KL all-reduce) with skrl's published KLAdaptiveLR rule
 * (kl_factor 2, lr_factor 1.5, min_lr 1e-6, max_lr 1e-2; threshold skrl_ppo_cfg.yaml:49-51 = 0.01) - rl_games'
 * `lr_schedule: adaptive` (rl_games_cat_solo.yaml:64-66, kl_threshold 0.008) is the same rule.  Neither library
 * is vendored in the reference: the rule itself is PARITY UNPINNED, the call site and the reduction are pinned. */
int catppo_kl_mean(catppo_ctx* ctx, catppo_iter_state* state, const float* diag, float* kl_out, void* stream);
int catppo_kl_adaptive_lr(catppo_ctx* ctx, catppo_iter_state* state, const float* kl, double kl_threshold,
                          double kl_factor, double lr_factor, double min_lr, double max_lr, void* stream);

/* ---- rollout forward: policy step + value (ONE entry since ABI 0.6) ------------------------------------------------
 * catppo_policy_step: both networks' forward on x [N, Dp] + the Gaussian head (replaces Agent.get_action_and_value /
 * get_value, cleanrl/ppo.py:104-119,186-189; the bootstrap value of :251 is the critic-only form, action == NULL).
 * Where the action comes from - exactly one of:
 *   eps          supplied N(0,1) noise [N, A]: a = mu + sigma * eps (Normal.sample(), cleanrl/ppo.py:111)
 *   state, step  on-device noise: Philox4x32-10 + Box-Muller inside the head kernel; element (env i, dim k) of rollout
 *                step `step` of iteration state->iteration uses counter {i, k/4, step, iteration}, key = state->seed,
 *                lane k%4 of the block; eps_out ([N, A], may be NULL) receives it, so a parity test can replay it
 *   given_action evaluate these actions (log-prob / value of stored actions)
 * value_dtype: CATPPO_F32, or CATPPO_F16 to store `value` as IEEE half (fp16 rollout planes).
 * Kernels by batch size (catppo_plan_log names them): <= 2048 rows step16_fwd_kernel (16-row tiles), 2049-4096 the 32-row
 * row-resident kernels, else layer-wise GEMM launches + head_act_kernel. */
int catppo_policy_step(catppo_ctx* ctx, const catppo_mlp_shape* shape, const float* params, const float* x, int64_t N,
                       const float* eps, const float* given_action, const catppo_iter_state* state, int32_t step,
                       float* eps_out, float* action, float* logprob, void* value, int value_dtype, void* stream);

/* The epoch gather (one launch for every minibatch of an epoch: minibatch m = contiguous slice m of the packed buffers)
 * with (a) either an index array `inds` (state == NULL) or a permutation computed on the fly
 * (inds == NULL; replaces torch.randperm, cleanrl/ppo.py:295): sample j of the epoch reads row P(j), P = a keyed
 * bijection of [0,total) (6-round Feistel network on the next power of four with cycle walking, round keys from
 * Philox(seed; iteration, epoch)) - no index array, no sort; inds_out ([total] int64, may be NULL) receives P for
 * parity tests - and (b) adv_dtype: element type of b_advantages (CATPPO_F16 for fp16 rollout planes; widened on
 * the way into the packed buffers). */
int catppo_ppo_gather_ex(catppo_ctx* ctx, const catppo_mlp_shape* shape, const float* b_obs, const float* b_actions,
                         const float* b_logprobs, const void* b_advantages, int adv_dtype, const float* b_returns_n,
                         const float* b_values_n, const int64_t* inds, const catppo_iter_state* state, int32_t epoch,
                         int64_t total, int64_t M, float* x_g, float* act_g, float* scal_g, double* adv_part_g,
                         int64_t* inds_out, void* stream);
/* ABI 0.5: the moments of ALL `n_epochs` keyed permutations of an iteration (catppo_ppo_gather_ex with `state`) in one
 * call, before the first gather: moments[(e * n_mb + m)] = {sum, sum of squares, rows} of minibatch m of epoch e, so an
 * env-sharded run exchanges the advantage statistics of an iteration with ONE all-reduce instead of one per epoch.
 * `parts_scratch`: n_epochs * n_mb * ceil(minibatch / 64) * 2 doubles.  Chunking and summation order are the
 * gather's: bit-identical to catppo_adv_moments_parts on the partials of the epoch's own gather.  `advantages`: the
 * (total,) plane, CATPPO_F32 or CATPPO_F16.   replaces: cleanrl/ppo.py:314-318 across ranks and epochs. */
int catppo_adv_moments_keyed(catppo_ctx* ctx, const void* advantages, int adv_dtype, const catppo_iter_state* state,
                             int32_t n_epochs, int64_t total, int64_t minibatch, double* parts_scratch,
                             double* moments, void* stream);

/* ---- fp16 rollout planes (BASELINE config 5) ------------------------------------------------------------------
 * catppo_rollout_store_ex: catppo_rollout_store with the three destination rows in `dtype`.
 * catppo_rms_moments_ex / _update_ex / _normalize_ex: RunningMeanStd over an input of `x_dtype` (widened exactly;
 * statistics, state and output stay fp32). */
int catppo_rollout_store_ex(catppo_ctx* ctx, const float* reward, const float* dones, const uint8_t* time_outs,
                            void* rewards_t, void* dones_t1, void* true_dones_t1, int dtype, int64_t N, void* stream);
int catppo_rms_moments_ex(catppo_ctx* ctx, const void* x, int x_dtype, int64_t N, int D, int64_t ldx, double* sums,
                          void* stream);
int catppo_rms_update_ex(catppo_ctx* ctx, const void* x, int x_dtype, int64_t N, int D, int64_t ldx, float* mean,
                         float* var, float* count, void* stream);
int catppo_rms_normalize_ex(catppo_ctx* ctx, const void* x, int x_dtype, int64_t N, int D, int64_t ldx,
                            const float* mean, const float* var, float eps, float* out, int64_t ldo, void* stream);

/* ---- GAE scan mode -----------------------------------------------------------------------------------------------
 * mode CATPPO_GAE_SERIAL : one lane per env walks t = T-1..0 (catppo_gae_ex; bit-exact with the reference loop).
 * mode CATPPO_GAE_SCAN   : the time axis of an env is split over the 2^k lanes of a lane group; every lane composes
 *   the affine maps A -> delta_t + c_t * A of its chunk, the group combines them with a wavefront-shuffle suffix
 *   scan, then every lane replays its chunk from the incoming value.  Fills the chip at small N (4096 envs: 64 waves
 *   serial, up to 1024 in scan mode); the composition reassociates the recurrence, so results agree with the serial
 *   mode to ~1e-6 relative (tests hold 1e-5), not bit for bit.  CleanRL recurrence (kind 0), fp32 planes. */
/* ---- fused rollout step ------------------------------------------------------------------------------------------
 * Everything between the simulator's state update and the next policy forward, in TWO launches instead of ten:
 *   launch 1 (16-env tiles)  process_action, episode counters, terminations, reward pick-up (catppo_env_pre_step);
 *                            all constraint terms -> cstr (catppo_cat_terms); per-workgroup column maxima and fp64
 *                            observation moments; the last workgroup to finish folds them (fixed order) into
 *                            xchg = {colmax[K] floored at 1e-6 | sum x [D] | sum x^2 [D]}
 *   [env-sharded: MAX / SUM all-reduce of xchg here]
 *   launch 2 (32-env tiles)  running-max EMA, probabilities, per-env / per-term maxima, episode statistics, reward
 *                            scaling, float dones (catppo_cat_apply); ConstraintManager.reset statistics of the envs
 *                            that reset + zeroing (catppo_cat_reset), episode_length / action history reset;
 *                            rollout-buffer rows (catppo_rollout_store); Chan merge of the observation normaliser
 *                            and normalised observation row obs[step+1] (catppo_rms_update + _normalize).  Every
 *                            workgroup derives the new running maxima / normaliser state redundantly from xchg; the
 *                            last one to finish writes them back and folds the reset statistics into `log_out`.
 * Same arithmetic and results as the separate calls (bit-exact: CaT, rewards, dones, statistics; normaliser: same
 * fp64 sums folded in a different fixed order, <= 1 ulp of the fp32 state). */
typedef struct catppo_rollout_step {
  /* sizes */
  int64_t N;
  int32_t A, D, K, n_terms;
  /* pre-step (catppo_env_pre_step) */
  const float* action_in; float* action; float* prev_action;
  int64_t* episode_length; int64_t max_episode_length;
  const float* hard_reset; int64_t hard_reset_stride;
  const float* reward_src; int64_t reward_stride;
  uint8_t* time_outs; uint8_t* terminated; uint8_t* reset; float* reward;
  /* constraint terms (catppo_cat_terms) */
  const catppo_term_desc* desc;       /* HOST array [n_terms] */
  const float* forces; int64_t forces_env_stride; int32_t H, B;
  const float* command; int32_t command_ld;
  float* cstr;
  /* CaT (catppo_cat_apply) */
  const int32_t* term_off; const float* term_dp;   /* HOST arrays */
  float min_p, tau, one_minus_tau; int32_t first_call;
  float* rm; float* cstr_prob; float* dones; float* ep_viol; float* ep_prob; float* probs;
  /* ConstraintManager.reset statistics: log_out[2*n_terms] <- means over the envs that reset (log_prev if none) */
  const float* log_prev; float* log_out;
  int32_t zero_action_on_reset;
  /* rollout rows */
  void* rewards_t; void* dones_t1; void* true_dones_t1; int32_t plane_dtype;
  /* observation normaliser + next observation row */
  const float* obs_raw; int64_t obs_ld;
  float* obs_mean; float* obs_var; float* obs_count; float obs_eps; double obs_rows_total;
  float* obs_out; int64_t obs_out_ld;
  /* exchange buffer, device: K floats (as doubles' storage is separate) - see catppo_rollout_xchg_layout */
  void* xchg;
  /* env-sharded, ONE collective per env step (ABI 0.3): catppo_rollout_pre writes this rank's record into `xchg`; the
   * caller all-gathers the records of all ranks (catppo_allgather, `bytes` of catppo_rollout_xchg_layout each, rank order) into
   * `xchg_gathered`, and catppo_rollout_post folds them itself - column maxima by MAX (exact), moment sums in rank order
   * (identical on every rank).  xchg_records = number of records (0 / 1: read `xchg`, e.g. after MAX / SUM all-reduces). */
  const void* xchg_gathered;
  int32_t xchg_records;
  /* ABI 0.4: the simulator's state advance inside catppo_rollout_pre.  A simulator whose new state already exists as
   * one contiguous [N, sim_row_bytes] block somewhere else (here: the synthetic stream's next slab; the reference's
   * `scene.update`, cat/cat_env.py:60-90, is that copy) hands it over as `sim_src`: every input of THIS call that points
   * into the state block [sim_state, sim_state + N * sim_row_bytes) - term tensors, forces, command, hard_reset,
   * reward_src, obs_raw - is read from `sim_src` at the same offset, and the launch copies the rows to `sim_state`
   * (each workgroup its own envs), so that after it every view of the state block is current (catppo_rollout_post and
   * everybody else read it as before).  One launch and a 5 us copy kernel less per env step.  sim_src == NULL: the
   * caller has updated the state block itself.  sim_row_bytes % 16 == 0. */
  const void* sim_src; void* sim_state; int64_t sim_row_bytes;
} catppo_rollout_step;
/* bytes of one exchange record and the byte offset of its fp64 sums (K constraint columns, D observation width) */
int catppo_rollout_xchg_layout(int K, int D, uint64_t* bytes, uint64_t* sum_offset);
uint64_t catppo_rollout_step_sizeof(void);   /* sizeof(catppo_rollout_step): lets a binding check its struct layout */
/* phase 1 (two launches: the per-tile kernel and the fold of its partial rows into `xchg`) and phase 2 (one launch); call
 * both back to back on one GPU; all-reduce xchg in between when env-sharded:
 * floats [0,K) with MAX, doubles at byte offset `sum_offset` of catppo_rollout_xchg_layout [2*D] with SUM) */
int catppo_rollout_pre(catppo_ctx* ctx, const catppo_rollout_step* a, void* stream);
int catppo_rollout_post(catppo_ctx* ctx, const catppo_rollout_step* a, void* stream);
/* ABI 0.5: the tail of catppo_rollout_post - ONE workgroup that, after all others, publishes the new running maxima (rm)
 * and observation-normaliser state (obs_mean / obs_var / obs_count) and folds the reset statistics into log_out - off the
 * chain forward -> pre -> fold -> post -> forward of an env step.  With on = 1 a post launch ends without that tail (and
 * without the "last workgroup arrives" hand-shake in front of it); the tail runs as one more workgroup of the NEXT
 * catppo_rollout_pre launch on the same stream (nothing in between reads what it writes: the policy forward consumes
 * obs_out, the simulator its own state), or as a launch of its own from catppo_rollout_defer_tail(ctx, -1, stream) (flush,
 * the mode stays) / the next catppo_rollout_post / catppo_rollout_defer_tail(ctx, 0, stream).  CONTRACT while on: rm, obs_mean, obs_var, obs_count
 * and log_out of a step are valid (in stream order) only after one of those calls; every other output of the step is
 * valid after catppo_rollout_post as before.  Values are bit-identical either way: the tail derives the state with the same
 * device functions from the same exchange record(s) - which must stay untouched until then (they are: the next writer is
 * the fold behind the next catppo_rollout_pre / the next all-gather).  The rollout loop of cleanrl/ppo.py's PPOTrainer
 * switches it on around its env steps and off (= flush) before GAE.  One stream per context while it is on: the pending
 * tail and its reset-statistics rows belong to the context; a call that flushes the tail from ANOTHER stream is ordered
 * behind the tail's launch by an event (round 6).  Reference: the statistics concerned are
 * ConstraintManager's running maxima (cat/constraint_manager.py:58-61), RunningMeanStd's state (cleanrl/ppo.py:48-62)
 * and the episode log of ConstraintManager.reset (cat/constraint_manager.py:190-211). */
int catppo_rollout_defer_tail(catppo_ctx* ctx, int on, void* stream);    /* on: 1 | 0 (= off + flush) | -1 (flush only) */

/* ---- rl_games front end: episode bookkeeping with float dones ---------------------------------------
 * One env step of CaTA2CAgent.play_steps' bookkeeping (rl_games/cat_common.py:71-92), one launch, no host sync:
 *   current_rewards += rewards; current_shaped_rewards += shaped_rewards; current_lengths += 1
 *   done_i = dones_i >= 1.0;  the three AverageMeters are updated with current_*[done]
 *   current_rewards *= (1 - dones); current_shaped_rewards *= (1 - dones)   (FLOAT not_dones);  current_lengths[done] = 0
 * rewards / shaped_rewards / current_*rewards: [N, value_size]; dones, current_lengths: [N] (fp32, rl_games keeps the
 * lengths in fp32); done_mask_out: optional [N] bytes (what `dones.ge(1.0)` was), for an observer that wants indices.
 * catppo_rlg_meters (device memory, initialised by catppo_rlg_meters_init) restates rl_games' torch_ext.AverageMeter
 * (rl_games 1.6.1 is not vendored in the reference tree: published rule, parity unpinned against rl_games itself):
 *   update(values): n = rows; if n == 0 return; new_mean = mean(values); size = min(n, max_size);
 *                   old = min(max_size - size, current_size); current_size = old + size;
 *                   mean = (mean * old + new_mean * size) / (old + size)                                            */
#define CATPPO_RLG_MAX_VALUE_SIZE 4
typedef struct catppo_rlg_meters {
  float mean_rewards[CATPPO_RLG_MAX_VALUE_SIZE];
  float mean_shaped_rewards[CATPPO_RLG_MAX_VALUE_SIZE];
  float mean_lengths;
  int32_t size_rewards, size_shaped, size_lengths;   /* AverageMeter.current_size */
  int32_t max_size;                                   /* games_to_track */
  int32_t last_done_count;                            /* episodes that ended in the latest step */
} catppo_rlg_meters;
int catppo_rlg_meters_init(catppo_ctx* ctx, catppo_rlg_meters* meters, int max_size, void* stream);
int catppo_rlg_episode_step(catppo_ctx* ctx, const float* rewards, const float* shaped_rewards, const float* dones,
                            int value_size, float* current_rewards, float* current_shaped_rewards,
                            float* current_lengths, int64_t N, catppo_rlg_meters* meters, uint8_t* done_mask_out,
                            void* stream);

/* ---- HIP graphs ------------------------------------------------------------------------------------------------
 * Capture every launch the library (or anything else) enqueues on `stream` between begin and end into a hipGraph,
 * instantiate it, and replay it with one call.  `stream` must not be the legacy default stream.  No library call
 * allocates or synchronises while a capture is active (catppo_reserve returns CATPPO_E_ARG then). */
int catppo_graph_begin(catppo_ctx* ctx, void* stream);
/* graph_id == NULL: end the capture WITHOUT keeping its graph (error path: something between begin and end failed, the
 * partial graph must never be replayed); a no-op when no capture is active */
int catppo_graph_end(catppo_ctx* ctx, void* stream, int* graph_id, int* n_nodes);
int catppo_graph_launch(catppo_ctx* ctx, int graph_id, void* stream);
int catppo_graph_destroy(catppo_ctx* ctx, int graph_id);
/* ---- collectives (RCCL over xGMI, one process per GPU) ---------------------------------------------------------
 * librccl is loaded at run time (dlopen) by the first of these calls; the library has no link-time dependency on it.
 * catppo_comm_unique_id: rank 0 creates the id (128 bytes) and ships it to the other ranks by any means (the host
 * code here uses the launcher's rendezvous store); every rank then calls catppo_comm_init.  The collectives are
 * in place, enqueue on `stream`, are capturable in a hipGraph, and reduce `count` elements of `dtype`
 * (CATPPO_F32 / CATPPO_F64).   semantics replaced: skrl/ppo.py:126-131 (parameter broadcast), :534-537 (gradient
 * reduction), :562-564 (KL all-reduce); the CleanRL path of the reference has no collective. */
enum { CATPPO_SUM = 0, CATPPO_MAX = 1 };
#define CATPPO_UNIQUE_ID_BYTES 128
/* ctx == NULL: can this process use the collectives at all (librccl loadable, every entry point present: CATPPO_OK /
 * CATPPO_E_COMM)?  No communicator, no bootstrap thread: what every rank checks BEFORE anybody enters the blocking
 * catppo_comm_init.  ctx != NULL: the world size of its communicator (>= 1), 0 = none. */
int catppo_comm_probe(catppo_ctx* ctx);
int catppo_comm_unique_id(uint8_t* out128);
int catppo_comm_init(catppo_ctx* ctx, int rank, int world, const uint8_t* unique_id128);
int catppo_comm_destroy(catppo_ctx* ctx);
int catppo_allreduce(catppo_ctx* ctx, void* buf, int64_t count, int dtype, int op, void* stream);
int catppo_broadcast(catppo_ctx* ctx, void* buf, int64_t count, int dtype, int root, void* stream);
/* recv[rank * bytes ...] = send of that rank, for every rank (ncclAllGather of raw bytes; send != recv) */
int catppo_allgather(catppo_ctx* ctx, const void* send, void* recv, int64_t bytes, void* stream);

/* ---- ABI 0.4: gradient all-reduce in per-layer buckets, overlapped with the backward pass ------------------------
 * semantics replaced: skrl/ppo.py:534-537 (every gradient reduced after backward()).  With the switch on AND a
 * communicator present, catppo_ppo_minibatch_grad / _packed fold the partials of hidden layer l (and, with the last
 * hidden layer, the heads and log-std) as soon as that layer's weight-gradient launch is enqueued - on the context's
 * side stream - and SUM-all-reduce that bucket of the flat gradient there, while the launches of the layers below
 * still run on `stream`; the side stream joins `stream` before the call returns its stream order, so `grad` is the
 * GLOBAL sum when the next operation on `stream` (catppo_clip_adam*) reads it and the caller must NOT all-reduce it
 * again.  Sums per element are those of the single fold launch (bit-identical on a world of one); the whole sequence
 * is capturable.  catppo_set_grad_overlap(ctx, -1) queries: 1 when the next gradient call will reduce the gradient itself. */
/* ABI 0.5: on == 2 selects the "tail" form - NO extra launch: once the launch that folds every layer above the first is
 * enqueued (dw_fold_kernel), those ranges of the flat gradient are all-reduced on the side stream under the final fold
 * launch of the first layer's own partials, which are reduced on the caller's stream behind a join.  Measured on a world of
 * one: see profiles/r5_grad_overlap_tail_world1.txt. */
int catppo_set_grad_overlap(catppo_ctx* ctx, int on);      /* on: 0 | 1 | 2 set (CATPPO_OK), -1 query (0 / 1) */

/* ---- test hook (ABI 0.6) ------------------------------------------------------------------------------------------
 * buf != NULL ([2][M] int32, device): the head / loss kernel of every following catppo_ppo_minibatch_* call writes the clip
 * branch each sample of the minibatch took - surrogate codes in [0, M) (ratio against 1 +- clip_coef), value-loss codes in
 * [M, 2 M) (newvalue - old value against +- clip_coef): 0 inside, 1 below, 2 above; cleanrl/ppo.py:320-341.  NULL: off.
 * What tests/test_gpu_parity_sizes.py compares with the oracle's branches when two parameter trajectories part. */
int catppo_debug_clip_branches(catppo_ctx* ctx, int32_t* buf);

#ifdef __cplusplus
}
#endif
#endif /* CATPPO_H */
