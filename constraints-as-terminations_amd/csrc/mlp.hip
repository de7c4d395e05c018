// Actor-critic MLP on MI355X: rollout policy step, PPO minibatch forward/loss/backward.
//
// Replaces cleanrl/ppo.py:71-123 (Agent) and :298-352 (minibatch update up to backward()).
// Hidden layers run on the fp32 MFMA GEMM of gemm_f32.h (critic + actor grouped in one launch,
// bias+ELU / ELU' fused in the epilogue).  The A-wide / 1-wide output heads, the Gaussian
// log-prob / entropy, the clipped PPO losses AND their analytic gradients w.r.t. the last
// hidden activations are one wave-per-row VALU kernel (head_loss) - or, from 4096 rows up and a
// last layer of 128 / 256 columns, the epilogue of the last forward GEMM itself (fwd_head_kernel:
// the activated tile stays in LDS, the head products run on the MFMA): no autograd graph, no
// (M,12) temporaries, no host sync.  Weight gradients are split-K over the batch with
// deterministic two-stage reduction (no float atomics => run-to-run reproducible); a layer's
// weight-gradient and data-gradient GEMMs share one launch (gemm_pair_kernel), the minibatch is
// gathered once per epoch (catppo_ppo_gather) and every partial is folded by one launch.
// Round 4: the hidden layers below the last one of a 256-wide network run as ONE row-resident launch
// (rows_fwd_kernel, fwd_rows.h) in the update phase, the same kernel with heads is the rollout forward, and the
// first layer's weight-gradient launch carries the fold of the layers above it (dw_fold_kernel).
#include "common.h"
#include "gemm_f32.h"
#include "rng.h"

#include <cmath>
#include <cstdlib>
#include <type_traits>

namespace {
#include "mlp_common.h"
#include "mlp_forward.h"
#include "mlp_loss.h"
#include "step16.h"
#include "mlp_backward.h"
#include "mlp_optim.h"

template <typename F>
int dispatch_cpl(int hl, F&& f) {
  switch (hl) {
    case 64: f(std::integral_constant<int, 1>{}); return 0;
    case 128: f(std::integral_constant<int, 2>{}); return 0;
    case 256: f(std::integral_constant<int, 4>{}); return 0;
    case 512: f(std::integral_constant<int, 8>{}); return 0;
    default: return -1;
  }
}

}  // namespace

// =============================================================================== C ABI
extern "C" int catppo_mlp_layout_of(const catppo_mlp_shape* shape, catppo_mlp_layout* out) {
  return layout_of(shape, out);
}

extern "C" uint64_t catppo_mlp_workspace_bytes(const catppo_mlp_shape* shape, int64_t rows) {
  catppo_mlp_layout L;
  if (layout_of(shape, &L) != CATPPO_OK || rows < 1) return 0;
  MlpWs w{};
  carve(shape, L, rows, true, nullptr, 0, &w);
  return w.bytes + 4096;
}

static int mlp_prologue(catppo_ctx* ctx, const catppo_mlp_shape* shape, int64_t M, bool training,
                        catppo_mlp_layout* L, MlpWs* w, const char* fn) {
  if (!ctx) return CATPPO_E_ARG;
  if (layout_of(shape, L) != CATPPO_OK) return catppo_fail(ctx, CATPPO_E_ARG, "%s: unsupported MLP shape", fn);
  if (M < 1 || M > (int64_t(1) << 30)) return catppo_fail(ctx, CATPPO_E_ARG, "%s: bad row count", fn);
  if (!carve(shape, *L, M, training, static_cast<char*>(ctx->ws), ctx->ws_bytes, w))
    return catppo_fail(ctx, CATPPO_E_WORKSPACE, "%s: workspace too small (%llu < %llu B); call catppo_reserve", fn,
                       (unsigned long long)ctx->ws_bytes, (unsigned long long)w->bytes);
  return CATPPO_OK;
}

namespace {
// fused_fwd_kernel applies when: fp32 MFMA, every hidden width a multiple of 128 (a wave owns 32 columns of a 256 /
// 128-column chunk), a head width the head code knows, the two activation tiles + the weight rings fit the LDS, and the
// batch is in the window where one 32-row workgroup per CU (x 2 networks) beats the layer-wise launches: 2049-4096
// rows (measured: 4096 rows -0.2 ms per 24-step rollout, 2048 rows equal, below that the workgroup's ~35 us serial
// time loses to the launch-bound small GEMMs).  CATPPO_FUSED_FWD=0 disables it (A/B), CATPPO_FUSED_FWD_MIN_ROWS /
// _MAX_ROWS move the window (the tests pin it open to cover small and ragged batches).
bool fused_fwd_plan(const catppo_mlp_shape* sh, const catppo_mlp_layout& L, int64_t rows, FusedFwdArgs* fa, size_t* lds) {
  static const int enabled = env_int("CATPPO_FUSED_FWD", 1);
  static const int max_rows = env_int("CATPPO_FUSED_FWD_MAX_ROWS", 4096);
  static const int min_rows = env_int("CATPPO_FUSED_FWD_MIN_ROWS", 2049);
  if (!enabled || rows > max_rows || rows < min_rows || sh->mfma_bf16 != 0) return false;
  const int nl = sh->n_hidden;
  const int hl = sh->hidden[nl - 1];
  if (hl != 128 && hl != 256 && hl != 512) return false;
  int w0 = L.obs_pad, w1 = 0;
  for (int l = 0; l < nl; ++l) {
    if (sh->hidden[l] % 128 != 0) return false;
    int& dst = (l % 2 == 0) ? w1 : w0;       // layer l writes act1 for even l, act0 for odd l
    dst = dst > sh->hidden[l] ? dst : sh->hidden[l];
  }
  fa->Dp = L.obs_pad, fa->n_hidden = nl;
  fa->ld0 = w0 + 4, fa->ld1 = w1 + 4;
  for (int l = 0; l < nl; ++l) fa->hidden[l] = sh->hidden[l];
  for (int net = 0; net < 2; ++net)
    for (int l = 0; l <= nl; ++l) fa->off_w[net][l] = L.off_w[net][l], fa->off_b[net][l] = L.off_b[net][l];
  *lds = sizeof(float) * ((size_t)kFR * (fa->ld0 + fa->ld1) + kFRing);
  // fused_chunk's pipeline deliberately runs past the end of a contraction: its last iterations stage up to three 16-k
  // weight slabs beyond the last row of a chunk (never multiplied) and read A fragments past K in the LDS tile.  Both
  // are in bounds only because of how the buffers are laid out - checked here instead of assumed: (1) every hidden
  // weight matrix is followed by at least 64 more floats of the flat parameter buffer (its bias, the next layer),
  // (2) the two activation tiles are followed by the weight rings inside the same dynamic-LDS allocation (act0's
  // overrun lands in act1, act1's in the rings: >= 64 floats each).  A layout that breaks either takes the layer-wise path.
  for (int net = 0; net < 2; ++net)
    for (int l = 0; l < nl; ++l)
      if (L.off_w[net][l] + (int64_t)sh->hidden[l] * L.in_dim[l] + 64 > L.n_flat) return false;
  static_assert(kFRing >= 64, "the weight rings double as the over-read margin of the activation tiles");
  if ((size_t)kFR * fa->ld1 < 64) return false;
  return *lds <= 160 * 1024;
}

// rows_fwd_kernel<R> applies when: fp32 MFMA, every layer it computes is 256 wide (eight waves x 32 columns, outputs of a
// layer held in accumulators until the tile may be overwritten), the tile + rings fit the LDS.  `n_layers` hidden layers
// are computed (training: all but the last; rollout: all).
bool rows_fwd_plan(const catppo_mlp_shape* sh, const catppo_mlp_layout& L, int n_layers, int R, FusedFwdArgs* fa,
                   size_t* lds) {
  if (sh->mfma_bf16 != 0 || n_layers < 1 || n_layers > sh->n_hidden) return false;
  for (int l = 0; l < n_layers; ++l)
    if (sh->hidden[l] != rowsfwd::kWidth) return false;
  const int wmax = L.obs_pad > rowsfwd::kWidth ? L.obs_pad : rowsfwd::kWidth;
  fa->Dp = L.obs_pad, fa->n_hidden = n_layers;
  fa->ld0 = wmax + 4, fa->ld1 = 0;            // (w + 4) / 4 odd: 16 rows of a b128 read hit 16 distinct 4-bank slots
  for (int l = 0; l < n_layers; ++l) fa->hidden[l] = sh->hidden[l];
  for (int net = 0; net < 2; ++net)
    for (int l = 0; l <= sh->n_hidden; ++l) fa->off_w[net][l] = L.off_w[net][l], fa->off_b[net][l] = L.off_b[net][l];
  // a 48-wide first layer reads its second 32-k slab 16 floats past every weight row: the last row's over-read must
  // stay inside the flat buffer (it lands in the bias that follows)
  // and the run-ahead requests of slabs past the last one read up to 160 floats past every weight matrix (its bias
  // and the next layer follow it in the flat buffer)
  for (int net = 0; net < 2; ++net)
    for (int l = 0; l < n_layers; ++l)
      if (L.off_w[net][l] + (int64_t)sh->hidden[l] * L.in_dim[l] + 160 > L.n_flat) return false;
  if (L.obs_pad > 256) return false;          // observation tile: eight float4 per thread
  *lds = R == 64 ? rowsfwd::lds_bytes<64>(fa->ld0) : rowsfwd::lds_bytes<32>(fa->ld0);
  return *lds <= 160 * 1024;
}

template <int R, bool TRAIN, int NETS, int NL>
void rows_fwd_launch_k(const FusedFwdArgs& a, size_t lds, int64_t tiles, int nets, hipStream_t s) {
  auto kern = rows_fwd_kernel<R, TRAIN, NETS, NL>;
  if (lds > 64 * 1024)
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  hipLaunchKernelGGL(kern, dim3((unsigned)tiles, NETS == 2 ? 1 : nets), dim3(rowsfwd::kThreads), lds, s, a);
}

// training (R = 64, activations stored): one to three layers, one or both networks per workgroup
bool rows_fwd_launch_train(const FusedFwdArgs& fa, size_t lds, int64_t rows, int n_cu, hipStream_t s) {
  FusedFwdArgs a = fa;
  const int64_t tiles = cdiv64(rows, 64);
  // enough row tiles to fill the chip: one workgroup walks both networks (one round of workgroups, the observation
  // tile of a row block fetched by one CU); fewer: one workgroup per (tile, network)
  a.nets_per_wg = tiles >= n_cu ? 2 : 1;
  a.store_policy = 0;
  const int nl = a.n_hidden;
  if (a.nets_per_wg == 2) {
    if (nl == 1) rows_fwd_launch_k<64, true, 2, 1>(a, lds, tiles, 2, s);
    else if (nl == 2) rows_fwd_launch_k<64, true, 2, 2>(a, lds, tiles, 2, s);
    else if (nl == 3) rows_fwd_launch_k<64, true, 2, 3>(a, lds, tiles, 2, s);
    else return false;
  } else {
    if (nl == 1) rows_fwd_launch_k<64, true, 1, 1>(a, lds, tiles, 2, s);
    else if (nl == 2) rows_fwd_launch_k<64, true, 1, 2>(a, lds, tiles, 2, s);
    else if (nl == 3) rows_fwd_launch_k<64, true, 1, 3>(a, lds, tiles, 2, s);
    else return false;
  }
  return true;
}

// rollout (R = 32, heads): one workgroup per (tile, network)
bool rows_fwd_launch_rollout(const FusedFwdArgs& fa, size_t lds, int64_t rows, int nets, hipStream_t s) {
  FusedFwdArgs a = fa;
  a.nets_per_wg = 1;
  const int64_t tiles = cdiv64(rows, 32);
  const int nl = a.n_hidden;
  if (nl == 1) rows_fwd_launch_k<32, false, 1, 1>(a, lds, tiles, nets, s);
  else if (nl == 2) rows_fwd_launch_k<32, false, 1, 2>(a, lds, tiles, nets, s);
  else if (nl == 3) rows_fwd_launch_k<32, false, 1, 3>(a, lds, tiles, nets, s);
  else return false;
  return true;
}

// rows_fwd_wide_kernel (fwd_rows_wide.h) applies when: fp32 MFMA, the first layer is 128 / 256 / 512 wide (512: consumed
// in two 256-column chunks by the layer above, which must be computed here too), every other computed layer 128 / 256
// wide, the padded observation width <= 64 (the observation tile persists next to the activation tile), and the run-ahead
// weight requests stay inside the flat parameter buffer.  Networks whose computed layers are ALL 256 wide keep
// rows_fwd_kernel (one in-place tile: also fits wide observations).
bool rows_wide_plan(const catppo_mlp_shape* sh, const catppo_mlp_layout& L, int n_layers, int R, FusedFwdArgs* fa,
                    size_t* lds, int* nch) {
  if (sh->mfma_bf16 != 0 || n_layers < 1 || n_layers > 3 || n_layers > sh->n_hidden) return false;
  if (L.obs_pad > 64) return false;
  const int w0 = sh->hidden[0];
  if (w0 != 128 && w0 != 256 && w0 != 512) return false;
  if (w0 == 512 && n_layers < 2) return false;
  for (int l = 1; l < n_layers; ++l)
    if (sh->hidden[l] != 128 && sh->hidden[l] != 256) return false;
  *nch = w0 == 512 ? 2 : 1;
  fa->Dp = L.obs_pad, fa->n_hidden = n_layers;
  fa->ld0 = rowsfwd::kTileLd, fa->ld1 = L.obs_pad + 4;      // (w + 4) / 4 odd for Dp = 16 / 32 / 48 / 64: conflict-free b128 rows
  for (int l = 0; l < n_layers; ++l) fa->hidden[l] = sh->hidden[l];
  for (int net = 0; net < 2; ++net)
    for (int l = 0; l <= sh->n_hidden; ++l) fa->off_w[net][l] = L.off_w[net][l], fa->off_b[net][l] = L.off_b[net][l];
  // over-reads: `pre` reads k 32..63 of every first-layer row (a 48-wide row: 16 floats into the next row / the bias),
  // the long contractions request up to three 32-k slabs past the end of a weight row range (<= 160 floats past a matrix)
  for (int net = 0; net < 2; ++net)
    for (int l = 0; l < n_layers; ++l)
      if (L.off_w[net][l] + (int64_t)sh->hidden[l] * L.in_dim[l] + 160 > L.n_flat) return false;
  *lds = R == 64 ? rowsfwd::wide_lds_bytes<64>(fa->ld1) : rowsfwd::wide_lds_bytes<32>(fa->ld1);
  return *lds <= 160 * 1024;
}

template <int R, bool TRAIN, int NETS, int NL, int NCH>
void rows_wide_launch_k(const FusedFwdArgs& a, size_t lds, int64_t tiles, int nets, hipStream_t s) {
  auto kern = rows_fwd_wide_kernel<R, TRAIN, NETS, NL, NCH>;
  if (lds > 64 * 1024)
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  hipLaunchKernelGGL(kern, dim3((unsigned)tiles, NETS == 2 ? 1 : nets), dim3(rowsfwd::kThreads), lds, s, a);
}

template <int R, bool TRAIN, int NETS>
bool rows_wide_dispatch(const FusedFwdArgs& a, size_t lds, int64_t tiles, int nets, int nch, hipStream_t s) {
  const int nl = a.n_hidden;
  if (nch == 1) {
    if (nl == 1) rows_wide_launch_k<R, TRAIN, NETS, 1, 1>(a, lds, tiles, nets, s);
    else if (nl == 2) rows_wide_launch_k<R, TRAIN, NETS, 2, 1>(a, lds, tiles, nets, s);
    else if (nl == 3) rows_wide_launch_k<R, TRAIN, NETS, 3, 1>(a, lds, tiles, nets, s);
    else return false;
  } else {
    if (nl == 2) rows_wide_launch_k<R, TRAIN, NETS, 2, 2>(a, lds, tiles, nets, s);
    else if (nl == 3) rows_wide_launch_k<R, TRAIN, NETS, 3, 2>(a, lds, tiles, nets, s);
    else return false;
  }
  return true;
}

// training (R = 64, activations stored)
bool rows_wide_launch_train(const FusedFwdArgs& fa, size_t lds, int nch, int64_t rows, int n_cu, hipStream_t s) {
  FusedFwdArgs a = fa;
  const int64_t tiles = cdiv64(rows, 64);
  a.nets_per_wg = tiles >= n_cu ? 2 : 1;
  if (a.nets_per_wg == 2) return rows_wide_dispatch<64, true, 2>(a, lds, tiles, 2, nch, s);
  return rows_wide_dispatch<64, true, 1>(a, lds, tiles, 2, nch, s);
}

// rollout (R = 32, heads): one workgroup per (tile, network)
bool rows_wide_launch_rollout(const FusedFwdArgs& fa, size_t lds, int nch, int64_t rows, int nets, hipStream_t s) {
  FusedFwdArgs a = fa;
  a.nets_per_wg = 1;
  return rows_wide_dispatch<32, false, 1>(a, lds, cdiv64(rows, 32), nets, nch, s);
}

// rollout policy step shared by catppo_policy_act / _ex / _rng and catppo_value / _ex
int policy_core(catppo_ctx* ctx, const catppo_mlp_shape* shape, const float* params, const float* x, int64_t N,
                const float* eps, const float* given_action, float* action, float* logprob, void* value,
                int value_dtype, const catppo_iter_state* rng_state, int rng_step, float* eps_out, bool critic_only,
                void* stream, const char* fn) {
  catppo_mlp_layout L;
  MlpWs w{};
  if (int rc = mlp_prologue(ctx, shape, N, false, &L, &w, fn)) return rc;
  CATPPO_CHECK_ARG(ctx, params && x && value && (critic_only || (action && logprob)));
  CATPPO_CHECK_ARG(ctx, value_dtype == CATPPO_F32 || value_dtype == CATPPO_F16);
  hipStream_t s = static_cast<hipStream_t>(stream);
  {
    // round 6: 16-row tiles (step16.h) for batches below the 32-row kernels' window - 2048 envs of an env-sharded rank, cfg1.
    // CATPPO_STEP16_FWD=0 keeps the layer-wise launches (A/B); CATPPO_STEP16_FWD_MAX_ROWS moves the bound.
    static const int s16f_on = env_int("CATPPO_STEP16_FWD", 1);
    static const int s16f_max = env_int("CATPPO_STEP16_FWD_MAX_ROWS", 2048);
    if (s16f_on && N <= s16f_max && shape->mfma_bf16 == 0 && shape->n_hidden == 3 && N * 512 * 4 < (int64_t(1) << 31)) {
      FusedFwdArgs fa{};
      fa.x = x, fa.params = params, fa.M = N, fa.Dp = L.obs_pad, fa.n_hidden = 3, fa.n_flat = L.n_flat;
      for (int net = 0; net < 2; ++net)
        for (int l = 0; l <= 3; ++l) fa.off_w[net][l] = L.off_w[net][l], fa.off_b[net][l] = L.off_b[net][l];
      fa.net0 = 0;
      fa.logstd = params + L.off_logstd, fa.eps = eps, fa.given = given_action, fa.A = shape->act_dim;
      fa.action = action, fa.logprob = logprob, fa.value_out = value, fa.value_f16 = (int)(value_dtype == CATPPO_F16);
      fa.rng_state = rng_state, fa.rng_step = rng_step, fa.eps_out = eps_out, fa.do_head = 1;
      bool done16 = false;
      const int tiles16 = (int)cdiv64(N, step16::kR);
      auto launch16 = [&](auto dp, auto n0, auto n1, auto n2) {
        constexpr int DP = decltype(dp)::value, N0 = decltype(n0)::value, N1 = decltype(n1)::value, N2 = decltype(n2)::value;
        constexpr size_t lds = sizeof(float) * step16::lds_floats<DP, N0, N1, N2>();
        auto kern = step16_fwd_kernel<DP, N0, N1, N2>;
        if (lds > 64 * 1024)
          (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        hipLaunchKernelGGL(kern, dim3((unsigned)tiles16, critic_only ? 1 : 2), dim3(step16::kThreads), lds, s, fa);
        done16 = true;
      };
      using std::integral_constant;
      const int h0 = shape->hidden[0], h1 = shape->hidden[1], h2 = shape->hidden[2];
#define CATPPO_S16(DP_, A_, B_, C_)                                                                          \
      if (!done16 && L.obs_pad == DP_ && h0 == A_ && h1 == B_ && h2 == C_)                                    \
        launch16(integral_constant<int, DP_>{}, integral_constant<int, A_>{}, integral_constant<int, B_>{},  \
                 integral_constant<int, C_>{});
      CATPPO_S16(48, 512, 256, 128)
      CATPPO_S16(48, 256, 256, 256)
#undef CATPPO_S16
      if (done16) {
        catppo_plan_note(ctx, "rollout forward, %lld rows: step16_fwd_kernel<%d, %d, %d, %d> + heads, %d tiles of 16 rows x %d networks, ONE "
                         "launch [<= %d rows, fp32, a compiled shape]", (long long)N, L.obs_pad, h0, h1, h2, tiles16, critic_only ? 1 : 2, s16f_max);
        CATPPO_CHECK_LAUNCH(ctx);
        return CATPPO_OK;
      }
    }
  }
  {
    // row-resident forward with full-line weight loads (round 4) for networks whose hidden layers are all 256 wide, same
    // window as fused_fwd_kernel (which keeps the other shapes): 32.1 -> 30 us per env step at cfg2, rollout 1.75 -> 1.70 ms
    // (interleaved A/B, profiles/r4_ab_rows_fwd.txt); CATPPO_ROWS_FWD_ROLLOUT=0 falls back to fused_fwd_kernel
    static const int rows_rollout = env_int("CATPPO_ROWS_FWD_ROLLOUT", 1);
    static const int rr_max = env_int("CATPPO_FUSED_FWD_MAX_ROWS", 4096), rr_min = env_int("CATPPO_FUSED_FWD_MIN_ROWS", 2049);
    FusedFwdArgs ra{};
    size_t rlds = 0;
    if (rows_rollout && N <= rr_max && N >= rr_min && shape->n_hidden <= 3 &&
        rows_fwd_plan(shape, L, shape->n_hidden, 32, &ra, &rlds)) {
      ra.x = x, ra.params = params, ra.M = N;
      ra.net0 = 0;
      ra.logstd = params + L.off_logstd, ra.eps = eps, ra.given = given_action, ra.A = shape->act_dim;
      ra.action = action, ra.logprob = logprob, ra.value_out = value, ra.value_f16 = (int)(value_dtype == CATPPO_F16);
      ra.rng_state = rng_state, ra.rng_step = rng_step, ra.eps_out = eps_out, ra.do_head = 1;
      rows_fwd_launch_rollout(ra, rlds, N, critic_only ? 1 : 2, s);
      catppo_plan_note(ctx, "rollout forward, %lld rows: rows_fwd_kernel<32> + heads, %lld tiles x %d networks, ONE launch "
                       "[every hidden layer 256 wide, %d..%d rows, fp32]", (long long)N, (long long)cdiv64(N, 32), critic_only ? 1 : 2, rr_min, rr_max);
      CATPPO_CHECK_LAUNCH(ctx);
      return CATPPO_OK;
    }
  }
  {
    // round 5: networks that are not 256 wide throughout (the reference's 512 / 256 / 128) - rows_fwd_wide_kernel<32>;
    // CATPPO_ROWS_WIDE=0 / CATPPO_ROWS_WIDE_ROLLOUT=0 fall back to fused_fwd_kernel (A/B)
    static const int wide_on = env_int("CATPPO_ROWS_WIDE", 1) && env_int("CATPPO_ROWS_WIDE_ROLLOUT", 1);
    static const int rr_max = env_int("CATPPO_FUSED_FWD_MAX_ROWS", 4096), rr_min = env_int("CATPPO_FUSED_FWD_MIN_ROWS", 2049);
    FusedFwdArgs wa{};
    size_t wlds = 0;
    int nch = 1;
    const int hl = shape->hidden[shape->n_hidden - 1];
    if (wide_on && N <= rr_max && N >= rr_min && (hl == 128 || hl == 256) &&
        rows_wide_plan(shape, L, shape->n_hidden, 32, &wa, &wlds, &nch)) {
      wa.x = x, wa.params = params, wa.M = N;
      wa.net0 = 0;
      wa.logstd = params + L.off_logstd, wa.eps = eps, wa.given = given_action, wa.A = shape->act_dim;
      wa.action = action, wa.logprob = logprob, wa.value_out = value, wa.value_f16 = (int)(value_dtype == CATPPO_F16);
      wa.rng_state = rng_state, wa.rng_step = rng_step, wa.eps_out = eps_out, wa.do_head = 1;
      if (rows_wide_launch_rollout(wa, wlds, nch, N, critic_only ? 1 : 2, s)) {
        catppo_plan_note(ctx, "rollout forward, %lld rows: rows_fwd_wide_kernel<32> + heads, %lld tiles x %d networks, ONE launch "
                         "[first layer %d wide in %d chunk(s), other layers 128 / 256, padded observations <= 64, fp32]",
                         (long long)N, (long long)cdiv64(N, 32), critic_only ? 1 : 2, shape->hidden[0], nch);
        CATPPO_CHECK_LAUNCH(ctx);
        return CATPPO_OK;
      }
    }
  }
  {
    FusedFwdArgs fa{};
    size_t lds = 0;
    if (fused_fwd_plan(shape, L, N, &fa, &lds)) {      // small batches: every layer + the head in one launch
      fa.x = x, fa.params = params, fa.M = N;
      fa.net0 = 0;
      fa.logstd = params + L.off_logstd, fa.eps = eps, fa.given = given_action, fa.A = shape->act_dim;
      fa.action = action, fa.logprob = logprob, fa.value_out = value, fa.value_f16 = (int)(value_dtype == CATPPO_F16);
      fa.rng_state = rng_state, fa.rng_step = rng_step, fa.eps_out = eps_out, fa.do_head = 1;
      if (lds > 64 * 1024)
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(fused_fwd_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
      hipLaunchKernelGGL(fused_fwd_kernel, dim3((unsigned)cdiv64(N, kFR), critic_only ? 1 : 2), dim3(kFT), lds, s, fa);
      catppo_plan_note(ctx, "rollout forward, %lld rows: fused_fwd_kernel (round 3: two ping-pong tiles, 16-k slabs), ONE launch "
                       "[shape outside the row-resident kernels: widths / observation width / 512-wide head]", (long long)N);
      CATPPO_CHECK_LAUNCH(ctx);
      return CATPPO_OK;
    }
  }
  // bf16 operands, large batches (BASELINE configs[4]: 32768 envs): bf16-stored activations between the layer-wise launches
  // (round 6, gemm_f32.h "act16"); the last hidden layer writes fp32 for head_act_kernel.  CATPPO_ACT16=0: fp32-stored.
  static const int act16_fwd = env_int("CATPPO_ACT16", 1);
  const bool fwd16 = act16_fwd && shape->mfma_bf16 == 1 && shape->n_hidden >= 2 && N >= 4096 && w.w16 != nullptr;
  if (fwd16) forward_hidden16(shape, L, params, x, N, w, critic_only ? 1 : 2, s, shape->n_hidden, true);
  else forward_hidden(shape, L, params, x, N, w, 0, critic_only ? 1 : 2, s);
  if (fwd16) catppo_plan_note(ctx, "rollout forward, %lld rows: bf16-stored activations between the layer-wise launches", (long long)N);
  catppo_plan_note(ctx, "rollout forward, %lld rows: %d layer-wise GEMM launches (gemm_f32_kernel) + head_act_kernel "
                   "[outside the one-launch window %s, or operand precision %d != fp32]", (long long)N, shape->n_hidden,
                   "CATPPO_FUSED_FWD_MIN_ROWS..MAX_ROWS (2049..4096)", shape->mfma_bf16);
  CATPPO_CHECK_LAUNCH(ctx);
  const int nl = shape->n_hidden, A = critic_only ? 0 : shape->act_dim;
  // one row per wave (4 per workgroup), up to 2048 workgroups: measured 9.5 us at 4096 rows against 13.7 us with 16 rows
  // per workgroup - the parallelism of many short workgroups beats amortising the 16 x HL head-weight staging
  int64_t nblk = cdiv64(N, 4);
  if (nblk > 2048) nblk = 2048;
  const float* nul = nullptr;
  const int rc = dispatch_cpl(shape->hidden[nl - 1], [&](auto cpl) {
    hipLaunchKernelGGL((head_act_kernel<decltype(cpl)::value>), dim3((unsigned)nblk), dim3(256),
                       sizeof(float) * 16 * shape->hidden[nl - 1], s, (const float*)w.H[0][nl - 1],
                       critic_only ? nul : (const float*)w.H[1][nl - 1], params + L.off_w[0][nl],
                       params + L.off_b[0][nl], critic_only ? nul : params + L.off_w[1][nl],
                       critic_only ? nul : params + L.off_b[1][nl], critic_only ? nul : params + L.off_logstd,
                       critic_only ? nul : eps, critic_only ? nul : given_action, N, A, action, logprob, value,
                       (int)(value_dtype == CATPPO_F16), critic_only ? nullptr : rng_state, rng_step, eps_out);
  });
  if (rc) return catppo_fail(ctx, CATPPO_E_ARG, "%s: last hidden width unsupported", fn);
  CATPPO_CHECK_LAUNCH(ctx);
  return CATPPO_OK;
}
}  // namespace

// ABI 0.6: the one rollout-forward entry (catppo_policy_act / _ex / _rng, catppo_value / _ex of ABI <= 0.5 are inline
// wrappers in include/catppo_compat.h).  action == NULL: critic only.  Action noise: `eps` (supplied N(0,1)), or `state`
// (Philox counter {env, dim / 4, step, iteration}, eps_out receives it), or `given_action` (evaluate these actions).
extern "C" int catppo_policy_step(catppo_ctx* ctx, const catppo_mlp_shape* shape, const float* params, const float* x,
                                  int64_t N, const float* eps, const float* given_action, const catppo_iter_state* state,
                                  int32_t step, float* eps_out, float* action, float* logprob, void* value, int value_dtype,
                                  void* stream) {
  CATPPO_CHECK_ARG(ctx, ctx != nullptr);
  CATPPO_CHECK_ARG(ctx, state == nullptr || (step >= 0 && eps == nullptr));
  const bool critic_only = action == nullptr;
  CATPPO_CHECK_ARG(ctx, !critic_only || (logprob == nullptr && eps == nullptr && given_action == nullptr && state == nullptr));
  return policy_core(ctx, shape, params, x, N, eps, given_action, action, logprob, value, value_dtype, state, step, eps_out,
                     critic_only, stream, __func__);
}

namespace {
// forward + losses + backward on an ALREADY GATHERED minibatch: xmb [M,Dp], act [M,A], scal [4][M]
// {old log-prob, advantage, normalised return, normalised value}, adv_part [nbg][2] fp64 advantage moments
int minibatch_grad_core(catppo_ctx* ctx, const catppo_mlp_shape* shape, const catppo_mlp_layout& L, MlpWs& w,
                        const catppo_ppo_hparams* hp, const float* params, int64_t M, const float* vrms_mean,
                        const float* vrms_var, const float* adv_stats, float* grad, float* diag, hipStream_t s,
                        NormEmit* ne = nullptr);
}  // namespace

extern "C" int catppo_ppo_minibatch_grad(catppo_ctx* ctx, const catppo_mlp_shape* shape,
                                         const catppo_ppo_hparams* hp, const float* params, const float* b_obs,
                                         const float* b_actions, const float* b_logprobs,
                                         const float* b_advantages, const float* b_returns_n,
                                         const float* b_values_n, const int64_t* mb_inds, int64_t M,
                                         const float* vrms_mean, const float* vrms_var, const float* adv_stats,
                                         float* grad, float* diag, void* stream) {
  catppo_mlp_layout L;
  MlpWs w{};
  if (int rc = mlp_prologue(ctx, shape, M, true, &L, &w, __func__)) return rc;
  CATPPO_CHECK_ARG(ctx, hp && params && b_obs && b_actions && b_logprobs && b_advantages && b_returns_n &&
                            b_values_n && mb_inds && vrms_mean && vrms_var && grad && diag);
  CATPPO_CHECK_ARG(ctx, !hp->adv_stats_external || adv_stats != nullptr);
  hipStream_t s = static_cast<hipStream_t>(stream);
  // 1. gather the minibatch (ppo.py:300-302,314,331-337 index with mb_inds)
  hipLaunchKernelGGL(ppo_gather_kernel, dim3((unsigned)cdiv64(M, kGatherRows), 1), dim3(256), 0, s, b_obs, b_actions,
                     b_logprobs, b_advantages, b_returns_n, b_values_n, mb_inds, M, M, L.obs_pad, shape->act_dim,
                     w.xmb, w.act, w.scal, w.adv_part, (const catppo_iter_state*)nullptr, 0, 0, (int64_t*)nullptr);
  CATPPO_CHECK_LAUNCH(ctx);
  return minibatch_grad_core(ctx, shape, L, w, hp, params, M, vrms_mean, vrms_var, adv_stats, grad, diag, s);
}

extern "C" int catppo_ppo_gather_ex(catppo_ctx* ctx, const catppo_mlp_shape* shape, const float* b_obs,
                                    const float* b_actions, const float* b_logprobs, const void* b_advantages,
                                    int adv_dtype, const float* b_returns_n, const float* b_values_n,
                                    const int64_t* inds, const catppo_iter_state* state, int32_t epoch, int64_t total,
                                    int64_t M, float* x_g, float* act_g, float* scal_g, double* adv_part_g,
                                    int64_t* inds_out, void* stream) {
  CATPPO_CHECK_ARG(ctx, ctx != nullptr);
  catppo_mlp_layout L;
  CATPPO_CHECK_ARG(ctx, shape && catppo_mlp_layout_of(shape, &L) == CATPPO_OK);
  CATPPO_CHECK_ARG(ctx, b_obs && b_actions && b_logprobs && b_advantages && b_returns_n && b_values_n);
  CATPPO_CHECK_ARG(ctx, (inds != nullptr) != (state != nullptr));    // exactly one source of the permutation
  if (inds != nullptr) state = nullptr;
  CATPPO_CHECK_ARG(ctx, adv_dtype == CATPPO_F32 || adv_dtype == CATPPO_F16);
  CATPPO_CHECK_ARG(ctx, x_g && act_g && scal_g && adv_part_g && total >= 1 && total < (int64_t(1) << 31) && M >= 1);
  const int64_t n_mb = cdiv64(total, M);
  CATPPO_CHECK_ARG(ctx, n_mb <= 65535);
  hipLaunchKernelGGL(ppo_gather_kernel, dim3((unsigned)cdiv64(M, kGatherRows), (unsigned)n_mb), dim3(256), 0,
                     static_cast<hipStream_t>(stream), b_obs, b_actions, b_logprobs,
                     static_cast<const float*>(b_advantages), b_returns_n, b_values_n, inds, total,
                     M, L.obs_pad, shape->act_dim, x_g, act_g, scal_g, adv_part_g, state, (int)epoch,
                     (int)(adv_dtype == CATPPO_F16), inds_out);
  CATPPO_CHECK_LAUNCH(ctx);
  return CATPPO_OK;
}

extern "C" int catppo_ppo_minibatch_grad_packed(catppo_ctx* ctx, const catppo_mlp_shape* shape,
                                                const catppo_ppo_hparams* hp, const float* params,
                                                const float* x_mb, const float* act_mb, const float* scal_mb,
                                                const double* adv_part_mb, int64_t M, const float* vrms_mean,
                                                const float* vrms_var, const float* adv_stats, float* grad,
                                                float* diag, void* stream) {
  catppo_mlp_layout L;
  MlpWs w{};
  if (int rc = mlp_prologue(ctx, shape, M, true, &L, &w, __func__)) return rc;
  CATPPO_CHECK_ARG(ctx, hp && params && x_mb && act_mb && scal_mb && adv_part_mb && vrms_mean && vrms_var && grad &&
                            diag);
  CATPPO_CHECK_ARG(ctx, !hp->adv_stats_external || adv_stats != nullptr);
  CATPPO_CHECK_ARG(ctx, (reinterpret_cast<uintptr_t>(x_mb) & 15) == 0);
  // the kernels only read these: point the workspace view at the caller's gathered slices
  w.xmb = const_cast<float*>(x_mb), w.act = const_cast<float*>(act_mb), w.scal = const_cast<float*>(scal_mb);
  w.adv_part = const_cast<double*>(adv_part_mb);
  return minibatch_grad_core(ctx, shape, L, w, hp, params, M, vrms_mean, vrms_var, adv_stats, grad, diag,
                             static_cast<hipStream_t>(stream));
}

namespace {
int minibatch_grad_core(catppo_ctx* ctx, const catppo_mlp_shape* shape, const catppo_mlp_layout& L, MlpWs& w,
                        const catppo_ppo_hparams* hp, const float* params, int64_t M, const float* vrms_mean,
                        const float* vrms_var, const float* adv_stats, float* grad, float* diag, hipStream_t s,
                        NormEmit* ne) {
  const int nl = shape->n_hidden, A = shape->act_dim, HL = shape->hidden[nl - 1];
  const int nbg = (int)cdiv64(M, kGatherRows);
  // small minibatches (env-sharded runs: 2048 samples per rank): 16-row tiles double the workgroup count of a launch
  // that would otherwise occupy a quarter of the CUs
  const bool small_tiles = HL <= 256 && cdiv64(M, head_rows(HL)) < 128;
  const int TRh = small_tiles ? 16 : head_rows(HL);
  int nbh = (int)cdiv64(M, TRh);
  // head_loss blocks = weight-gradient partials folded afterwards.  Its LDS tile decides residency: when only one
  // block fits a CU (HL >= 256) a second round of blocks cannot overlap the first, so one block per CU walks
  // several tiles and pays the set-up (head weights, advantage statistics, partial flush) once.
  const size_t head_lds =
      sizeof(float) * ((size_t)16 * HL + 2 * (size_t)TRh * HL + TRh * 16 + 48 + 4);
  const int head_cap = 2 * head_lds > 160 * 1024 ? kHeadMaxBlocks / 2 : kHeadMaxBlocks;
  if (nbh > head_cap) nbh = head_cap;

  // Large minibatches: the last hidden layer, the heads, the loss and the backward through the heads are ONE
  // launch (fwd_head_kernel).  Needs the full last-layer width in one tile (128 or 256 columns), a 16-aligned
  // contraction, and enough 64-row tiles to fill the chip (otherwise the 64x64-tile GEMM + head_loss pair has more
  // workgroups).  CATPPO_FUSED_HEAD=0 keeps the two launches.
  static const int fused_head_env = env_int("CATPPO_FUSED_HEAD", 1);
  static const int fused_head_min = env_int("CATPPO_FUSED_HEAD_MIN_WG", 128);    // workgroups of the fused launch (M >= 4096)
  const int RB = (int)cdiv64(M, 64);
  const bool fused_head = fused_head_env && (HL == 128 || HL == 256) && nl >= 2 &&
                          L.in_dim[nl - 1] % gemm::BK == 0 && 2 * RB >= fused_head_min && A <= 15;
  // Round 6, bf16 operands (BASELINE configs[4]): activations and dZ STORED as bf16, bf16 weight copies (gemm_f32.h "act16").
  // Needs the fused head launch (the 64-row head_loss path reads fp32 activations) and the plain single-stream backward.
  // CATPPO_ACT16=0: fp32-stored activations rounded at every use (rounds 2-5; A/B).
  static const int act16_env = env_int("CATPPO_ACT16", 1);
  const bool act16 = act16_env && shape->mfma_bf16 == 1 && fused_head && !ctx->use_side && ctx->grad_overlap != 1 && w.w16 != nullptr &&
                     getenv("CATPPO_NO_PAIR") == nullptr;
  // Round 6: small minibatches (an env-sharded rank's 2048 rows, cfg1) - forward, heads, loss, head backward and the data
  // gradients of the hidden layers in ONE launch of 16-row workgroups (step16.h), then every layer's weight gradient in
  // one grouped launch (dw_multi_kernel) and the fold: 3 launches instead of 10.  CATPPO_STEP16=0 keeps the layer-wise
  // launches (A/B), CATPPO_STEP16_MAX_ROWS moves the upper bound of the window.
  bool step16_done = false;
  {
    static const int s16_on = env_int("CATPPO_STEP16", 1);
    static const int s16_max = env_int("CATPPO_STEP16_MAX_ROWS", 4096);
    const bool plain = !ctx->use_side;      // (the side-stream experiment forks per layer: layer-wise launches only)
    if (s16_on && plain && M <= s16_max && shape->mfma_bf16 == 0 && nl == 3 && A <= 15 && M * 512 * 4 < (int64_t(1) << 31)) {
      step16::Args sa{};
      sa.x = w.xmb, sa.params = params, sa.M = M, sa.n_flat = L.n_flat;
      for (int net = 0; net < 2; ++net) {
        for (int l = 0; l <= nl; ++l) sa.off_w[net][l] = L.off_w[net][l], sa.off_b[net][l] = L.off_b[net][l];
        for (int l = 0; l < nl; ++l) sa.H[net][l] = l + 1 < nl ? w.H[net][l] : nullptr, sa.dZ[net][l] = w.dZ[net][l];
      }
      HeadArgs& g = sa.g;
      g.W4c = params + L.off_w[0][nl], g.b4c = params + L.off_b[0][nl];
      g.W4a = params + L.off_w[1][nl], g.b4a = params + L.off_b[1][nl];
      g.logstd = params + L.off_logstd;
      g.act = w.act, g.oldlogp = w.scal, g.adv = w.scal + M, g.ret_n = w.scal + 2 * M, g.val_n = w.scal + 3 * M;
      g.adv_part = w.adv_part, g.n_adv_part = nbg;
      g.adv_stats = hp->adv_stats_external ? adv_stats : nullptr;
      g.vrms_mean = vrms_mean, g.vrms_var = vrms_var;
      g.part_w = w.head_w, g.part_s = w.head_s, g.branch_out = ctx->branch_out;
      g.M = M, g.A = A, g.hp = *hp;
      const int tiles16 = (int)cdiv64(M, step16::kR);
      auto launch16 = [&](auto dp, auto n0, auto n1, auto n2) {
        constexpr int DP = decltype(dp)::value, N0 = decltype(n0)::value, N1 = decltype(n1)::value, N2 = decltype(n2)::value;
        constexpr size_t lds = sizeof(float) * step16::lds_floats<DP, N0, N1, N2>();
        auto kern = step16_kernel<DP, N0, N1, N2>;
        if (lds > 64 * 1024)
          (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        hipLaunchKernelGGL(kern, dim3((unsigned)tiles16, 2), dim3(step16::kThreads), lds, s, sa);
        step16_done = true;
      };
      using std::integral_constant;
      const int h0 = shape->hidden[0], h1 = shape->hidden[1], h2 = shape->hidden[2];
#define CATPPO_S16(DP_, A_, B_, C_)                                                                          \
      if (!step16_done && L.obs_pad == DP_ && h0 == A_ && h1 == B_ && h2 == C_)                               \
        launch16(integral_constant<int, DP_>{}, integral_constant<int, A_>{}, integral_constant<int, B_>{},  \
                 integral_constant<int, C_>{});
      CATPPO_S16(48, 512, 256, 128)      // the reference's Agent (45 / 48-d observations): cfg1, cfg3
      CATPPO_S16(48, 256, 256, 256)      // BASELINE configs[1]
#undef CATPPO_S16
      if (step16_done) {
        catppo_plan_note(ctx, "minibatch %lld rows: step16_kernel<%d, %d, %d, %d> - forward, heads, PPO loss, head backward and the "
                         "hidden layers' data gradients in ONE launch, %d tiles of 16 rows x 2 networks [<= %d rows, fp32, "
                         "a compiled shape]", (long long)M, L.obs_pad, h0, h1, h2, tiles16, s16_max);
        CATPPO_CHECK_LAUNCH(ctx);
        nbh = tiles16;
      }
    }
  }
  if (step16_done) {
  } else if (fused_head) {
    // hidden layers below the last: ONE row-resident launch (fwd_rows.h) when they are all 256 wide and the minibatch has
    // enough 64-row tiles, else the layer-wise GEMM launches.  CATPPO_ROWS_FWD=0 keeps the latter (A/B).
    static const int rows_fwd_env = env_int("CATPPO_ROWS_FWD", 1);
    static const int rows_fwd_min = env_int("CATPPO_ROWS_FWD_MIN_ROWS", 8192);
    FusedFwdArgs ra{};
    size_t rlds = 0;
    if (act16) {
      // bf16 copies of W_1 .. W_{nl-1} (as stored and transposed) + the layer-wise forward with bf16-stored activations
      const int64_t tot = forward_hidden16(shape, L, params, w.xmb, M, w, 2, s, nl - 1, false);
      catppo_plan_note(ctx, "minibatch %lld rows, bf16-stored activations: fwd0_w16_kernel (layer 0 + %lld weights as bf16, stored + "
                       "transposed, in one launch) + %d layer-wise forward GEMM launch(es) on bf16-stored operands", (long long)M,
                       (long long)tot, nl - 2);
    } else if (rows_fwd_env && M >= rows_fwd_min && M <= (1 << 20) && nl - 1 <= 3 &&
        rows_fwd_plan(shape, L, nl - 1, 64, &ra, &rlds)) {
      ra.x = w.xmb, ra.params = params, ra.M = M, ra.net0 = 0, ra.do_head = 0;
      for (int net = 0; net < 2; ++net)
        for (int l = 0; l < nl - 1; ++l) ra.Hout[net][l] = w.H[net][l];
      rows_fwd_launch_train(ra, rlds, M, ctx->n_cu, s);
      catppo_plan_note(ctx, "minibatch %lld rows, forward of hidden layers 0..%d: rows_fwd_kernel<64>, %lld row tiles, %s "
                       "[all 256 wide, >= %d rows, fp32]", (long long)M, nl - 2, (long long)cdiv64(M, 64),
                       cdiv64(M, 64) >= ctx->n_cu ? "one workgroup walks both networks" : "one workgroup per (tile, network)", rows_fwd_min);
    } else {
      // round 5: the same for networks that are not 256 wide throughout (reference: 512 / 256 below the 128-wide last
      // layer): rows_fwd_wide_kernel<64>; CATPPO_ROWS_WIDE=0 keeps the layer-wise launches (A/B)
      static const int wide_on = env_int("CATPPO_ROWS_WIDE", 1);
      FusedFwdArgs wa{};
      size_t wlds = 0;
      int nch = 1;
      bool done = false;
      if (wide_on && rows_fwd_env && M >= rows_fwd_min && M <= (1 << 20) &&
          rows_wide_plan(shape, L, nl - 1, 64, &wa, &wlds, &nch)) {
        wa.x = w.xmb, wa.params = params, wa.M = M, wa.net0 = 0, wa.do_head = 0;
        for (int net = 0; net < 2; ++net)
          for (int l = 0; l < nl - 1; ++l) wa.Hout[net][l] = w.H[net][l];
        done = rows_wide_launch_train(wa, wlds, nch, M, ctx->n_cu, s);
        if (done)
          catppo_plan_note(ctx, "minibatch %lld rows, forward of hidden layers 0..%d: rows_fwd_wide_kernel<64>, %lld row tiles "
                           "[first layer %d wide in %d chunk(s), other layers 128 / 256, padded observations <= 64, >= %d rows, fp32]",
                           (long long)M, nl - 2, (long long)cdiv64(M, 64), shape->hidden[0], nch, rows_fwd_min);
      }
      if (!done) {
        forward_hidden(shape, L, params, w.xmb, M, w, 0, 2, s, nl - 1);
        catppo_plan_note(ctx, "minibatch %lld rows, forward of hidden layers 0..%d: %d layer-wise GEMM launches "
                         "[not row-resident: < %d rows, operand precision %d, a width outside {128, 256, (512 first)}, or "
                         "padded observations > 64 with a non-256 layer]", (long long)M, nl - 2, nl - 1, rows_fwd_min, shape->mfma_bf16);
      }
    }
    CATPPO_CHECK_LAUNCH(ctx);
    Params p{};
    p.xcd_legacy = xcd_legacy();
    p.nets = 2, p.splits = 1;
    p.I = (int)M, p.J = HL, p.Kc = L.in_dim[nl - 1];
    p.lda = p.Kc, p.ldb = p.Kc, p.ldc = HL;
    for (int net = 0; net < 2; ++net) {
      p.op[net].A = w.H[net][nl - 2];
      p.op[net].B = params + L.off_w[net][nl - 1];
      p.op[net].bias = params + L.off_b[net][nl - 1];
      p.op[net].C = nullptr;              // the activations of the last layer never leave the CU
      if (act16) p.op[net].B = reinterpret_cast<const float*>(w.w16 + L.off_w[net][nl - 1]);
    }
    if (act16) p.Kc /= 2, p.lda /= 2, p.ldb /= 2;      // bf16-stored operands: FLOAT units (gemm_f32.h)
    HeadArgs g{};
    g.dZc = w.dZ[0][nl - 1], g.dZa = w.dZ[1][nl - 1];
    g.W4c = params + L.off_w[0][nl], g.b4c = params + L.off_b[0][nl];
    g.W4a = params + L.off_w[1][nl], g.b4a = params + L.off_b[1][nl];
    g.logstd = params + L.off_logstd;
    g.act = w.act, g.oldlogp = w.scal, g.adv = w.scal + M, g.ret_n = w.scal + 2 * M, g.val_n = w.scal + 3 * M;
    g.adv_part = w.adv_part, g.n_adv_part = nbg;
    g.adv_stats = hp->adv_stats_external ? adv_stats : nullptr;
    g.vrms_mean = vrms_mean, g.vrms_var = vrms_var;
    g.part_w = w.head_w, g.part_s = w.head_s, g.branch_out = ctx->branch_out;
    g.M = M, g.A = A, g.hp = *hp;
    auto launch_fh = [&](auto hl, auto prec) {
      constexpr int HLc = decltype(hl)::value, PR = decltype(prec)::value;
      constexpr size_t lds = sizeof(float) * fwd_head_lds_floats<HLc>();
      auto kern = fwd_head_kernel<HLc, PR>;
      if (lds > 64 * 1024)   // per call: the attribute belongs to the current device's copy of the kernel
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
      kern<<<dim3(RB, 1, 2), dim3(256), lds, s>>>(p, g);
    };
    using std::integral_constant;
    const int pr = act16 ? 3 : shape->mfma_bf16;
    if (HL == 256) {
      if (pr == 0) launch_fh(integral_constant<int, 256>{}, integral_constant<int, 0>{});
      else if (pr == 1) launch_fh(integral_constant<int, 256>{}, integral_constant<int, 1>{});
      else if (pr == 3) launch_fh(integral_constant<int, 256>{}, integral_constant<int, 3>{});
      else launch_fh(integral_constant<int, 256>{}, integral_constant<int, 2>{});
    } else {
      if (pr == 0) launch_fh(integral_constant<int, 128>{}, integral_constant<int, 0>{});
      else if (pr == 1) launch_fh(integral_constant<int, 128>{}, integral_constant<int, 1>{});
      else if (pr == 3) launch_fh(integral_constant<int, 128>{}, integral_constant<int, 3>{});
      else launch_fh(integral_constant<int, 128>{}, integral_constant<int, 2>{});
    }
    catppo_plan_note(ctx, "last hidden layer + heads + PPO loss + head backward: fwd_head_kernel<%d, prec %d>, %d tiles x 2 networks "
                     "[last layer 128 / 256 wide and >= %d workgroups]", HL, pr, RB, fused_head_min);
    CATPPO_CHECK_LAUNCH(ctx);
    nbh = RB;
  } else {
    // 2. hidden layers forward, both nets per launch
    forward_hidden(shape, L, params, w.xmb, M, w, 0, 2, s);
    CATPPO_CHECK_LAUNCH(ctx);

    // 3. heads + losses + gradient w.r.t. last hidden pre-activations
    HeadArgs g{};
    g.Hc = w.H[0][nl - 1], g.Ha = w.H[1][nl - 1];
    g.dZc = w.dZ[0][nl - 1], g.dZa = w.dZ[1][nl - 1];
    g.W4c = params + L.off_w[0][nl], g.b4c = params + L.off_b[0][nl];
    g.W4a = params + L.off_w[1][nl], g.b4a = params + L.off_b[1][nl];
    g.logstd = params + L.off_logstd;
    g.act = w.act, g.oldlogp = w.scal, g.adv = w.scal + M, g.ret_n = w.scal + 2 * M, g.val_n = w.scal + 3 * M;
    g.adv_part = w.adv_part, g.n_adv_part = nbg;
    g.adv_stats = hp->adv_stats_external ? adv_stats : nullptr;
    g.vrms_mean = vrms_mean, g.vrms_var = vrms_var;
    g.part_w = w.head_w, g.part_s = w.head_s, g.branch_out = ctx->branch_out;
    g.M = M, g.A = A, g.hp = *hp;
    const int rc = dispatch_cpl(HL, [&](auto cpl) {
      constexpr int CPL = decltype(cpl)::value;
      if constexpr (CPL <= 4) {
        if (small_tiles) {
          head_loss_kernel<CPL, 16><<<dim3(nbh), dim3(head_waves<CPL>() * 64), head_lds, s>>>(g);
          return;
        }
      }
      auto kern = head_loss_kernel<CPL>;
      if (head_lds > 64 * 1024)
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                                  (int)head_lds);
      kern<<<dim3(nbh), dim3(head_waves<CPL>() * 64), head_lds, s>>>(g);
    });
    if (rc) return catppo_fail(ctx, CATPPO_E_ARG, "%s: last hidden width unsupported", __func__);
    catppo_plan_note(ctx, "minibatch %lld rows: %d layer-wise forward GEMM launches + head_loss_kernel (%d-row tiles, %d blocks) "
                     "[fused last-layer launch needs a 128 / 256-wide last layer and >= %d workgroups = %d rows]",
                     (long long)M, nl, TRh, nbh, fused_head_min, fused_head_min * 32);
    CATPPO_CHECK_LAUNCH(ctx);
  }

  // 4. backward through the hidden layers; split-K partials for every weight gradient
  SegTable segs{};
  auto add_seg = [&](const float* src, float* dst, int64_t count, int64_t stride, int n_parts, int mode,
                     float scale) {
    Seg& sg = segs.s[segs.n++];
    sg.src = src, sg.dst = dst, sg.count = count, sg.stride = stride, sg.n_parts = n_parts, sg.mode = mode,
    sg.scale = scale;
  };
  // Backward, default: per hidden layer ONE launch holding the split-K weight-gradient GEMM and the data-gradient
  // GEMM (launch_dw_dx_pair), every layer with its own partial buffers, ONE fold launch at the end.
  // CATPPO_SIDE_STREAM=1 (measured slower, kept for A/B): the weight gradients are forked to the context's side
  // stream as soon as a layer's dZ exists and joined before returning to the caller's stream order.
  const bool fork = ctx->use_side;
  // catppo_set_grad_overlap + a communicator: fold and all-reduce the gradient in per-layer buckets on the side stream
  // while the backward launches of the layers below run on `s` (see the end of the layer loop)
  // (the 16-row step has no per-layer launches to hide buckets behind: with either overlap mode it folds once and reduces
  // the whole gradient in one grouped operation, the `tail` form's fallback below)
  const bool overlap = !fork && ctx->grad_overlap == 1 && ctx->comm != nullptr && !step16_done;
  // round 5, "tail" form: no extra launch; the ranges that are final after dw_fold_kernel travel on the side stream under
  // the final fold launch, the first layer's own ranges behind it on `s`
  const bool tail = !fork && ctx->comm != nullptr && (ctx->grad_overlap == 2 || (ctx->grad_overlap == 1 && step16_done));
  bool tail_forked = false;
  if (ne != nullptr && (fork || overlap || tail))
    return catppo_fail(ctx, CATPPO_E_ARG, "%s: the one-call optimiser step cannot run with the side-stream weight "
                       "gradients or the gradient buckets (something reduces the gradient between fold and clip)", __func__);
  if (ne != nullptr) ne->n_slots = 0;
  const int bf16 = shape->mfma_bf16;   // 0 fp32 MFMA, 1 bf16 operands, 2 split-bf16 (bf16x3)
  hipStream_t side = fork ? ctx->side : s;
#define CATPPO_HIP_OK(call)                                                                          \
  do {                                                                                               \
    hipError_t e__ = (call);                                                                         \
    if (e__ != hipSuccess)                                                                           \
      return catppo_fail(ctx, CATPPO_E_HIP, "%s: %s failed: %s", __func__, #call, hipGetErrorString(e__)); \
  } while (0)
  // split-K weight-gradient problem of hidden layer l (tiling rule shared by every backward path)
  auto dw_params = [&](int l) {
    const int out = shape->hidden[l], in = L.in_dim[l];
    Params pw{};
    pw.xcd_legacy = xcd_legacy();
    pw.nets = 2;
    pw.I = out, pw.J = in, pw.Kc = (int)M;
    pw.lda = out, pw.ldb = in, pw.ldc = in;
    const int tiles = ((out + 127) / 128) * ((in + 127) / 128) * 2;
    int splits = 512 / (tiles > 0 ? tiles : 1);
    const int max_by_rows = (int)cdiv64(M, 4 * gemm::BK);
    if (splits > max_by_rows) splits = max_by_rows;
    if (splits > split_cap(out, in)) splits = split_cap(out, in);
    if (splits < 1) splits = 1;
    int per = (int)cdiv64(M, splits);
    per = (per + gemm::BK - 1) / gemm::BK * gemm::BK;
    if (!(out >= 128 && in >= 128 && per >= 256)) {
      // 64x64 tiles will be used (narrow layer, or a minibatch too small for 256-row contraction chunks): size the
      // split for ~512 workgroups with at least 128 contraction rows each.  At 2048 samples the old rule cut a
      // 256x512 layer into 2048 workgroups of 64 rows - four slabs of work between a prologue and a 33 MB partial store.
      const int t64 = ((out + 63) / 64) * ((in + 63) / 64) * 2;
      constexpr int target_wg = 512;      // workgroups a 64x64-tile weight gradient aims for (256 / 384 / 1024 measured: profiles/r6_ab_dw_target.txt)
      splits = target_wg / (t64 > 0 ? t64 : 1);
      const int max128 = (int)(M / 128);
      if (splits > max128) splits = max128;
      if (splits > split_cap(out, in)) splits = split_cap(out, in);
      if (splits < 1) splits = 1;
      per = (int)cdiv64(M, splits);
      per = (per + gemm::BK - 1) / gemm::BK * gemm::BK;
    }
    if (act16) per = (per + 31) / 32 * 32;            // 32-k slabs of the bf16-stored weight-gradient loop
    splits = (int)cdiv64(M, per);
    pw.splits = splits;
    pw.kc_per_split = per;
    pw.c_split_stride = 2 * (int64_t)out * in;    // [split][net][out*in]
    for (int net = 0; net < 2; ++net) {
      pw.op[net].A = w.dZ[net][l];
      pw.op[net].B = l == 0 ? w.xmb : w.H[net][l - 1];
      pw.op[net].C = w.wpart[l] + (int64_t)net * out * in;
      pw.op[net].dbias = w.bpart[l] + (int64_t)net * splits * out;   // [net][split][out]
    }
    return pw;
  };
  const bool head_by_net = fused_head || step16_done;     // head partial rows [0, nbh) actor, [nbh, 2 nbh) critic
  auto add_head_segs = [&]() {
    // head partials + diagnostics ride along with the reduction launch
    const int NS = head_scalars(A);
    const int64_t wrow = (int64_t)(A + 1) * HL;
    const float* cw = w.head_w + (head_by_net ? (int64_t)nbh * wrow : 0);
    const float* cs = w.head_s + (head_by_net ? (int64_t)nbh * NS : 0);
    add_seg(w.head_w, grad + L.off_w[1][nl], (int64_t)A * HL, wrow, nbh, 0, 1.0f);
    add_seg(cw + (int64_t)A * HL, grad + L.off_w[0][nl], HL, wrow, nbh, 0, 1.0f);
    add_seg(w.head_s, grad + L.off_b[1][nl], A, NS, nbh, 0, 1.0f);
    add_seg(cs + A, grad + L.off_b[0][nl], 1, NS, nbh, 0, 1.0f);
    add_seg(w.head_s + A + 1, grad + L.off_logstd, A, NS, nbh, 0, 1.0f);
    add_seg(w.head_s + 2 * A + 1, diag, kHeadDiag, NS, head_by_net ? 2 * nbh : nbh, 1, hp->inv_global_batch);
  };
  if (step16_done) {
    // every dZ is in memory: all weight gradients in ONE grouped launch of 64x64-tile split-K workgroups, the layer with
    // the longest contraction chunks first
    DwMulti dm{};
    int order[CATPPO_MAX_HIDDEN];
    for (int l = 0; l < nl; ++l) order[l] = l;
    Params pws[CATPPO_MAX_HIDDEN];
    for (int l = 0; l < nl; ++l) pws[l] = dw_params(l);
    for (int i = 0; i < nl; ++i)
      for (int j = i + 1; j < nl; ++j)
        if (pws[order[j]].kc_per_split > pws[order[i]].kc_per_split) { const int t = order[i]; order[i] = order[j]; order[j] = t; }
    int total = 0;
    for (int i = 0; i < nl; ++i) {
      const Params& pw = pws[order[i]];
      dm.p[i] = pw;
      dm.tiles[i] = tiles_of<64, 64>(pw);
      dm.first[i] = total;
      total += dm.tiles[i] * pw.nets * pw.splits;
    }
    dm.first[nl] = total, dm.n = nl;
    constexpr size_t dw_lds = gemm::smem_bytes<64, 64, false, false>();
    hipLaunchKernelGGL(dw_multi_kernel, dim3((unsigned)total), dim3(256), dw_lds, s, dm);
    catppo_plan_note(ctx, "weight gradients of all %d hidden layers: dw_multi_kernel, %d workgroups of 64x64 split-K tiles, ONE launch", nl, total);
    CATPPO_CHECK_LAUNCH(ctx);
    for (int l = nl - 1; l >= 0; --l) {
      const int out = shape->hidden[l], in = L.in_dim[l], splits = pws[l].splits;
      for (int net = 0; net < 2; ++net) {
        add_seg(w.wpart[l] + (int64_t)net * out * in, grad + L.off_w[net][l], (int64_t)out * in, 2 * (int64_t)out * in, splits, 0, 1.0f);
        add_seg(w.bpart[l] + (int64_t)net * splits * out, grad + L.off_b[net][l], out, out, splits, 0, 1.0f);
      }
      if (l == nl - 1) add_head_segs();
    }
  }
  for (int l = step16_done ? -1 : nl - 1; l >= 0; --l) {
    const int out = shape->hidden[l], in = L.in_dim[l];
    // dZ_l is complete on the main stream here: fork
    if (fork) {
      CATPPO_HIP_OK(hipEventRecord(ctx->ev_fork[l], s));
      CATPPO_HIP_OK(hipStreamWaitEvent(side, ctx->ev_fork[l], 0));
    }
    // weight gradient: dW[out,in] = dZ^T . Xin      (contraction over the M rows)
    const Params pw = dw_params(l);
    const int splits = pw.splits, per = pw.kc_per_split;
    static const bool no_pair = getenv("CATPPO_NO_PAIR") != nullptr;
    const bool pair = l > 0 && !fork && !no_pair;
    // (round 4, measured and removed: a 256 x 64 tile for the narrow first layer - one workgroup per CU owning all 256
    // output rows of a network for its slice of the batch, dZ_0 and the observations read once - was 3.5 us per step
    // SLOWER than the 64x64 tiling, 9.48 vs 9.39 ms of update phase, profiles/r4_ab_dw0_tile.txt: twice the partial
    // bytes for the fold and 64 single-dword write-through stores per lane in the epilogue of a workgroup that only
    // multiplies 8 slabs)
    // the first layer's weight gradient shares its launch with the fold of the layers above it (dw_fold_kernel) when it
    // is the plain 64x64-tile fp32 launch on the caller's stream; CATPPO_DW0_FOLD=0 keeps GEMM and fold apart (A/B)
    static const int dw0_fold = env_int("CATPPO_DW0_FOLD", 1);
    const bool dw_with_fold = !pair && l == 0 && dw0_fold && !fork && !overlap && segs.n > 0 &&
                              (act16 || !(pw.I >= 128 && pw.J >= 128 && pw.kc_per_split >= 256));      // launch_gemm_auto's 128x128 rule
    if (dw_with_fold) {
      const int t64 = tiles_of<64, 64>(pw), n_gemm = t64 * pw.nets * pw.splits;
      constexpr size_t lds = gemm::smem_bytes<64, 64, false, false>();
      static_assert(lds >= 4096, "the fold workgroups use 1024 floats of the same allocation");
      const dim3 grid((unsigned)(n_gemm + kFoldX * segs.n));
      double* nslots = ne ? ne->part + ne->n_slots : (double*)nullptr;
      // (round 6, again: 128x64 tiles for this GEMM - two accumulators per wave, the observations read once per 128 rows, partials
      //  through the staged 16-byte stores round 4 did not have - measured 3 us per step SLOWER at cfg2 and at the reference shapes)
      if (act16)
        hipLaunchKernelGGL(dw_fold_kernel<5>, grid, dim3(256), lds, s, pw, segs, t64, n_gemm, hp->ent_coef, hp->vf_coef, nslots);
      else if (bf16 == 2)
        hipLaunchKernelGGL(dw_fold_kernel<2>, grid, dim3(256), lds, s, pw, segs, t64, n_gemm, hp->ent_coef, hp->vf_coef, nslots);
      else if (bf16 == 1)
        hipLaunchKernelGGL(dw_fold_kernel<1>, grid, dim3(256), lds, s, pw, segs, t64, n_gemm, hp->ent_coef, hp->vf_coef, nslots);
      else
        hipLaunchKernelGGL(dw_fold_kernel<0>, grid, dim3(256), lds, s, pw, segs, t64, n_gemm, hp->ent_coef, hp->vf_coef, nslots);
      CATPPO_CHECK_LAUNCH(ctx);
      catppo_plan_note(ctx, "layer 0 weight gradient (%d x %d, %d splits of %d rows) + fold of the %d partial segments of the other "
                       "layers / heads: dw_fold_kernel, %d + %d workgroups", out, in, splits, per, segs.n, n_gemm, kFoldX * segs.n);
      if (ne) ne->n_slots += kFoldX * segs.n;
      segs.n = 0;        // folded; what is added below (this layer's own partials) goes to the final fold launch
      if (tail) {
        // every range of the flat gradient except the first layer's (W0 | b0 of both networks) is final now
        CATPPO_HIP_OK(hipEventRecord(ctx->ev_fork[0], s));
        CATPPO_HIP_OK(hipStreamWaitEvent(ctx->side, ctx->ev_fork[0], 0));
        int64_t off[3], cnt[3];
        off[0] = L.off_logstd, cnt[0] = L.off_w[0][0] - L.off_logstd;
        off[1] = L.off_w[0][1], cnt[1] = L.off_w[1][0] - L.off_w[0][1];
        off[2] = L.off_w[1][1], cnt[2] = L.n_flat - L.off_w[1][1];
        if (int rc = catppo_internal_allreduce_ranges(ctx, grad, off, cnt, 3, ctx->side)) return rc;
        CATPPO_HIP_OK(hipEventRecord(ctx->ev_join, ctx->side));
        tail_forked = true;
      }
    } else if (!pair) {
      if (act16) launch_gemm_prec<64, 64, false, false, gemm::EPI_PARTIAL, 5>(pw, side);     // bf16-stored dZ_0, fp32 observations
      else launch_gemm_auto<false, false, gemm::EPI_PARTIAL>(pw, side, bf16);
      catppo_plan_note(ctx, "layer %d weight gradient (%d x %d, %d splits of %d rows): gemm_f32_kernel, split-K partials "
                       "[own launch: first layer without the fold (precision %d / switches), or the side-stream experiment]",
                       l, out, in, splits, per, bf16);
      CATPPO_CHECK_LAUNCH(ctx);
    }
    for (int net = 0; net < 2; ++net) {
      add_seg(w.wpart[l] + (int64_t)net * out * in, grad + L.off_w[net][l], (int64_t)out * in,
              2 * (int64_t)out * in, splits, 0, 1.0f);
      add_seg(w.bpart[l] + (int64_t)net * splits * out, grad + L.off_b[net][l], out, out, splits, 0, 1.0f);
    }
    if (l == nl - 1) add_head_segs();
    if (l > 0) {
      // data gradient: dZ_{l-1} = (dZ_l . W_l) * elu'(H_{l-1})
      Params px{};
      px.xcd_legacy = xcd_legacy();
      px.nets = 2;
      px.splits = 1;
      px.I = (int)M, px.J = in, px.Kc = out;
      px.lda = out, px.ldb = in, px.ldc = in, px.ldaux = in;
      for (int net = 0; net < 2; ++net) {
        px.op[net].A = w.dZ[net][l];
        px.op[net].B = params + L.off_w[net][l];
        px.op[net].C = w.dZ[net][l - 1];
        px.op[net].aux = w.H[net][l - 1];
      }
      if (act16) {
        // dZ_l (A) and the transposed bf16 weight copy (B, [in][out]) are K-contiguous: contraction sizes in FLOAT units;
        // aux (H_{l-1}) and the output dZ_{l-1} are bf16-stored: ldaux / ldc in bf16 elements
        px.Kc = out / 2, px.lda = out / 2, px.ldb = out / 2;
        for (int net = 0; net < 2; ++net) px.op[net].B = reinterpret_cast<const float*>(w.w16t + L.off_w[net][l]);
      }
      if (pair && act16) {
        launch_dw_dx_pair16(pw, px, s, ctx->n_cu);
        catppo_plan_note(ctx, "layer %d weight gradient (%d x %d, %d splits of %d rows) + data gradient (%lld x %d, k = %d): "
                         "gemm_pair_kernel on bf16-stored operands, ONE launch", l, out, in, splits, per, (long long)M, in, out);
      } else if (pair) {
        launch_dw_dx_pair(pw, px, s, bf16, ctx->n_cu);
        catppo_plan_note(ctx, "layer %d weight gradient (%d x %d, %d splits of %d rows) + data gradient (%lld x %d, k = %d): "
                         "gemm_pair_kernel, ONE launch%s", l, out, in, splits, per, (long long)M, in, out,
                         M <= kSmallRows ? " [<= 4096 rows: 64x64 weight-gradient tiles]" : "");
      } else {
        launch_gemm_auto<true, false, gemm::EPI_MUL_DELU>(px, s, bf16);
        catppo_plan_note(ctx, "layer %d data gradient: gemm_f32_kernel (own launch)", l);
      }
      CATPPO_CHECK_LAUNCH(ctx);
    }
    if (overlap) {
      // Bucket l = {W_l, b_l of both networks} (+ heads and log-std with the last hidden layer): its partials are
      // complete once the launch above is done, so its fold and its all-reduce go to the side stream NOW and run under
      // the launches of layers l-1 .. 0.  Per element the sums are those of the single fold launch (seg_reduce treats
      // every segment independently), the ranges of a bucket are contiguous per network in the flat layout
      // (W_l | b_l | W_l+1 ...) and travel as one grouped RCCL operation.
      // (the first layer's bucket has nothing left to hide behind - its weight gradient is the last GEMM of the step -
      // so it stays on `s`: one fork / join pair less, measured 24 us per step for three forks on a world of one)
      hipStream_t bs = l > 0 ? ctx->side : s;
      if (l > 0) {
        CATPPO_HIP_OK(hipEventRecord(ctx->ev_fork[l], s));
        CATPPO_HIP_OK(hipStreamWaitEvent(ctx->side, ctx->ev_fork[l], 0));
      }
      hipLaunchKernelGGL(seg_reduce_kernel, dim3(256, segs.n), dim3(256), 0, bs, segs, hp->ent_coef, hp->vf_coef,
                         (double*)nullptr, (catppo_iter_state*)nullptr, 0.0, 0.0);
      CATPPO_CHECK_LAUNCH(ctx);
      segs.n = 0;
      int64_t off[3], cnt[3];
      int nr = 0;
      const bool last = l == nl - 1;
      for (int net = 0; net < 2; ++net) {
        // end of this network's (W_l, b_l) = start of its next layer; the bucket of the last hidden layer runs on
        // through the head layer to the end of the network's block
        const int64_t end = last ? (net == 0 ? L.off_w[1][0] : L.n_flat) : L.off_w[net][l + 1];
        off[nr] = L.off_w[net][l], cnt[nr] = end - L.off_w[net][l], ++nr;
      }
      if (last) off[nr] = L.off_logstd, cnt[nr] = L.off_w[0][0] - L.off_logstd, ++nr;
      // join BEFORE the first layer's own all-reduce: every operation on the communicator is then ordered by stream
      // dependencies (no two of them concurrently in flight on different streams), inside a captured graph too
      if (l == 0 && nl > 1) CATPPO_HIP_OK(hipStreamWaitEvent(s, ctx->ev_join, 0));
      if (int rc = catppo_internal_allreduce_ranges(ctx, grad, off, cnt, nr, bs)) return rc;
      if (l == 1) CATPPO_HIP_OK(hipEventRecord(ctx->ev_join, ctx->side));   // the last forked bucket
    }
  }
  if (overlap) return CATPPO_OK;      // `s` has joined the side stream in front of the last bucket
  // every split-K / head partial of the minibatch is folded into the flat gradient by one launch
  hipLaunchKernelGGL(seg_reduce_kernel, dim3(256, segs.n), dim3(256), 0, side, segs, hp->ent_coef, hp->vf_coef,
                     ne ? ne->part + ne->n_slots : (double*)nullptr, ne ? ne->st : (catppo_iter_state*)nullptr,
                     ne ? ne->beta1 : 0.0, ne ? ne->beta2 : 0.0);
  catppo_plan_note(ctx, "final fold: seg_reduce_kernel, %d segments x 256 workgroups%s", segs.n,
                   ne ? " + squared-norm slots and Adam step advance (one-call optimiser step)" : "");
  CATPPO_CHECK_LAUNCH(ctx);
  if (ne) ne->n_slots += 256 * segs.n;
  if (tail) {
    if (tail_forked) {      // join first: two operations on one communicator are never in flight on two streams at once
      CATPPO_HIP_OK(hipStreamWaitEvent(s, ctx->ev_join, 0));
      int64_t off[2] = {L.off_w[0][0], L.off_w[1][0]}, cnt[2] = {L.off_w[0][1] - L.off_w[0][0], L.off_w[1][1] - L.off_w[1][0]};
      if (int rc = catppo_internal_allreduce_ranges(ctx, grad, off, cnt, 2, s)) return rc;
    } else {                // shapes whose first-layer weight gradient does not share its launch with the fold: one all-reduce
      int64_t off[1] = {0}, cnt[1] = {L.n_flat};
      if (int rc = catppo_internal_allreduce_ranges(ctx, grad, off, cnt, 1, s)) return rc;
    }
  }
  if (fork) {
    CATPPO_HIP_OK(hipEventRecord(ctx->ev_join, side));
    CATPPO_HIP_OK(hipStreamWaitEvent(s, ctx->ev_join, 0));
  }
#undef CATPPO_HIP_OK
  return CATPPO_OK;
}
}  // namespace

extern "C" int catppo_clip_adam(catppo_ctx* ctx, float* params, float* grad, float* exp_avg, float* exp_avg_sq,
                                int64_t n_flat, float max_grad_norm, double lr, double beta1, double beta2,
                                double eps, int64_t step, void* stream) {
  CATPPO_CHECK_ARG(ctx, ctx != nullptr);
  CATPPO_CHECK_ARG(ctx, params && grad && exp_avg && exp_avg_sq && n_flat >= 1 && step >= 1);
  hipStream_t s = static_cast<hipStream_t>(stream);
  WsCarver ws(ctx);
  double* part = ws.take<double>(kNormBlocks);
  CATPPO_NEED_WS(ctx, part);
  int nb = (int)cdiv64(n_flat, 256 * 4);
  if (nb > kNormBlocks) nb = kNormBlocks;
  hipLaunchKernelGGL(sqnorm_partial_kernel, dim3(nb), dim3(256), 0, s, (const float*)grad, n_flat, part);
  CATPPO_CHECK_LAUNCH(ctx);
  // bias corrections in double on the host, like torch.optim.Adam's Python scalars
  const double bc1 = 1.0 - pow(beta1, (double)step);
  const double bc2 = 1.0 - pow(beta2, (double)step);
  const float step_size = (float)(lr / bc1);
  const float bc2_sqrt = (float)sqrt(bc2);
  int nblk = (int)cdiv64(n_flat, 256 * 4);
  if (nblk > 1024) nblk = 1024;
  hipLaunchKernelGGL(clip_adam_kernel, dim3(nblk), dim3(256), 0, s, params, grad, exp_avg, exp_avg_sq, n_flat,
                     (const double*)part, nb, max_grad_norm, (float)beta1, (float)beta2, (float)(1.0 - beta1),
                     (float)(1.0 - beta2), (float)eps, step_size, bc2_sqrt);
  CATPPO_CHECK_LAUNCH(ctx);
  return CATPPO_OK;
}


extern "C" int catppo_clip_adam_dev(catppo_ctx* ctx, float* params, float* grad, float* exp_avg, float* exp_avg_sq,
                                    int64_t n_flat, float max_grad_norm, double beta1, double beta2, double eps,
                                    catppo_iter_state* state, void* stream) {
  CATPPO_CHECK_ARG(ctx, ctx != nullptr);
  CATPPO_CHECK_ARG(ctx, params && grad && exp_avg && exp_avg_sq && n_flat >= 1 && state);
  hipStream_t s = static_cast<hipStream_t>(stream);
  WsCarver ws(ctx);
  double* part = ws.take<double>(kNormBlocks);
  CATPPO_NEED_WS(ctx, part);
  int nb = (int)cdiv64(n_flat, 256 * 4);
  if (nb > kNormBlocks) nb = kNormBlocks;
  hipLaunchKernelGGL(sqnorm_partial_step_kernel, dim3(nb), dim3(256), 0, s, (const float*)grad, n_flat, part, state, beta1,
                     beta2);
  CATPPO_CHECK_LAUNCH(ctx);
  int nblk = (int)cdiv64(n_flat, 256 * 4);
  if (nblk > 1024) nblk = 1024;
  hipLaunchKernelGGL(clip_adam_dev_kernel, dim3(nblk), dim3(256), 0, s, params, grad, exp_avg, exp_avg_sq, n_flat,
                     (const double*)part, nb, max_grad_norm, beta1, beta2, (float)eps,
                     (const catppo_iter_state*)state);
  CATPPO_CHECK_LAUNCH(ctx);
  return CATPPO_OK;
}

// One optimiser step of a single process in one call (ABI 0.4): catppo_ppo_minibatch_grad_packed + catppo_clip_adam_dev
// with the squared gradient norm emitted by the fold launches (NormEmit) instead of a launch that re-reads the gradient.
extern "C" int catppo_ppo_minibatch_step_packed(catppo_ctx* ctx, const catppo_mlp_shape* shape,
                                                const catppo_ppo_hparams* hp, float* params, const float* x_mb,
                                                const float* act_mb, const float* scal_mb, const double* adv_part_mb,
                                                int64_t M, const float* vrms_mean, const float* vrms_var,
                                                const float* adv_stats, float* grad, float* diag, float* exp_avg,
                                                float* exp_avg_sq, float max_grad_norm, double beta1, double beta2,
                                                double eps, catppo_iter_state* state, void* stream) {
  catppo_mlp_layout L;
  MlpWs w{};
  if (int rc = mlp_prologue(ctx, shape, M, true, &L, &w, __func__)) return rc;
  CATPPO_CHECK_ARG(ctx, hp && params && x_mb && act_mb && scal_mb && adv_part_mb && vrms_mean && vrms_var && grad &&
                            diag && exp_avg && exp_avg_sq && state);
  CATPPO_CHECK_ARG(ctx, !hp->adv_stats_external || adv_stats != nullptr);
  CATPPO_CHECK_ARG(ctx, (reinterpret_cast<uintptr_t>(x_mb) & 15) == 0);
  hipStream_t s = static_cast<hipStream_t>(stream);
  w.xmb = const_cast<float*>(x_mb), w.act = const_cast<float*>(act_mb), w.scal = const_cast<float*>(scal_mb);
  w.adv_part = const_cast<double*>(adv_part_mb);
  NormEmit ne;
  ne.part = w.norm_part, ne.st = state, ne.beta1 = beta1, ne.beta2 = beta2;
  if (int rc = minibatch_grad_core(ctx, shape, L, w, hp, params, M, vrms_mean, vrms_var, adv_stats, grad, diag, s, &ne))
    return rc;
  if (ne.n_slots < 1 || ne.n_slots > kNormSlots)
    return catppo_fail(ctx, CATPPO_E_ARG, "%s: %d squared-norm slots", __func__, ne.n_slots);
  int nblk = (int)cdiv64(L.n_flat, 256 * 4);
  if (nblk > 1024) nblk = 1024;
  hipLaunchKernelGGL(clip_adam_dev_kernel, dim3(nblk), dim3(256), 0, s, params, grad, exp_avg, exp_avg_sq,
                     (int64_t)L.n_flat, (const double*)ne.part, ne.n_slots, max_grad_norm, beta1, beta2, (float)eps,
                     (const catppo_iter_state*)state);
  catppo_plan_note(ctx, "clip + Adam: clip_adam_dev_kernel, %d workgroups, norm from %d slots of the fold launches", nblk, ne.n_slots);
  CATPPO_CHECK_LAUNCH(ctx);
  return CATPPO_OK;
}

// Test hook (tests/test_gpu_parity_sizes.py): buf != NULL - [2][M] int32 - makes the head / loss kernel of
// every following catppo_ppo_minibatch_* call write the clip branch each sample took (surrogate codes [0, M), value-loss codes
// [M, 2 M): 0 inside, 1 below, 2 above the clip range; cleanrl/ppo.py:320-341); NULL switches it off.  Costs nothing when off.
extern "C" int catppo_debug_clip_branches(catppo_ctx* ctx, int32_t* buf) {
  if (!ctx) return CATPPO_E_ARG;
  ctx->branch_out = buf;
  return CATPPO_OK;
}

#ifdef STEP16_TL
extern "C" int catppo_debug_step16_tl(void* buf) {     // timeline builds only: not part of include/catppo.h
  unsigned long long* pbuf = static_cast<unsigned long long*>(buf);
  return hipMemcpyToSymbol(HIP_SYMBOL(step16::g_s16tl), &pbuf, sizeof(pbuf)) == hipSuccess ? 0 : -1;
}
#endif

#ifdef FUSED_TL
// timeline builds only (tools/rows_fwd_timeline.py): the training launch of rows_fwd_kernel on its own
extern "C" int catppo_debug_rows_fwd(catppo_ctx* ctx, const catppo_mlp_shape* shape, const float* params, const float* x,
                                     int64_t M, int n_layers, void* stream) {
  catppo_mlp_layout L;
  MlpWs w{};
  if (int rc = mlp_prologue(ctx, shape, M, true, &L, &w, __func__)) return rc;
  FusedFwdArgs ra{};
  size_t rlds = 0;
  if (!rows_fwd_plan(shape, L, n_layers, 64, &ra, &rlds)) return CATPPO_E_ARG;
  ra.x = x, ra.params = params, ra.M = M, ra.net0 = 0, ra.do_head = 0;
  for (int net = 0; net < 2; ++net)
    for (int l = 0; l < n_layers; ++l) ra.Hout[net][l] = w.H[net][l];
  if (!rows_fwd_launch_train(ra, rlds, M, ctx->n_cu, static_cast<hipStream_t>(stream))) return CATPPO_E_ARG;
  CATPPO_CHECK_LAUNCH(ctx);
  return CATPPO_OK;
}
#endif
