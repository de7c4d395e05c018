// Fused rollout step: everything between the simulator's state update and the next policy forward in THREE launches
// (two in rounds 2-3: the fold behind rollout_pre was a "last workgroup to arrive" tree inside that launch).
//
// The reference spends, per env step, ~15 eager launches + one host sync PER constraint term
// (cat/constraint_manager.py:39-82,213-229), six launches + a nonzero() sync in CaTEnv.step (cat/cat_env.py:92-121),
// ~12 launches in RunningMeanStd (cleanrl/ppo.py:12-62) and seven buffer copies (cleanrl/ppo.py:203-226).  Round 1 of
// this build did the same work in ten launches (env_pre_step, cat_terms, cat_reduce_ema, cat_finish, cat_reset, a
// masked fill, rollout_store, rms_moments_partial, rms_final_merge, rms_normalize); at 4096 envs every one of them is
// latency bound (72-92 % of the wave cycles parked), so the step costs ten launch gaps.  Here:
//
//   rollout_pre   16-env tiles.  process_action, counters, terminations, raw reward; all constraint terms -> cstr;
//                 per-workgroup column maxima and fp64 observation moments (one partial row per workgroup).
//   rollout_fold  one workgroup per 16 columns folds the partial rows in fixed order into the exchange buffer
//                 {colmax[K] | sum x, sum x^2 [2D]}.  Round 4: a launch boundary (~1.5 us) in place of two in-kernel
//                 hand-shakes (arrive on a device-scope ticket, ~3.5 us coherent re-read of the partial rows, each):
//                 1.58 -> 1.43 ms per 24-step rollout at cfg2 (profiles/r4_ab_rollout_fold_launch.txt).  The old tree
//                 stays behind CATPPO_ROLLOUT_TREE=1; both orders are fixed, maxima are order independent, the fp64
//                 moment sums differ in the last bit between the two (tests hold either against the oracle's bar).
//   [env-sharded runs all-reduce the exchange buffer here: MAX for the maxima, SUM for the moments]
//   rollout_post  32-env tiles.  Every workgroup derives the new running maxima (EMA) and the merged normaliser
//                 statistics from the exchange buffer on its own (K + D values: cheaper than another launch), then
//                 does the CaT probabilities / statistics / reward / dones of its envs, the manager-reset statistics
//                 of the envs that reset (+ zeroing of their accumulators, episode length, action history), the
//                 rollout-buffer rows and the normalised next-observation rows.  The last workgroup to finish writes
//                 the new state back (nobody reads it any more in this launch) and folds the reset statistics.
//
// Arithmetic: identical statements in identical order to cat_step.hip / rms.hip / env_step.hip (this file is
// compiled with -ffp-contract=off as well), so the termination masks stay bit-exact.
#include "terms_eval.h"
#include <cstdlib>
#include "xwg.h"

namespace {

using namespace terms;
// -DROLLOUT_TL (tools/rollout_timeline.py): thread 0 of every workgroup stamps the wall clock (100 MHz) at the phase
// boundaries of the two kernels into a buffer set through catppo_debug_rollout_tl().  Compiles away otherwise.
#ifdef ROLLOUT_TL
__device__ unsigned long long* g_rtl;     // [2 kernels][1024 workgroups][16 stamps]
#define RL_TL(k, i) do { if (threadIdx.x == 0 && g_rtl) g_rtl[((k) * 1024 + blockIdx.x) * 16 + (i)] = wall_clock64(); } while (0)
#else
#define RL_TL(k, i) do { } while (0)
#endif

constexpr int kThreads = 256;
constexpr int kPostRows = 32;
constexpr int kMaxObsPerThread = 2;     // D <= 512

struct PreArgs {
  int64_t N;
  int A, D, K;
  const float* action_in;
  float* action;
  float* prev_action;
  int64_t* ep_len;
  int64_t max_len;
  const float* hard_reset;
  int64_t hr_stride;
  const float* reward_src;
  int64_t rw_stride;
  uint8_t* time_outs;
  uint8_t* terminated;
  uint8_t* reset;
  float* reward;
  const float* forces;
  int64_t fstride;
  int H, B;
  const float* command;
  int cld;
  float* cstr;
  const float* obs_raw;
  int64_t obs_ld;
  float* colmax_partial;    // [grid][K]
  double* osum_partial;     // [grid][2D]
  float* colmax_group;      // [n_groups][K]    second level of the fold tree
  double* osum_group;       // [n_groups][2D]
  float* x_colmax;          // exchange buffer
  double* x_sums;
  unsigned int* ticket;     // [0] = launch-wide ticket, [1 + g] = ticket of workgroup group g
  // simulator state advance (catppo_rollout_step::sim_src): rows of row_q4 float4 copied src -> dst by the tile's workgroup
  const float4* sim_src;
  float4* sim_dst;
  int sim_row_q4;
  int tree;                 // 1: fold the partial rows inside this launch (rounds 2-3); 0: rollout_fold_kernel does it
  int nblk;                 // workgroups that walk tiles (the grid holds one more when a deferred post tail rides along)
  int tail_nblk;            // workgroups of the post launch whose tail rides along (its partial rows)
};
constexpr int kFoldGroup = 32;    // workgroups per first-level fold
constexpr int kMaxPreBlocks = 1024;

// Block-wide fold of partial[nrows][ncols] over the rows.  Thread (c, g): column c, row group g of G <= 256 / ncols
// groups; a thread walks rows g, g+G, ... with four accumulators (row j of the thread goes to accumulator j % 4, a tail of
// fewer than four rows to accumulator 0), the G group results are combined through LDS in ascending g.  The order of the
// combination depends only on (nrows, ncols) => deterministic.  Up to 16 rows per thread (every fold of the rollout
// step) are ALL requested before the first is used: the rows come from the coherence point (xwg_load), a round trip of
// ~2 us each, and a fold used to pay three or four of them in sequence.  finish(c, value): one thread per column.
template <typename T, typename Op, typename Fin>
__device__ __forceinline__ void block_fold(const T* __restrict__ partial, int nrows, int ncols, T init, Op op, T* lds,
                                           Fin finish) {
  constexpr int kInFlight = 16;
  const int Wc = ncols < kThreads ? ncols : kThreads;
  // narrow folds (one column: 256 candidate groups) keep 16 groups unless that leaves a thread more than 16 rows:
  // the LDS combination below is serial in G
  const int Gmax = kThreads / Wc;
  const int Gneed = (nrows + kInFlight - 1) / kInFlight;
  const int G = Gmax <= 16 ? Gmax : (Gneed > 16 ? (Gneed < Gmax ? Gneed : Gmax) : 16);
  const int c0 = threadIdx.x % Wc, g = threadIdx.x / Wc;
  for (int cb = 0; cb < ncols; cb += Wc) {
    const int c = cb + c0;
    T acc = init;
    if (g < G && c < ncols) {
      T a0 = init, a1 = init, a2 = init, a3 = init;
      const int n_t = g < nrows ? (nrows - g + G - 1) / G : 0;       // rows of this thread
      if (n_t <= kInFlight) {
        T x[kInFlight];
#pragma unroll
        for (int j = 0; j < kInFlight; ++j) {
          const int bb = g + j * G;
          x[j] = xwg_load(partial + (int64_t)(bb < nrows ? bb : (nrows - 1)) * ncols + c);   // past the end: re-read, unused
        }
        const int n4 = n_t / 4 * 4;
#pragma unroll
        for (int j = 0; j < kInFlight; ++j) {
          if (j < n4) {
            if (j % 4 == 0) a0 = op(a0, x[j]);
            if (j % 4 == 1) a1 = op(a1, x[j]);
            if (j % 4 == 2) a2 = op(a2, x[j]);
            if (j % 4 == 3) a3 = op(a3, x[j]);
          } else if (j < n_t) {
            a0 = op(a0, x[j]);
          }
        }
      } else {
        int b = g;
        for (; b + 3 * G < nrows; b += 4 * G) {
          const T x0 = xwg_load(partial + (int64_t)b * ncols + c);
          const T x1 = xwg_load(partial + (int64_t)(b + G) * ncols + c);
          const T x2 = xwg_load(partial + (int64_t)(b + 2 * G) * ncols + c);
          const T x3 = xwg_load(partial + (int64_t)(b + 3 * G) * ncols + c);
          a0 = op(a0, x0), a1 = op(a1, x1), a2 = op(a2, x2), a3 = op(a3, x3);
        }
        for (; b < nrows; b += G) a0 = op(a0, xwg_load(partial + (int64_t)b * ncols + c));
      }
      acc = op(op(a0, a1), op(a2, a3));
    }
    lds[threadIdx.x] = acc;
    __syncthreads();
    if (g == 0 && c < ncols) {
      for (int gg = 1; gg < G; ++gg) acc = op(acc, lds[gg * Wc + c0]);
      finish(c, acc);
    }
    __syncthreads();
  }
}

struct TermMetaS {
  int32_t off[kMaxTerms + 1];
  float dp[kMaxTerms];
};

struct PostArgs {
  int64_t N;
  int A, D, K, n_terms;
  const float* cstr;
  float min_p, tau, one_minus_tau;
  int first_call;
  float* rm;
  float* reward;
  const uint8_t* reset;
  const uint8_t* time_outs;
  float* cstr_prob;
  float* dones;
  float* ep_viol;
  float* ep_prob;
  float* probs;
  int64_t* ep_len;
  float* action;
  float* prev_action;
  int zero_action;
  const float* log_prev;
  float* log_out;
  void* rewards_t;
  void* dones_t1;
  void* true_dones_t1;
  int planes_f16;
  const float* obs_raw;
  int64_t obs_ld;
  float* obs_mean;
  float* obs_var;
  float* obs_count;
  float obs_eps;
  double obs_n;
  float* obs_out;
  int64_t obs_out_ld;
  const float* x_colmax;
  const double* x_sums;
  int x_records;          // > 1: x_colmax / x_sums point at record 0 of `x_records` gathered records, x_stride bytes apart
  int64_t x_stride;
  double* reset_part;     // [grid][2 n_terms + 1]: per term {sum violation, sum probability}, then the reset count
  unsigned int* ticket;
  int defer;              // 1: no tail in this launch (post_tail_deferred runs it from a later launch)
  uint32_t d_magic, k_magic;   // catppo_div_magic(D), (K)
};
static_assert(sizeof(PostArgs) <= sizeof(catppo_ctx::post_tail_args), "catppo_ctx::post_tail_args too small");

__device__ __forceinline__ void store_plane(void* p, int64_t i, float v, int f16) {
  if (f16) reinterpret_cast<_Float16*>(p)[i] = (_Float16)v;
  else reinterpret_cast<float*>(p)[i] = v;
}

// new running maximum of column c (constraint_manager.py:58-61) from the exchange record(s) and the state of the previous
// step (every workgroup of rollout_post_kernel, into LDS; the last one to arrive - or the deferred tail - writes it back).
// Split into the loads and the arithmetic so that rollout_post_kernel can request the operands up front.
__device__ __forceinline__ float load_colmax(const PostArgs& a, int c) {
  float m = a.x_colmax[c];
  for (int w = 1; w < a.x_records; ++w)     // gathered records of the other ranks: MAX is exact and order independent
    m = nanmax(m, reinterpret_cast<const float*>(reinterpret_cast<const char*>(a.x_colmax) + w * a.x_stride)[c]);
  return m;
}
__device__ __forceinline__ float running_max_from(const PostArgs& a, float m, float rm_c) {
  if (a.first_call) return m;
  const float x = rm_c * a.tau;              // rm.mul_(tau)
  const float y = a.one_minus_tau * m;       // (1-tau) * cmax
  return x + y;                              // .add_()
}
__device__ __forceinline__ float derive_running_max(const PostArgs& a, int c) {
  return running_max_from(a, load_colmax(a, c), a.first_call ? 0.0f : a.rm[c]);
}
// merged observation normaliser of column c (cleanrl/ppo.py:48-62, the op order of rms.hip): new mean / variance
__device__ __forceinline__ void load_sums(const PostArgs& a, int c, double* sx_out, double* sxx_out) {
  const int D = a.D;
  double sx = a.x_sums[c], sxx = a.x_sums[D + c];
  for (int w = 1; w < a.x_records; ++w) {   // rank order: the same sums on every rank
    const double* xs = reinterpret_cast<const double*>(reinterpret_cast<const char*>(a.x_sums) + w * a.x_stride);
    sx += xs[c], sxx += xs[D + c];
  }
  *sx_out = sx, *sxx_out = sxx;
}
__device__ __forceinline__ void normaliser_from(const PostArgs& a, double sx, double sxx, float mean, float var, float cnt,
                                                float nf, float tot, float* new_mean, float* new_var) {
  const double m = sx / a.obs_n;
  double v = sxx / a.obs_n - m * m;
  if (v < 0.0) v = 0.0;
  const float bm = (float)m, bv = (float)v;
  const float delta = bm - mean;
  float t = delta * nf;
  t = t / tot;
  *new_mean = mean + t;
  const float m_a = var * cnt;
  const float m_b = bv * nf;
  float d2 = delta * delta;
  d2 = d2 * cnt;
  d2 = d2 * nf;
  d2 = d2 / tot;
  float M2 = m_a + m_b;
  M2 = M2 + d2;
  *new_var = M2 / tot;
}
__device__ __forceinline__ void derive_normaliser(const PostArgs& a, int c, float cnt, float nf, float tot, float* new_mean,
                                                  float* new_var) {
  double sx, sxx;
  load_sums(a, c, &sx, &sxx);
  normaliser_from(a, sx, sxx, a.obs_mean[c], a.obs_var[c], cnt, nf, tot, new_mean, new_var);
}

// Reset statistics of a post launch: its `nblk` partial rows {sum violation, sum probability per term | number of envs
// that reset} folded in ONE pass (the rows come from the coherence point, ~3.5 us per round trip - the count used to be
// its own fold in front of this one) into the log slot.  One workgroup.
__device__ __forceinline__ void fold_reset_log(const PostArgs& a, const int nblk, double* red) {
  const int nt = a.n_terms;
  __shared__ double s_sum[2 * kMaxTerms + 1];
  block_fold<double>(a.reset_part, nblk, 2 * nt + 1, 0.0, [](double x, double y) { return x + y; }, red,
                     [&](int c, double v) { s_sum[c] = v; });
  const double n = s_sum[2 * nt];
  if (threadIdx.x < 2 * nt) {
    const int c = threadIdx.x;                  // column c = 2*t (violation) | 2*t+1 (probability)
    const double v = s_sum[c];
    if (n > 0.0) a.log_out[c] = (c & 1) ? (float)(v / n) : (float)(v / n) * 100.0f;
    else if (a.log_prev != nullptr) a.log_out[c] = a.log_prev[c];
  }
}

// The tail of a rollout_post launch when it is DEFERRED (catppo_rollout_defer_tail): run by one workgroup of a LATER
// launch - the extra workgroup of the next rollout_pre, or rollout_post_tail_kernel - i.e. after every workgroup of the
// post launch has finished, with nothing in between having touched the state or the exchange record(s).  The new running
// maxima and the merged normaliser are derived once more - the same functions on the same inputs the post launch's
// workgroups used - and published; then the reset statistics.  `red`: LDS, kThreads doubles.
__device__ __forceinline__ void post_tail_deferred(const PostArgs& a, const int nblk, double* red) {
  const int K = a.K, D = a.D;
  for (int c = threadIdx.x; c < K; c += kThreads) a.rm[c] = derive_running_max(a, c);      // column c reads rm[c] only
  if (a.obs_raw != nullptr) {
    const float cnt = a.obs_count[0];
    const float nf = (float)a.obs_n;
    const float tot = cnt + nf;
    for (int c = threadIdx.x; c < D; c += kThreads) {
      float new_mean, new_var;
      derive_normaliser(a, c, cnt, nf, tot, &new_mean, &new_var);
      a.obs_mean[c] = new_mean, a.obs_var[c] = new_var;
    }
    __syncthreads();                            // every thread holds the old count
    if (threadIdx.x == 0) a.obs_count[0] = tot;
  }
  if (a.log_out != nullptr) fold_reset_log(a, nblk, red);
}

__global__ __launch_bounds__(kThreads) void rollout_pre_kernel(const TermTable tab, const PreArgs a, const PostArgs tl) {
  __shared__ double fold_lds[kThreads];
  // one workgroup more than a.nblk: the deferred tail of the previous rollout_post launch (state it publishes is read by
  // the NEXT post launch only; nothing the other workgroups of this launch touch)
  if ((int)blockIdx.x == a.nblk) {
    post_tail_deferred(tl, a.tail_nblk, fold_lds);
    return;
  }
  extern __shared__ float tile[];          // [kRows*K] constraint tile + [K] running column maxima
  float* cmax = tile + kRows * a.K;
  // the id lists of the terms, staged in LDS: the lane-varying lookup ids[j] is then an LDS read in front of the
  // state load instead of a second global-memory round trip through the kernel-argument segment
  __shared__ int s_ids[kMaxTerms][CATPPO_TERM_MAX_IDS];
  const int K = a.K, A = a.A, D = a.D;
  // (the wave index as a scalar: tab.d[t] below is then read with scalar loads - their own counter - instead of vector
  //  loads from the kernel-argument segment that would queue up behind the state copy's requests)
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  RL_TL(0, 0);
  for (int c = threadIdx.x; c < K; c += kThreads) cmax[c] = -__builtin_inff();
  for (int o = threadIdx.x; o < tab.n * CATPPO_TERM_MAX_IDS; o += kThreads)
    s_ids[o / CATPPO_TERM_MAX_IDS][o % CATPPO_TERM_MAX_IDS] = tab.d[o / CATPPO_TERM_MAX_IDS].ids[o % CATPPO_TERM_MAX_IDS];
  __syncthreads();
  // observation moments: thread (c, g) = column c, row group g of the 16-row tile (G = 256 / D groups, one when D > 128)
  const int Dc = D < kThreads ? (D > 0 ? D : 1) : kThreads;
  const int OG = kThreads / Dc;
  const int oc = threadIdx.x % Dc, og = threadIdx.x / Dc;
  double os1[kMaxObsPerThread], os2[kMaxObsPerThread];
#pragma unroll
  for (int q = 0; q < kMaxObsPerThread; ++q) os1[q] = 0.0, os2[q] = 0.0;

  const int64_t n_tiles = (a.N + kRows - 1) / kRows;
  RL_TL(0, 1);
  for (int64_t tile_i = blockIdx.x; tile_i < n_tiles; tile_i += a.nblk) {
    const int64_t r0 = tile_i * kRows;
    const int rows = (int)((a.N - r0) < kRows ? (a.N - r0) : kRows);

    // ---- simulator state advance: this tile's rows of the new state block go to the persistent state buffer.  Nothing
    //      in this launch reads the destination (every state input has been re-based onto the source by the host side),
    //      so the copy is independent traffic beside the dependent load chains of the terms below.
    //      Round 5: the rows are REQUESTED here (which also brings the lines the terms read into the CU's cache) and
    //      stored behind the terms: gfx9 counts loads and stores in one in-order counter, so with the stores in front
    //      every wait of the term loads below also waited for the copy's stores to reach memory.
    constexpr int kCopyQ = 8;                       // float4 per thread held in registers: 16 rows x 2 KB
    const int copy_n = a.sim_src != nullptr ? rows * a.sim_row_q4 : 0;
    const bool copy_regs = copy_n <= kCopyQ * kThreads;
    const int64_t copy_base = r0 * a.sim_row_q4;
    // (a native vector type: arrays of HIP's float4 struct are copied through memcpy and end up in scratch)
    typedef float f4v __attribute__((ext_vector_type(4)));
    const f4v* copy_src = reinterpret_cast<const f4v*>(a.sim_src) + copy_base;
    f4v* copy_dst = reinterpret_cast<f4v*>(a.sim_dst) + copy_base;
    f4v cq[kCopyQ];
    if (copy_regs) {
#pragma unroll
      for (int j = 0; j < kCopyQ; ++j) {
        const int e = threadIdx.x + j * kThreads;
        cq[j] = e < copy_n ? copy_src[e] : f4v{0.0f, 0.0f, 0.0f, 0.0f};
      }
    }

    // ---- constraint terms (the action-rate term reads action_in / the not yet shifted action buffer)
    for (int t = wave; t < tab.n; t += kThreads / 64) {
      const catppo_term_desc& d = tab.d[t];
      const int W = d.width;
      const int col0 = tab.off[t];
      const uint32_t wm = tab.wmagic[t];
      for (int w = lane; w < rows * W; w += 64) {
        const int e = fast_div(w, wm), j = w - e * W;
        tile[e * K + col0 + j] = eval_term(d, s_ids[t], r0 + e, j, a.forces, a.fstride, a.H, a.B, a.command, a.cld);
      }
    }
    RL_TL(0, 9);
    // ---- counters, terminations, raw reward (cat_env.py:92-97)
    if (threadIdx.x < rows) {
      const int64_t i = r0 + threadIdx.x;
      const int64_t len = a.ep_len[i] + 1;
      a.ep_len[i] = len;
      const bool to = len >= a.max_len;
      const bool term = a.hard_reset[i * a.hr_stride] > 0.5f;
      a.time_outs[i] = to;
      a.terminated[i] = term;
      a.reset[i] = to || term;
      a.reward[i] = a.reward_src[i * a.rw_stride];
    }
    RL_TL(0, 10);
    // ---- observation moments of the tile (fp64, fixed order)
    if (a.obs_raw != nullptr && og < OG) {
#pragma unroll
      for (int q = 0; q < kMaxObsPerThread; ++q) {
        const int c = oc + q * kThreads;
        if (c < D) {
          float v[kRows];                         // all loads of the tile first (independent), then the fp64 adds
#pragma unroll
          for (int k = 0; k < kRows; ++k) {
            const int r = og + k * OG;
            v[k] = r < rows ? a.obs_raw[(r0 + r) * a.obs_ld + c] : 0.0f;
          }
          double s1 = os1[q], s2 = os2[q];
#pragma unroll
          for (int k = 0; k < kRows; ++k) {
            if (og + k * OG < kRows) {
              const double x = (double)v[k];
              s1 += x;
              s2 += x * x;
            }
          }
          os1[q] = s1, os2[q] = s2;
        }
      }
    }
    RL_TL(0, 11);
    if (copy_regs) {
#pragma unroll
      for (int j = 0; j < kCopyQ; ++j) {
        const int e = threadIdx.x + j * kThreads;
        if (e < copy_n) copy_dst[e] = cq[j];
      }
    } else {
      for (int e = threadIdx.x; e < copy_n; e += kThreads) copy_dst[e] = copy_src[e];
    }
    __syncthreads();
    RL_TL(0, 12);
    // ---- flush the tile, running column maxima, and only now shift the action history (process_action)
    float* dst = a.cstr + r0 * K;
    for (int e = threadIdx.x; e < rows * K; e += kThreads) dst[e] = tile[e];
    for (int c = threadIdx.x; c < K; c += kThreads) {
      float m = cmax[c];
      for (int r = 0; r < rows; ++r) m = nanmax(m, tile[r * K + c]);
      cmax[c] = m;
    }
    for (int e = threadIdx.x; e < rows * A; e += kThreads) {
      const int64_t o = r0 * A + e;
      a.prev_action[o] = a.action[o];
      a.action[o] = a.action_in[o];
    }
    __syncthreads();
  }
  RL_TL(0, 2);
  for (int c = threadIdx.x; c < K; c += kThreads) xwg_store(a.colmax_partial + (int64_t)blockIdx.x * K + c, cmax[c]);
  if (a.obs_raw != nullptr) {
    // combine the row groups of the block in ascending g (fixed order), one partial row per block
#pragma unroll
    for (int q = 0; q < kMaxObsPerThread; ++q) {
      const int c = oc + q * kThreads;
      for (int pass = 0; pass < 2; ++pass) {
        __syncthreads();
        fold_lds[threadIdx.x] = pass == 0 ? os1[q] : os2[q];
        __syncthreads();
        if (og == 0 && c < D && (q == 0 || D > kThreads)) {
          double t = fold_lds[oc];
          for (int gg = 1; gg < OG; ++gg) t += fold_lds[gg * Dc + oc];
          xwg_store(a.osum_partial + (int64_t)blockIdx.x * 2 * D + pass * D + c, t);
        }
      }
    }
  }
  // round 4: the partial rows are folded by a small launch of their own (rollout_fold_kernel, below) - the two
  // hand-shakes of the in-launch tree cost more than a kernel boundary; a.tree keeps the tree for A/B
  if (!a.tree) return;
  // ---- fold tree, two levels, no extra launch: the last workgroup of every group of 32 folds the group's partial rows,
  //      the last of those folds the group rows into the exchange buffer.  Every fold is at most 32 rows deep, and the
  //      order of every sum depends only on the grid size - not on which workgroup happens to arrive last.
  const int nblk = a.nblk;
  const int grp = blockIdx.x / kFoldGroup, n_grp = (nblk + kFoldGroup - 1) / kFoldGroup;
  const int g0 = grp * kFoldGroup, g_rows = (nblk - g0) < kFoldGroup ? (nblk - g0) : kFoldGroup;
  RL_TL(0, 3);
  const bool last1 = last_block_arrives(a.ticket + 1 + grp, (unsigned)g_rows);
  RL_TL(0, 4);
  if (!last1) return;
  block_fold<float>(a.colmax_partial + (int64_t)g0 * K, g_rows, K, -__builtin_inff(),
                    [](float x, float y) { return nanmax(x, y); }, reinterpret_cast<float*>(fold_lds),
                    [&](int c, float m) { xwg_store(a.colmax_group + (int64_t)grp * K + c, m); });
  if (a.obs_raw != nullptr)
    block_fold<double>(a.osum_partial + (int64_t)g0 * 2 * D, g_rows, 2 * D, 0.0, [](double x, double y) { return x + y; },
                       fold_lds, [&](int c, double v) { xwg_store(a.osum_group + (int64_t)grp * 2 * D + c, v); });
  RL_TL(0, 5);
  const bool last2 = last_block_arrives(a.ticket, (unsigned)n_grp);
  RL_TL(0, 6);
  if (!last2) return;
  block_fold<float>(a.colmax_group, n_grp, K, -__builtin_inff(), [](float x, float y) { return nanmax(x, y); },
                    reinterpret_cast<float*>(fold_lds), [&](int c, float m) {
                      a.x_colmax[c] = (m < 1e-6f) ? 1e-6f : m;      // clamp(min=1e-6); NaN stays NaN like torch
                    });
  if (a.obs_raw != nullptr)
    block_fold<double>(a.osum_group, n_grp, 2 * D, 0.0, [](double x, double y) { return x + y; }, fold_lds,
                       [&](int c, double v) { a.x_sums[c] = v; });
  RL_TL(0, 7);
}

// Fold of rollout_pre's per-workgroup partial rows into the exchange record, as a launch of its own (round 4).  The
// in-launch tree needed two "last workgroup to arrive" hand-shakes, each an arrive (atomic ticket) plus a ~3.5 us round
// trip to the coherence point for the rows: ~11 us at the end of a 22.8 us launch; a kernel boundary inside the stream
// costs ~1.5 us and these few workgroups a few us.  One workgroup per 16 columns of {K column maxima | 2 D moment sums};
// thread (c, g): column c, rows g, g + 16, ... - every row of a pass requested before the first is used, sums in a fixed
// order (thread: ascending rows; then the 16 row groups ascending through LDS) => bit-reproducible; maxima are exact in
// any order.  The summation order of the fp64 moments differs from the tree's: same bars, other last bits.
__global__ __launch_bounds__(256) void rollout_fold_kernel(const float* __restrict__ colmax_partial,
                                                           const double* __restrict__ osum_partial, const int nblk,
                                                           const int K, const int D2, float* __restrict__ x_colmax,
                                                           double* __restrict__ x_sums) {
  __shared__ double lds[256];
  const int c16 = threadIdx.x & 15, g = threadIdx.x >> 4;
  const int col = blockIdx.x * 16 + c16;            // [0, K): a maximum; [K16, K16 + D2): a moment sum (K16 = K rounded up to 16)
  const int K16 = (K + 15) / 16 * 16;
  const bool is_max = blockIdx.x * 16 < K16;
  constexpr int kBatch = 16;
  if (is_max) {
    float m = -__builtin_inff();
    if (col < K) {
      for (int b0 = g; b0 < nblk; b0 += 16 * kBatch) {
        float x[kBatch];
#pragma unroll
        for (int j = 0; j < kBatch; ++j) {
          const int b = b0 + 16 * j;
          x[j] = colmax_partial[(int64_t)(b < nblk ? b : nblk - 1) * K + col];     // past the end: re-read, ignored
        }
#pragma unroll
        for (int j = 0; j < kBatch; ++j)
          if (b0 + 16 * j < nblk) m = nanmax(m, x[j]);
      }
    }
    reinterpret_cast<float*>(lds)[threadIdx.x] = m;
    __syncthreads();
    if (g == 0 && col < K) {
      for (int gg = 1; gg < 16; ++gg) m = nanmax(m, reinterpret_cast<float*>(lds)[gg * 16 + c16]);
      x_colmax[col] = (m < 1e-6f) ? 1e-6f : m;      // clamp(min=1e-6); NaN stays NaN like torch
    }
  } else {
    const int sc = col - K16;
    double a = 0.0;
    if (sc < D2) {
      for (int b0 = g; b0 < nblk; b0 += 16 * kBatch) {
        double x[kBatch];
#pragma unroll
        for (int j = 0; j < kBatch; ++j) {
          const int b = b0 + 16 * j;
          x[j] = osum_partial[(int64_t)(b < nblk ? b : nblk - 1) * D2 + sc];
        }
#pragma unroll
        for (int j = 0; j < kBatch; ++j)
          if (b0 + 16 * j < nblk) a += x[j];
      }
    }
    lds[threadIdx.x] = a;
    __syncthreads();
    if (g == 0 && sc < D2) {
      for (int gg = 1; gg < 16; ++gg) a += lds[gg * 16 + c16];
      x_sums[sc] = a;
    }
  }
}

// Round 5: every load whose address does not depend on another load is requested at the top of the kernel, before the
// first barrier (tiles of the usual sizes: the generic loops below remain for wide constraint tables / observations).
// The kernel is a handful of dependent steps separated by barriers, and the compiler cannot move a load across one: the
// constraint tile, the episode statistics, the per-env bytes and the raw observation rows each used to pay their own
// memory round trip behind the barrier in front of them - four in sequence, on the chain of every env step.
constexpr int kPreC = 4;      // constraint elements per thread held in registers: 32 rows x K <= 1024
constexpr int kPreO = 8;      // observation elements per thread: 32 rows x D <= 2048
constexpr int kPreW = 2;      // (term, env) pairs per thread: n_terms <= 16

__global__ __launch_bounds__(kThreads) void rollout_post_kernel(const PostArgs a, const TermMetaS meta) {
  extern __shared__ float smem[];
  __shared__ double red[2 * kMaxTerms * kPostRows];             // reset statistics per (term, env) of the tile
  const int K = a.K, D = a.D, nt = a.n_terms;
  float* col_rm = smem;                                         // [K]
  float* col_dp = col_rm + K;                                   // [K]
  float* tile = col_dp + K;                                     // [kPostRows*K]
  float* tmax = tile + kPostRows * K;                           // [nt*kPostRows]
  float* s_mean = tmax + nt * kPostRows;                        // [D]
  float* s_var = s_mean + D;                                    // [D]
  float* s_den = s_var + D;                                     // [D]
  __shared__ int s_off[kMaxTerms + 1];
  __shared__ float s_tot;

  RL_TL(1, 0);
  const int64_t r0 = (int64_t)blockIdx.x * kPostRows;
  const int rows = (int)((a.N - r0) < kPostRows ? (a.N - r0) : kPostRows);
  const int n_el = rows * K;
  const float* src = a.cstr + r0 * K;
  // ---- requests up front
  const bool pre_c = n_el <= kPreC * kThreads;
  const bool pre_o = a.obs_raw != nullptr && rows * D <= kPreO * kThreads;
  float pc[kPreC], po[kPreO];
  if (pre_c) {
#pragma unroll
    for (int j = 0; j < kPreC; ++j) {
      const int e = threadIdx.x + j * kThreads;
      pc[j] = e < n_el ? src[e] : 0.0f;
    }
  }
  if (pre_o) {
#pragma unroll
    for (int j = 0; j < kPreO; ++j) {
      const int e = threadIdx.x + j * kThreads;
      po[j] = 0.0f;
      if (e < rows * D) {
        const int r = fast_div(e, a.d_magic), c = e - r * D;
        po[j] = a.obs_raw[(r0 + r) * a.obs_ld + c];
      }
    }
  }
  float pw_v[kPreW], pw_p[kPreW], pw_L[kPreW];
  bool pw_rs[kPreW];
#pragma unroll
  for (int j = 0; j < kPreW; ++j) {
    const int w = threadIdx.x + j * kThreads;
    const int t = w / kPostRows, e = w - t * kPostRows;
    pw_v[j] = 0.0f, pw_p[j] = 0.0f, pw_L[j] = 1.0f, pw_rs[j] = false;
    if (w < nt * kPostRows && e < rows) {
      const int64_t i = r0 + e;
      const int64_t gi = (int64_t)t * a.N + i;
      pw_v[j] = a.ep_viol[gi];
      pw_p[j] = a.ep_prob[gi];
      pw_rs[j] = a.reset[i] != 0;
      pw_L[j] = (float)a.ep_len[i];
    }
  }
  // operands of the state derivation below (one column per thread for the maxima, two for the normaliser)
  const bool pre_k = K <= kThreads;
  float pk_m = 0.0f, pk_rm = 0.0f;
  if (pre_k && threadIdx.x < K) {
    pk_m = load_colmax(a, threadIdx.x);
    if (!a.first_call) pk_rm = a.rm[threadIdx.x];
  }
  double pn_sx[kMaxObsPerThread], pn_sxx[kMaxObsPerThread];
  float pn_mean[kMaxObsPerThread], pn_var[kMaxObsPerThread], pn_cnt = 0.0f;
  if (a.obs_raw != nullptr) {
    pn_cnt = a.obs_count[0];
#pragma unroll
    for (int q = 0; q < kMaxObsPerThread; ++q) {
      const int c = threadIdx.x + q * kThreads;
      pn_sx[q] = 0.0, pn_sxx[q] = 0.0, pn_mean[q] = 0.0f, pn_var[q] = 0.0f;
      if (c < D) {
        load_sums(a, c, &pn_sx[q], &pn_sxx[q]);
        pn_mean[q] = a.obs_mean[c], pn_var[q] = a.obs_var[c];
      }
    }
  }
  float pe_reward = 0.0f;
  bool pe_rs = false, pe_to = false;
  if (threadIdx.x < rows) {
    const int64_t i = r0 + threadIdx.x;
    pe_reward = a.reward[i];
    pe_rs = a.reset[i] != 0;
    pe_to = a.time_outs[i] != 0;
  }

  if (threadIdx.x <= nt) s_off[threadIdx.x] = meta.off[threadIdx.x];
  __syncthreads();
  // ---- new running maxima (constraint_manager.py:58-61), identical in every workgroup
  for (int c = threadIdx.x; c < K; c += kThreads) {
    int t = 0;
    while (t + 1 < nt && c >= s_off[t + 1]) ++t;
    col_rm[c] = pre_k ? running_max_from(a, pk_m, pk_rm) : derive_running_max(a, c);
    col_dp[c] = meta.dp[t];
  }
  // ---- merged observation normaliser (cleanrl/ppo.py:48-62, the op order of rms.hip)
  if (a.obs_raw != nullptr) {
    const float cnt = pn_cnt;
    const float nf = (float)a.obs_n;
    const float tot = cnt + nf;
    if (threadIdx.x == 0) s_tot = tot;
#pragma unroll
    for (int q = 0; q < kMaxObsPerThread; ++q) {          // D <= kMaxObsPerThread * kThreads (check_step)
      const int c = threadIdx.x + q * kThreads;
      if (c < D) {
        float new_mean, new_var;
        normaliser_from(a, pn_sx[q], pn_sxx[q], pn_mean[q], pn_var[q], cnt, nf, tot, &new_mean, &new_var);
        s_mean[c] = new_mean;
        s_var[c] = new_var;
        s_den[c] = sqrtf(new_var + a.obs_eps);
      }
    }
  }
  __syncthreads();
  RL_TL(1, 1);

  float* pdst = a.probs ? a.probs + r0 * K : nullptr;
  auto prob_of = [&](const float x, const int c) {
    float p = 0.0f;
    if (x > 0.0f) {
      float q = x / col_rm[c];
      q = q < 0.0f ? 0.0f : (q > 1.0f ? 1.0f : q);
      const float s = q * col_dp[c];
      p = a.min_p + s;
    }
    return p;
  };
  if (pre_c) {
#pragma unroll
    for (int j = 0; j < kPreC; ++j) {
      const int e = threadIdx.x + j * kThreads;
      if (e < n_el) {
        const float p = prob_of(pc[j], e - fast_div(e, a.k_magic) * K);
        tile[e] = p;
        if (pdst) pdst[e] = p;
      }
    }
  } else {
    for (int e = threadIdx.x; e < n_el; e += kThreads) {
      const float p = prob_of(src[e], e % K);
      tile[e] = p;
      if (pdst) pdst[e] = p;
    }
  }
  __syncthreads();

  // ---- per (term, env): max over the term's columns, episode statistics, reset statistics
  auto term_env = [&](const int w, const float v_in, const float p_in, const bool rs, const float L) {
    const int t = w / kPostRows, e = w - t * kPostRows;
    double ra = 0.0, rb = 0.0;
    if (e < rows) {
      const float* row = tile + e * K;
      float m = row[s_off[t]];
      for (int c = s_off[t] + 1; c < s_off[t + 1]; ++c) m = nanmax(m, row[c]);
      tmax[t * kPostRows + e] = m;
      const int64_t gi = (int64_t)t * a.N + r0 + e;
      float v = v_in + (m > 0.0f ? 1.0f : 0.0f);
      float p = p_in + m;
      if (rs) {          // ConstraintManager.reset (constraint_manager.py:190-211) for the envs that reset
        ra = (double)(v / L);
        rb = (double)(p / L);
        v = 0.0f, p = 0.0f;
      }
      a.ep_viol[gi] = v;
      a.ep_prob[gi] = p;
    }
    red[w] = ra;
    red[nt * kPostRows + w] = rb;
  };
#pragma unroll
  for (int j = 0; j < kPreW; ++j) {
    const int w = threadIdx.x + j * kThreads;
    if (w < nt * kPostRows) term_env(w, pw_v[j], pw_p[j], pw_rs[j], pw_L[j]);
  }
  for (int w = threadIdx.x + kPreW * kThreads; w < nt * kPostRows; w += kThreads) {     // (n_terms <= 16: never taken)
    const int t = w / kPostRows, e = w - t * kPostRows;
    const int64_t i = r0 + (e < rows ? e : 0);
    term_env(w, a.ep_viol[(int64_t)t * a.N + i], a.ep_prob[(int64_t)t * a.N + i], a.reset[i] != 0, (float)a.ep_len[i]);
  }
  __syncthreads();
  if (threadIdx.x < nt) {
    const int t = threadIdx.x;
    double sa = 0.0, sb = 0.0;
    for (int e = 0; e < kPostRows; ++e) sa += red[t * kPostRows + e], sb += red[nt * kPostRows + t * kPostRows + e];
    xwg_store(a.reset_part + (int64_t)blockIdx.x * (2 * nt + 1) + 2 * t, sa);
    xwg_store(a.reset_part + (int64_t)blockIdx.x * (2 * nt + 1) + 2 * t + 1, sb);
  }
  // ---- per env: probability, reward, dones (cat_env.py:102-107,118-121), rollout rows, reset bookkeeping
  if (threadIdx.x < rows) {
    const int e = threadIdx.x;
    float p = tmax[e];
    for (int t = 1; t < nt; ++t) p = nanmax(p, tmax[t * kPostRows + e]);
    const int64_t i = r0 + e;
    a.cstr_prob[i] = p;
    const float omp = 1.0f - p;
    float r = pe_reward * omp;
    r = (r < 0.0f) ? 0.0f : r;
    a.reward[i] = r;
    const bool rs = pe_rs;
    {   // envs of this tile that reset: rows <= 32 live in the first half of wave 0
      const unsigned long long mask = __ballot(rs);
      if (threadIdx.x == 0) xwg_store(a.reset_part + (int64_t)blockIdx.x * (2 * nt + 1) + 2 * nt, (double)__popcll(mask));
    }
    const float dn = rs ? 1.0f : p;
    if (a.dones) a.dones[i] = dn;
    if (a.rewards_t != nullptr) {
      store_plane(a.rewards_t, i, r, a.planes_f16);
      store_plane(a.dones_t1, i, dn, a.planes_f16);
      store_plane(a.true_dones_t1, i, pe_to ? 1.0f : 0.0f, a.planes_f16);
    }
    if (rs) {
      a.ep_len[i] = 0;
      if (a.zero_action) {
        for (int k = 0; k < a.A; ++k) a.action[i * a.A + k] = 0.0f, a.prev_action[i * a.A + k] = 0.0f;
      }
    }
  }
  // ---- normalised next observation rows
  if (pre_o) {
#pragma unroll
    for (int j = 0; j < kPreO; ++j) {
      const int e = threadIdx.x + j * kThreads;
      if (e < rows * D) {
        const int r = fast_div(e, a.d_magic), c = e - r * D;
        const float v = po[j] - s_mean[c];
        a.obs_out[(r0 + r) * a.obs_out_ld + c] = v / s_den[c];
      }
    }
  } else if (a.obs_raw != nullptr) {
    for (int e = threadIdx.x; e < rows * D; e += kThreads) {
      const int r = e / D, c = e - r * D;
      const float v = a.obs_raw[(r0 + r) * a.obs_ld + c] - s_mean[c];
      a.obs_out[(r0 + r) * a.obs_out_ld + c] = v / s_den[c];
    }
  }
  RL_TL(1, 2);
  // deferred tail (catppo_rollout_defer_tail): this launch ends here - no hand-shake; one workgroup of the next
  // rollout_pre launch derives the new state once more and publishes it, and folds the rows written above
  if (a.defer) return;
  // (round 4 measured this tail as a one-workgroup launch of its own, like rollout_fold_kernel behind the pre kernel:
  // 1.51 against 1.44 ms per rollout - its work is a chain of three dependent memory round trips either way, and here
  // only ONE hand-shake stands in front of it, not two - so it stays with the last workgroup to arrive)
  const bool lastp = last_block_arrives(a.ticket, gridDim.x);
  RL_TL(1, 3);
  if (!lastp) return;
  // ---- last workgroup: publish the new state, fold the reset statistics
  for (int c = threadIdx.x; c < K; c += kThreads) a.rm[c] = col_rm[c];
  if (a.obs_raw != nullptr) {
    for (int c = threadIdx.x; c < D; c += kThreads) a.obs_mean[c] = s_mean[c], a.obs_var[c] = s_var[c];
    if (threadIdx.x == 0) a.obs_count[0] = s_tot;
  }
  if (a.log_out != nullptr) fold_reset_log(a, gridDim.x, red);
  RL_TL(1, 4);
}

// the deferred tail as a launch of its own: catppo_rollout_flush, or a post call that finds a tail still pending
__global__ __launch_bounds__(kThreads) void rollout_post_tail_kernel(const PostArgs a, const int nblk) {
  __shared__ double red[kThreads];
  post_tail_deferred(a, nblk, red);
}

#ifdef ROLLOUT_TL
extern "C" int catppo_debug_rollout_tl(void* buf) {     // timeline builds only: not part of include/catppo.h
  unsigned long long* p = static_cast<unsigned long long*>(buf);
  return hipMemcpyToSymbol(HIP_SYMBOL(g_rtl), &p, sizeof(p)) == hipSuccess ? 0 : -1;
}
#endif

inline uint64_t xchg_sum_offset(int K) { return ((uint64_t)K * sizeof(float) + 15) / 16 * 16; }

// a deferred post tail that no rollout_pre launch picked up: one small launch of its own
int flush_post_tail(catppo_ctx* ctx, hipStream_t stream) {
  if (!ctx->post_tail_pending) return CATPPO_OK;
  PostArgs tl;
  memcpy(&tl, ctx->post_tail_args, sizeof(tl));
  // the tail belongs behind the post launch: on the stream that launch went to, whatever the caller passes now
  hipStream_t ts = static_cast<hipStream_t>(ctx->post_tail_stream);
  hipLaunchKernelGGL(rollout_post_tail_kernel, dim3(1), dim3(kThreads), 0, ts, tl, ctx->post_tail_nblk);
  ctx->post_tail_pending = false;
  CATPPO_CHECK_LAUNCH(ctx);
  //  a caller on ANOTHER stream is about to enqueue work that reads what the tail publishes (running maxima,
  // normaliser state): order that stream behind the tail instead of leaving it to the caller
  if (stream != ts) {
    if (ctx->ev_tail == nullptr && hipEventCreateWithFlags(&ctx->ev_tail, hipEventDisableTiming) != hipSuccess)
      return catppo_fail(ctx, CATPPO_E_HIP, "flush_post_tail: hipEventCreate failed");
    if (hipEventRecord(ctx->ev_tail, ts) != hipSuccess || hipStreamWaitEvent(stream, ctx->ev_tail, 0) != hipSuccess)
      return catppo_fail(ctx, CATPPO_E_HIP, "flush_post_tail: cannot order the calling stream behind the deferred tail");
  }
  return CATPPO_OK;
}

int check_step(catppo_ctx* ctx, const catppo_rollout_step* a, const char* fn) {
  if (!ctx) return CATPPO_E_ARG;
  if (!a) return catppo_fail(ctx, CATPPO_E_ARG, "%s: null argument block", fn);
  const bool ok = a->N >= 1 && a->A >= 1 && a->K >= 1 && a->K <= 4096 && a->n_terms >= 1 && a->n_terms <= kMaxTerms &&
                  a->D >= 0 && a->D <= kMaxObsPerThread * kThreads && a->xchg != nullptr;
  if (!ok) return catppo_fail(ctx, CATPPO_E_ARG, "%s: sizes out of range (N=%lld A=%d D=%d K=%d n_terms=%d)", fn,
                              (long long)a->N, a->A, a->D, a->K, a->n_terms);
  return CATPPO_OK;
}

}  // namespace

extern "C" uint64_t catppo_rollout_step_sizeof(void) { return sizeof(catppo_rollout_step); }
// size of one exchange record and the offset of its fp64 sums (ABI 0.6: one call for the two of ABI 0.5)
extern "C" int catppo_rollout_xchg_layout(int K, int D, uint64_t* bytes, uint64_t* sum_offset) {
  if (K < 1 || !bytes || !sum_offset) return CATPPO_E_ARG;
  *sum_offset = xchg_sum_offset(K);
  *bytes = xchg_sum_offset(K) + (uint64_t)2 * (D > 0 ? D : 1) * sizeof(double);
  return CATPPO_OK;
}

extern "C" int catppo_rollout_pre(catppo_ctx* ctx, const catppo_rollout_step* a, void* stream) {
  if (int rc = check_step(ctx, a, __func__)) return rc;
  CATPPO_CHECK_ARG(ctx, a->action_in && a->action && a->prev_action && a->episode_length && a->hard_reset &&
                            a->reward_src && a->time_outs && a->terminated && a->reset && a->reward && a->desc && a->cstr);
  CATPPO_CHECK_ARG(ctx, a->obs_raw == nullptr || (a->D >= 1 && a->obs_ld >= a->D));
  TermTable tab;
  if (const char* why = build_table(a->desc, a->n_terms, a->forces, a->forces_env_stride, a->H, a->B, a->command,
                                    a->command_ld, a->K, &tab))
    return catppo_fail(ctx, CATPPO_E_ARG, "catppo_rollout_pre: %s", why);
  // the action-rate term (C12) is evaluated BEFORE the action history is shifted inside the same launch: it reads the
  // incoming action and the still current one instead of action / prev_action
  for (int t = 0; t < tab.n; ++t) {
    catppo_term_desc& d = tab.d[t];
    if (d.x == a->action && d.y == a->prev_action) {
      const int32_t ld_action = d.x_ld;
      d.x = a->action_in, d.x_ld = a->A;
      d.y = a->action, d.y_ld = ld_action;
    } else {
      CATPPO_CHECK_ARG(ctx, d.x != a->action && d.x != a->prev_action && d.y != a->action && d.y != a->prev_action);
    }
  }
  // simulator state advance inside this launch: re-base every input that lives in the state block onto the new block
  const char* st_lo = static_cast<const char*>(a->sim_state);
  const int64_t st_bytes = a->N * a->sim_row_bytes;
  if (a->sim_src != nullptr) {
    CATPPO_CHECK_ARG(ctx, a->sim_state != nullptr && a->sim_row_bytes >= 16 && a->sim_row_bytes % 16 == 0);
    CATPPO_CHECK_ARG(ctx, (reinterpret_cast<uintptr_t>(a->sim_src) & 15) == 0 && (reinterpret_cast<uintptr_t>(a->sim_state) & 15) == 0);
  }
  if (a->sim_src != nullptr) {
    // nothing this launch WRITES may live inside the state block: the row copy would race with it and the slab would win
    // (a third-party simulator that keeps e.g. its reward or episode_length buffer inside its state tensor must pass
    // sim_src = NULL and update the block itself)
    auto inside = [&](const void* q, int64_t bytes) {
      const char* c = static_cast<const char*>(q);
      return c != nullptr && c < st_lo + st_bytes && c + bytes > st_lo;
    };
    const bool clash = inside(a->action, a->N * a->A * 4) || inside(a->prev_action, a->N * a->A * 4) ||
                       inside(a->episode_length, a->N * 8) || inside(a->reward, a->N * 4) ||
                       inside(a->cstr, a->N * (int64_t)a->K * 4) || inside(a->time_outs, a->N) ||
                       inside(a->terminated, a->N) || inside(a->reset, a->N) || inside(a->xchg, 4);
    if (clash)
      return catppo_fail(ctx, CATPPO_E_ARG, "catppo_rollout_pre: an OUTPUT of the step (action / prev_action / "
                         "episode_length / reward / cstr / time_outs / terminated / reset / xchg) lies inside the simulator "
                         "state block that sim_src is copied over; pass sim_src = NULL and advance the state yourself");
  }
  auto rebase = [&](const float* q) -> const float* {
    const char* c = reinterpret_cast<const char*>(q);
    if (a->sim_src == nullptr || c == nullptr || c < st_lo || c >= st_lo + st_bytes) return q;
    return reinterpret_cast<const float*>(static_cast<const char*>(a->sim_src) + (c - st_lo));
  };
  for (int t = 0; t < tab.n; ++t) {
    tab.d[t].x = rebase(static_cast<const float*>(tab.d[t].x));
    tab.d[t].y = rebase(static_cast<const float*>(tab.d[t].y));
  }
  const size_t lds = sizeof(float) * ((size_t)kRows * a->K + a->K);
  CATPPO_CHECK_ARG(ctx, lds <= 150 * 1024);
  // few, fatter workgroups: the partial rows (= grid size) are folded by ONE workgroup at the end of the launch, so
  // 4096 envs run as 64 workgroups x 4 tiles (64-row fold) rather than 256 x 1
  // one 16-env tile per workgroup (the term evaluation of a tile is a chain of dependent loads: tiles in sequence
  // would add their latencies), up to 1024 workgroups; their partial rows are folded by a two-level tree
  int64_t nblk = cdiv64(a->N, kRows);
  if (nblk > kMaxPreBlocks) nblk = kMaxPreBlocks;
  const int64_t n_grp = cdiv64(nblk, kFoldGroup);
  const int Dq = a->D > 0 ? a->D : 1;
  WsCarver ws(ctx);
  float* cpart = ws.take<float>((uint64_t)nblk * a->K);
  double* opart = ws.take<double>((uint64_t)nblk * 2 * Dq);
  float* cgrp = ws.take<float>((uint64_t)n_grp * a->K);
  double* ogrp = ws.take<double>((uint64_t)n_grp * 2 * Dq);
  CATPPO_NEED_WS(ctx, cpart);
  CATPPO_NEED_WS(ctx, opart);
  CATPPO_NEED_WS(ctx, cgrp);
  CATPPO_NEED_WS(ctx, ogrp);
  PreArgs p{};
  p.N = a->N, p.A = a->A, p.D = a->D, p.K = a->K;
  p.action_in = a->action_in, p.action = a->action, p.prev_action = a->prev_action;
  p.ep_len = a->episode_length, p.max_len = a->max_episode_length;
  p.hard_reset = rebase(a->hard_reset), p.hr_stride = a->hard_reset_stride;
  p.reward_src = rebase(a->reward_src), p.rw_stride = a->reward_stride;
  p.time_outs = a->time_outs, p.terminated = a->terminated, p.reset = a->reset, p.reward = a->reward;
  p.forces = rebase(a->forces), p.fstride = a->forces_env_stride, p.H = a->H, p.B = a->B;
  p.command = rebase(a->command), p.cld = a->command_ld;
  p.cstr = a->cstr;
  p.obs_raw = rebase(a->obs_raw), p.obs_ld = a->obs_ld;
  p.sim_src = static_cast<const float4*>(a->sim_src), p.sim_dst = static_cast<float4*>(a->sim_state);
  p.sim_row_q4 = (int)(a->sim_row_bytes / 16);
  p.colmax_partial = cpart, p.osum_partial = opart;
  p.colmax_group = cgrp, p.osum_group = ogrp;
  p.x_colmax = static_cast<float*>(a->xchg);
  p.x_sums = reinterpret_cast<double*>(static_cast<char*>(a->xchg) + xchg_sum_offset(a->K));
  p.ticket = ctx->tickets + catppo_ctx::kTicketPre;      // [0] launch, [1 .. 32] groups
  // CATPPO_ROLLOUT_TREE=1: fold the partial rows inside the launch (two-level tree of rounds 2-3) instead of by
  // rollout_fold_kernel - A/B switch
  static const bool tree = [] { const char* e = getenv("CATPPO_ROLLOUT_TREE"); return e && e[0] == '1'; }();
  p.tree = tree ? 1 : 0;
  p.nblk = (int)nblk;
  // the deferred tail of the previous rollout_post launch rides along as one more workgroup (same stream: it must run
  // behind that launch; another stream gets the tail as a launch of its own, on the stream of the post launch)
  PostArgs tl{};
  int extra = 0;
  if (ctx->post_tail_pending) {
    // (the in-launch fold tree of CATPPO_ROLLOUT_TREE=1 writes the exchange record the tail still reads: no ride then)
    if (ctx->post_tail_stream == stream && !tree) {
      memcpy(&tl, ctx->post_tail_args, sizeof(tl));
      p.tail_nblk = ctx->post_tail_nblk;
      ctx->post_tail_pending = false;
      extra = 1;
    } else if (int rc = flush_post_tail(ctx, static_cast<hipStream_t>(stream))) {
      return rc;
    }
  }
  hipLaunchKernelGGL(rollout_pre_kernel, dim3((unsigned)nblk + extra), dim3(kThreads), lds, static_cast<hipStream_t>(stream),
                     tab, p, tl);
  CATPPO_CHECK_LAUNCH(ctx);
  if (!tree) {
    const int K16 = (a->K + 15) / 16 * 16, D2 = a->obs_raw != nullptr ? 2 * a->D : 0;
    hipLaunchKernelGGL(rollout_fold_kernel, dim3((unsigned)((K16 + D2 + 15) / 16)), dim3(256), 0,
                       static_cast<hipStream_t>(stream), (const float*)cpart, (const double*)opart, (int)nblk, a->K, D2,
                       p.x_colmax, p.x_sums);
    CATPPO_CHECK_LAUNCH(ctx);
  }
  return CATPPO_OK;
}

extern "C" int catppo_rollout_post(catppo_ctx* ctx, const catppo_rollout_step* a, void* stream) {
  if (int rc = check_step(ctx, a, __func__)) return rc;
  CATPPO_CHECK_ARG(ctx, a->cstr && a->term_off && a->term_dp && a->rm && a->cstr_prob && a->ep_viol && a->ep_prob &&
                            a->reward && a->reset && a->time_outs && a->episode_length);
  CATPPO_CHECK_ARG(ctx, a->rewards_t == nullptr || (a->dones_t1 && a->true_dones_t1));
  CATPPO_CHECK_ARG(ctx, a->plane_dtype == CATPPO_F32 || a->plane_dtype == CATPPO_F16);
  CATPPO_CHECK_ARG(ctx, a->obs_raw == nullptr || (a->obs_mean && a->obs_var && a->obs_count && a->obs_out &&
                                                  a->obs_ld >= a->D && a->obs_out_ld >= a->D && a->obs_rows_total >= 1.0));
  CATPPO_CHECK_ARG(ctx, !a->zero_action_on_reset || (a->action && a->prev_action));
  TermMetaS meta;
  int prev = 0;
  for (int t = 0; t <= a->n_terms; ++t) {
    if (a->term_off[t] < prev || a->term_off[t] > a->K) return catppo_fail(ctx, CATPPO_E_ARG, "rollout_post: term_off not monotone");
    prev = meta.off[t] = a->term_off[t];
  }
  CATPPO_CHECK_ARG(ctx, a->term_off[0] == 0 && a->term_off[a->n_terms] == a->K);
  for (int t = 0; t < a->n_terms; ++t) meta.dp[t] = a->term_dp[t];
  const int nt = a->n_terms, K = a->K, D = a->D;
  const size_t lds = sizeof(float) * ((size_t)2 * K + (size_t)kPostRows * K + (size_t)nt * kPostRows + (size_t)3 * D);
  if (lds > 140 * 1024) return catppo_fail(ctx, CATPPO_E_ARG, "rollout_post: K=%d / D=%d too wide for one LDS tile", K, D);
  const int nblk = (int)cdiv64(a->N, kPostRows);
  // a tail still pending (two post calls with no pre call between them): it reads the state this launch is about to read
  if (int rc = flush_post_tail(ctx, static_cast<hipStream_t>(stream))) return rc;
  const bool defer = ctx->rollout_defer;
  double* rpart = nullptr;
  if (defer) {
    // the rows outlive this call (the workspace is anybody's between two launches): a buffer of the context
    const uint64_t need = (uint64_t)nblk * (nt * 2 + 1) * sizeof(double);
    if (need > ctx->post_rpart_bytes) {
      if (ctx->post_rpart) (void)hipFree(ctx->post_rpart);     // (synchronises: nobody reads the old rows any more)
      ctx->post_rpart = nullptr, ctx->post_rpart_bytes = 0;
      if (hipMalloc(reinterpret_cast<void**>(&ctx->post_rpart), need) != hipSuccess)
        return catppo_fail(ctx, CATPPO_E_HIP, "catppo_rollout_post: hipMalloc of %llu B failed", (unsigned long long)need);
      ctx->post_rpart_bytes = need;
    }
    rpart = ctx->post_rpart;
  } else {
    WsCarver ws(ctx);
    rpart = ws.take<double>((uint64_t)nblk * (nt * 2 + 1));
    CATPPO_NEED_WS(ctx, rpart);
  }
  PostArgs p{};
  p.N = a->N, p.A = a->A, p.D = D, p.K = K, p.n_terms = nt;
  p.cstr = a->cstr;
  p.min_p = a->min_p, p.tau = a->tau, p.one_minus_tau = a->one_minus_tau, p.first_call = a->first_call;
  p.rm = a->rm, p.reward = a->reward, p.reset = a->reset, p.time_outs = a->time_outs;
  p.cstr_prob = a->cstr_prob, p.dones = a->dones, p.ep_viol = a->ep_viol, p.ep_prob = a->ep_prob, p.probs = a->probs;
  p.ep_len = a->episode_length, p.action = a->action, p.prev_action = a->prev_action;
  p.zero_action = a->zero_action_on_reset;
  p.log_prev = a->log_prev, p.log_out = a->log_out;
  p.rewards_t = a->rewards_t, p.dones_t1 = a->dones_t1, p.true_dones_t1 = a->true_dones_t1;
  p.planes_f16 = a->plane_dtype == CATPPO_F16;
  p.obs_raw = a->obs_raw, p.obs_ld = a->obs_ld;
  p.obs_mean = a->obs_mean, p.obs_var = a->obs_var, p.obs_count = a->obs_count, p.obs_eps = a->obs_eps;
  p.obs_n = a->obs_rows_total;
  p.obs_out = a->obs_out, p.obs_out_ld = a->obs_out_ld;
  const void* xbase = (a->xchg_records > 1 && a->xchg_gathered != nullptr) ? a->xchg_gathered : a->xchg;
  p.x_colmax = static_cast<const float*>(xbase);
  p.x_sums = reinterpret_cast<const double*>(static_cast<const char*>(xbase) + xchg_sum_offset(K));
  p.x_records = (a->xchg_records > 1 && a->xchg_gathered != nullptr) ? a->xchg_records : 1;
  p.x_stride = (int64_t)(xchg_sum_offset(K) + (uint64_t)2 * (a->D > 0 ? a->D : 1) * sizeof(double));
  p.reset_part = rpart;
  p.ticket = ctx->tickets + catppo_ctx::kTicketPost;
  p.defer = defer ? 1 : 0;
  p.d_magic = catppo_div_magic((uint32_t)(D > 0 ? D : 1)), p.k_magic = catppo_div_magic((uint32_t)K);
  if (lds > 64 * 1024)
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(rollout_post_kernel),
                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  hipLaunchKernelGGL(rollout_post_kernel, dim3(nblk), dim3(kThreads), lds, static_cast<hipStream_t>(stream), p, meta);
  CATPPO_CHECK_LAUNCH(ctx);
  if (defer) {
    memcpy(ctx->post_tail_args, &p, sizeof(p));
    ctx->post_tail_nblk = nblk;
    ctx->post_tail_stream = stream;
    ctx->post_tail_pending = true;
  }
  return CATPPO_OK;
}

extern "C" int catppo_rollout_defer_tail(catppo_ctx* ctx, int on, void* stream) {
  if (!ctx) return CATPPO_E_ARG;
  CATPPO_CHECK_ARG(ctx, on == 0 || on == 1 || on == -1);
  if (on == -1) return flush_post_tail(ctx, static_cast<hipStream_t>(stream));      // flush only, the mode stays (ABI 0.5: catppo_rollout_flush)
  ctx->rollout_defer = on != 0;
  return on ? CATPPO_OK : flush_post_tail(ctx, static_cast<hipStream_t>(stream));
}
