// Internal helpers shared by the libcatppo translation units (gfx950 only).
#pragma once

#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <cstring>

#include "../../include/catppo.h"

struct catppo_ctx {
  int device = 0;
  int n_cu = 256;
  void* ws = nullptr;        // generic workspace (device)
  uint64_t ws_bytes = 0;
  // side stream + events: the weight-gradient GEMMs of the backward pass run beside the
  // data-gradient chain (fork/join around catppo_ppo_minibatch_grad, capturable in a hipGraph)
  hipStream_t side = nullptr;
  bool use_side = false;     // CATPPO_SIDE_STREAM=1 forks the weight-gradient GEMMs (measured slower)
  hipEvent_t ev_fork[CATPPO_MAX_HIDDEN + 1] = {};
  hipEvent_t ev_join = nullptr;
  hipEvent_t ev_tail = nullptr;     // rollout.hip: orders another stream behind a deferred post tail (created on first use)
  // hipGraphs captured through catppo_graph_begin / _end (index = graph id; destroyed slots are null)
  static constexpr int kMaxGraphs = 64;
  hipGraphExec_t graphs[kMaxGraphs] = {};
  bool capturing = false;
  // RCCL communicator (comm.hip; librccl is dlopen'ed on first use)
  void* comm = nullptr;      // ncclComm_t
  int comm_rank = 0, comm_world = 0;
  // catppo_set_grad_overlap: the gradient all-reduce of an optimiser step runs per layer bucket on the side stream,
  // under the backward launches of the layers below (effective only while a communicator exists)
  // 0 off | 1 per-layer buckets (round 4: extra fold launches) | 2 "tail" (round 5: no extra launch - everything but the
  // first layer is reduced on the side stream under the final fold launch)
  int grad_overlap = 0;
  // device-side completion tickets of the "last workgroup folds" kernels (zero between launches)
  unsigned int* tickets = nullptr;   // [kTickets]
  static constexpr int kTickets = 64;
  static constexpr int kTicketPre = 0;     // rollout_pre: [0] launch-wide + [1..32] per workgroup group
  static constexpr int kTicketPost = 40;   // rollout_post
  // catppo_rollout_defer_tail (rollout.hip): the one-workgroup tail of rollout_post (publish the new running maxima /
  // normaliser state, fold the reset statistics) rides in the NEXT rollout_pre launch instead of standing at the end of
  // the post launch, behind a "last workgroup arrives" hand-shake, in front of the policy forward
  bool rollout_defer = false;
  bool post_tail_pending = false;
  int post_tail_nblk = 0;                               // workgroups of the post launch whose tail is pending
  void* post_tail_stream = nullptr;
  alignas(16) unsigned char post_tail_args[512] = {0};  // that launch's PostArgs
  double* post_rpart = nullptr;                         // its reset-statistics rows: owned (the workspace is reused by
  uint64_t post_rpart_bytes = 0;                        // whatever runs between the two launches), grown on demand
  // catppo_debug_clip_branches: when set, the head / loss kernels of the next gradient calls write the clip branch every
  // sample took ([2][M] int32: surrogate codes, then value-loss codes; 0 inside, 1 below, 2 above the clip range)
  int32_t* branch_out = nullptr;
  char err[512] = {0};
  // catppo_plan_log: when on, the dispatch code of the MLP entry points appends one line per launch decision (which
  // kernel a shape gets, and why) - written AT the decision sites, so it cannot drift from what runs
  bool plan_on = false;
  int plan_len = 0;
  char plan[8192] = {0};
};

inline void catppo_plan_note(catppo_ctx* ctx, const char* fmt, ...) {
  if (!ctx || !ctx->plan_on || ctx->plan_len >= (int)sizeof(ctx->plan) - 2) return;
  va_list ap;
  va_start(ap, fmt);
  const int n = vsnprintf(ctx->plan + ctx->plan_len, sizeof(ctx->plan) - ctx->plan_len - 1, fmt, ap);
  va_end(ap);
  if (n > 0) ctx->plan_len += n < (int)sizeof(ctx->plan) - ctx->plan_len - 1 ? n : (int)sizeof(ctx->plan) - ctx->plan_len - 2;
  ctx->plan[ctx->plan_len++] = '\n';
  ctx->plan[ctx->plan_len] = 0;
}

inline int catppo_fail(catppo_ctx* ctx, int code, const char* fmt, ...) {
  if (ctx) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(ctx->err, sizeof(ctx->err), fmt, ap);
    va_end(ap);
  }
  return code;
}

#define CATPPO_CHECK_ARG(ctx, cond)                                                      \
  do {                                                                                   \
    if (!(cond)) return catppo_fail(ctx, CATPPO_E_ARG, "%s: bad argument: %s", __func__, #cond); \
  } while (0)

#define CATPPO_CHECK_LAUNCH(ctx)                                                         \
  do {                                                                                   \
    hipError_t e__ = hipGetLastError();                                                  \
    if (e__ != hipSuccess)                                                               \
      return catppo_fail(ctx, CATPPO_E_HIP, "%s: launch failed: %s", __func__, hipGetErrorString(e__)); \
  } while (0)

// carve `bytes` (rounded to 256) from the workspace; returns nullptr when it does not fit
struct WsCarver {
  char* base;
  uint64_t cap, used = 0;
  explicit WsCarver(catppo_ctx* c) : base(static_cast<char*>(c->ws)), cap(c->ws_bytes) {}
  template <typename T>
  T* take(uint64_t count) {
    uint64_t bytes = (count * sizeof(T) + 255) & ~uint64_t(255);
    if (used + bytes > cap) return nullptr;
    T* p = reinterpret_cast<T*>(base + used);
    used += bytes;
    return p;
  }
};

#define CATPPO_NEED_WS(ctx, ptr)                                                          \
  do {                                                                                    \
    if (!(ptr))                                                                           \
      return catppo_fail(ctx, CATPPO_E_WORKSPACE, "%s: workspace too small (%llu B); call catppo_reserve", \
                         __func__, (unsigned long long)(ctx)->ws_bytes);                  \
  } while (0)

static inline int64_t cdiv64(int64_t a, int64_t b) { return (a + b - 1) / b; }

// Division of a small index by a run-time width without the ~35-instruction integer-division sequence (no hardware
// divider on gfx9): q = mulhi(e, magic), magic = floor(2^32 / d) + 1.  Exact whenever e * d < 2^32 - here e < 2^17 and
// d <= 4096 (checked exhaustively over that whole range on the host); d == 1 is encoded as magic 0.  The launch-bound
// kernels of the env step (one wave per SIMD: every instruction is latency) spent ~20 such divisions per thread.
static inline uint32_t catppo_div_magic(uint32_t d) { return d <= 1 ? 0u : (uint32_t)((uint64_t(1) << 32) / d) + 1u; }
__device__ __forceinline__ int fast_div(int e, uint32_t magic) { return magic ? (int)__umulhi((uint32_t)e, magic) : e; }

// cat_terms.hip: evaluate the term table into cstr[N,K]; with `colmax_partial` ([<=256][K], may be null) every
// workgroup also writes the column maxima of the rows it produced, *nblk_out = number of partial rows
int catppo_internal_launch_terms(catppo_ctx* ctx, const catppo_term_desc* desc, int n_terms, int64_t N,
                                 const float* forces, int64_t forces_env_stride, int H, int B, const float* command,
                                 int command_ld, float* cstr, int K, float* colmax_partial, int* nblk_out,
                                 hipStream_t stream);

// comm.hip: SUM all-reduce of n ranges [off[i], off[i] + cnt[i]) of one fp32 buffer as ONE grouped RCCL operation on `stream`
int catppo_internal_allreduce_ranges(catppo_ctx* ctx, float* base, const int64_t* off, const int64_t* cnt, int n,
                                     hipStream_t stream);

// NaN-propagating max (torch.max semantics): once NaN, stays NaN
__device__ __forceinline__ float nanmax(float m, float x) { return (x > m || x != x) ? x : m; }
