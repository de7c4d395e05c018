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

using gemm::Operands;
using gemm::Params;

constexpr int kMaxA = 16;
constexpr float kHalfLog2Pi = 0.91893853320467274178f;   // log(sqrt(2*pi))
constexpr float kEntConst = 1.41893853320467274178f;     // 0.5 + 0.5*log(2*pi)
constexpr int kHeadRowsPerBlock = 32;                    // head_loss row tile (16 rows when the last layer is 512 wide)
inline int head_rows(int HL) { return HL > 256 ? 16 : 32; }   // both tiles + head weights must fit 160 KB of LDS
constexpr int kGatherRows = 64;

// ------------------------------------------------------------------------------- layout
int layout_of(const catppo_mlp_shape* s, catppo_mlp_layout* L) {
  if (!s || !L) return CATPPO_E_ARG;
  if (s->obs_dim < 1 || s->act_dim < 1 || s->act_dim >= kMaxA) return CATPPO_E_ARG;  // slot act_dim = critic
  if (s->mfma_bf16 < 0 || s->mfma_bf16 > 2) return CATPPO_E_ARG;
  if (s->n_hidden < 1 || s->n_hidden > CATPPO_MAX_HIDDEN) return CATPPO_E_ARG;
  for (int l = 0; l < s->n_hidden; ++l)
    if (s->hidden[l] < 64 || s->hidden[l] % 64 != 0 || s->hidden[l] > 4096) return CATPPO_E_ARG;
  const int hl = s->hidden[s->n_hidden - 1];
  if (hl != 64 && hl != 128 && hl != 256 && hl != 512) return CATPPO_E_ARG;  // head kernel widths
  memset(L, 0, sizeof(*L));
  const int nl = s->n_hidden;
  L->obs_pad = (s->obs_dim + 15) / 16 * 16;
  auto r4 = [](int64_t x) { return (x + 3) / 4 * 4; };
  int64_t off = 0, np = 0;
  L->off_logstd = off;
  off += r4(s->act_dim);
  np += s->act_dim;
  for (int l = 0; l <= nl; ++l) L->in_dim[l] = l == 0 ? L->obs_pad : s->hidden[l - 1];
  for (int net = 0; net < 2; ++net) {
    for (int l = 0; l <= nl; ++l) {
      const int out = l < nl ? s->hidden[l] : (net == 0 ? 1 : s->act_dim);
      L->out_dim[net][l] = out;
      L->off_w[net][l] = off;
      off += r4((int64_t)out * L->in_dim[l]);
      L->off_b[net][l] = off;
      off += r4(out);
      np += (int64_t)out * (l == 0 ? s->obs_dim : L->in_dim[l]) + out;
    }
  }
  L->n_flat = off;
  L->n_params = np;
  return CATPPO_OK;
}

// ------------------------------------------------------------------------------- workspace
struct MlpWs {
  float* xmb;              // [M, Dp]          gathered observations
  float* act;              // [M, A]
  float* scal;             // [4][M]           oldlogp, adv, ret_n, val_n
  double* adv_part;        // [nb_gather][2]
  float* H[2][CATPPO_MAX_HIDDEN];   // activations per net / hidden layer [M, h_l]
  float* dZ[2][CATPPO_MAX_HIDDEN];  // pre-activation gradients
  float* wpart[CATPPO_MAX_HIDDEN];   // split-K partial weight gradients per layer (both nets)
  float* bpart[CATPPO_MAX_HIDDEN];   // split-K partial bias gradients per layer
  float* head_w;           // [nb_head][(A+1)*HL]
  float* head_s;           // [nb_head][kHeadScalars]
  double* norm_part;       // [kNormSlots]: squared-norm partials emitted by the launches that fold the gradient (NormEmit)
  uint64_t bytes;
};
constexpr int kHeadDiag = 8;
constexpr int kNormBlocks = 256;
constexpr int kNormSlots = 256 * 24;     // 256 workgroups per segment x kMaxSegs (static_assert at its definition)
inline int head_scalars(int A) { return 2 * A + 1 + kHeadDiag; }  // db4a[A], db4c, dlogstd[A], diag[8]

// split-K count cap: the partial sums are written once and re-read by the fold, so a layer may use as many
// splits as keep its partials under ~8 MB (32 for a 256x256 layer, 64+ for the narrow first layer, whose
// 8 tiles would otherwise leave most CUs with one latency-bound workgroup).
constexpr int kMinSplitCap = 32, kMaxSplitCap = 128;
inline int split_cap(int out, int in) {
  const int64_t bytes_per_split = 2 * (int64_t)out * in * (int64_t)sizeof(float);
  int64_t cap = (8 << 20) / bytes_per_split;
  cap = cap < kMinSplitCap ? kMinSplitCap : cap;
  return (int)(cap > kMaxSplitCap ? kMaxSplitCap : cap);
}

bool carve(const catppo_mlp_shape* s, const catppo_mlp_layout& L, int64_t M, bool training, char* base,
           uint64_t cap, MlpWs* w) {
  uint64_t used = 0;
  bool ok = true;
  auto take = [&](uint64_t bytes) -> char* {
    bytes = (bytes + 255) & ~uint64_t(255);
    char* p = base ? base + used : nullptr;
    used += bytes;
    if (base && used > cap) ok = false;
    return p;
  };
  const int nl = s->n_hidden, A = s->act_dim;
  const int64_t nbg = cdiv64(M, kGatherRows), nbh = cdiv64(M, 16);   // upper bound of head_loss blocks
  // reduction partials first: the non-MLP calls use the front of the workspace too, but never
  // concurrently with an MLP call on the same stream
  w->xmb = (float*)take(sizeof(float) * M * L.obs_pad);
  w->act = (float*)take(sizeof(float) * M * A);
  w->scal = (float*)take(sizeof(float) * 4 * M);
  w->adv_part = (double*)take(sizeof(double) * 2 * nbg);
  for (int net = 0; net < 2; ++net)
    for (int l = 0; l < nl; ++l) w->H[net][l] = (float*)take(sizeof(float) * M * s->hidden[l]);
  if (training) {
    for (int net = 0; net < 2; ++net)
      for (int l = 0; l < nl; ++l) w->dZ[net][l] = (float*)take(sizeof(float) * M * s->hidden[l]);
    // one partial buffer per layer: all weight-gradient partials of a minibatch are folded by ONE launch
    for (int l = 0; l < nl; ++l) {
      const int cap = split_cap(s->hidden[l], L.in_dim[l]);
      w->wpart[l] = (float*)take(sizeof(float) * 2 * cap * (int64_t)s->hidden[l] * L.in_dim[l]);
      w->bpart[l] = (float*)take(sizeof(float) * 2 * cap * s->hidden[l]);
    }
    w->head_w = (float*)take(sizeof(float) * nbh * (A + 1) * s->hidden[nl - 1]);
    w->head_s = (float*)take(sizeof(float) * nbh * head_scalars(A));
  }
  w->norm_part = (double*)take(sizeof(double) * kNormSlots);
  w->bytes = used;
  return ok;
}

// ------------------------------------------------------------------------------- GEMM launch
constexpr int kSmallRows = 4096;    // see launch_dw_dx_pair

static int env_int(const char* name, int dflt) {
  const char* e = getenv(name);
  return e ? atoi(e) : dflt;
}
// CATPPO_XCD_LEGACY=1: the round-2 workgroup -> tile order (A/B of the launch-wide XCD mapping, see gemm::xcd_tile_of)
static int xcd_legacy() {
  static const int v = env_int("CATPPO_XCD_LEGACY", 0);
  return v;
}

// fp32 launch with a wider contraction slab (latency-bound small-M launches: fewer global round trips per tile)
template <int BM, int BN, bool A_KC, bool B_KC, int EPI, int BKT>
void launch_gemm_bk(const Params& p, hipStream_t s) {
  dim3 grid(((p.J + BN - 1) / BN) * ((p.I + BM - 1) / BM), 1, p.nets * p.splits);
  constexpr size_t lds = gemm::smem_bytes<BM, BN, A_KC, B_KC, BKT>();
  auto kern = gemm::gemm_f32_kernel<BM, BN, A_KC, B_KC, EPI, BKT>;
  if (lds > 64 * 1024)     // per call, not once per process: the attribute belongs to the current device's copy of the kernel
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  kern<<<grid, dim3(256), lds, s>>>(p);
}

template <int BM, int BN, bool A_KC, bool B_KC, int EPI>
void launch_gemm(const Params& p, hipStream_t s, int prec, size_t lds_pad = 0) {
  dim3 grid(((p.J + BN - 1) / BN) * ((p.I + BM - 1) / BM), 1, p.nets * p.splits);   // 1-D tile index, see kernel
  const size_t lds = gemm::smem_bytes<BM, BN, A_KC, B_KC>() + lds_pad;
  if (lds_pad && prec == 0) {   // residency experiment (CATPPO_FWD_LDS_PAD): fp32 forward GEMMs only
    auto kern = gemm::gemm_f32_kernel<BM, BN, A_KC, B_KC, EPI>;
    if (lds > 64 * 1024)
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    kern<<<grid, dim3(256), lds, s>>>(p);
    return;
  }
  if (prec == 2)
    gemm::gemm_f32_kernel<BM, BN, A_KC, B_KC, EPI, gemm::BK, 2><<<grid, dim3(256), lds, s>>>(p);
  else if (prec == 1)
    gemm::gemm_f32_kernel<BM, BN, A_KC, B_KC, EPI, gemm::BK, 1><<<grid, dim3(256), lds, s>>>(p);
  else
    gemm::gemm_f32_kernel<BM, BN, A_KC, B_KC, EPI><<<grid, dim3(256), lds, s>>>(p);
}

// Tile choice from tools/gemm_probe on MI355X (M=16384, both nets per launch): 128x128 pays only when
// the contraction is long enough to amortise its heavier epilogue and there are >= 1.5 workgroups per
// CU; the data-gradient form (aux read + store epilogue) is always better with 64x64 tiles.
template <bool A_KC, bool B_KC, int EPI>
void launch_gemm_auto(const Params& p, hipStream_t s, int prec) {
  const int64_t big = (int64_t)((p.I + 127) / 128) * ((p.J + 127) / 128) * p.nets * p.splits;
  const int kc = EPI == gemm::EPI_PARTIAL ? p.kc_per_split : p.Kc;
  // weight gradients pick their split count to fill the chip, so only the shape matters there
  const bool use_big = EPI != gemm::EPI_MUL_DELU && p.I >= 128 && p.J >= 128 && kc >= 256 &&
                       (EPI == gemm::EPI_PARTIAL || big >= 384);
  if constexpr (EPI == gemm::EPI_BIAS_ELU) {
    // experiment hooks (A/B on the GPU box): CATPPO_FWD_TILE = 64x64 | 64x128 | 128x64, CATPPO_FWD_LDS_PAD = bytes of
    // unused LDS per workgroup (caps the number of resident workgroups per CU => the grid runs in several rounds)
    static const int tile_sel = [] {
      const char* e = getenv("CATPPO_FWD_TILE");
      if (!e) return 0;
      if (!strcmp(e, "64x64")) return 1;
      if (!strcmp(e, "64x128")) return 2;
      if (!strcmp(e, "128x64")) return 3;
      return 0;
    }();
    static const size_t pad = [] {
      const char* e = getenv("CATPPO_FWD_LDS_PAD");
      return e ? (size_t)atol(e) : (size_t)0;
    }();
    if (use_big && (tile_sel || pad)) {
      if (tile_sel == 1) launch_gemm<64, 64, A_KC, B_KC, EPI>(p, s, prec, pad);
      else if (tile_sel == 2) launch_gemm<64, 128, A_KC, B_KC, EPI>(p, s, prec, pad);
      else if (tile_sel == 3) launch_gemm<128, 64, A_KC, B_KC, EPI>(p, s, prec, pad);
      else launch_gemm<128, 128, A_KC, B_KC, EPI>(p, s, prec, pad);
      return;
    }
  }
  if (use_big) {
    launch_gemm<128, 128, A_KC, B_KC, EPI>(p, s, prec);
    return;
  }
  if constexpr (EPI != gemm::EPI_PARTIAL) {
    static const int small_bk = env_int("CATPPO_SMALL_BK", 64);      // 16 | 32 | 64, see kSmallRows
    if (prec == 0 && small_bk > 16 && p.I <= kSmallRows && p.Kc % small_bk == 0 && p.Kc >= 2 * small_bk) {
      if (small_bk == 32) launch_gemm_bk<64, 64, A_KC, B_KC, EPI, 32>(p, s);
      else launch_gemm_bk<64, 64, A_KC, B_KC, EPI, 64>(p, s);
      return;
    }
  }
  launch_gemm<64, 64, A_KC, B_KC, EPI>(p, s, prec);
}

template <int BM, int BN>
constexpr int tiles_of(const Params& p) { return ((p.J + BN - 1) / BN) * ((p.I + BM - 1) / BM); }

// weight gradient (problem 0: 128x128 tiles when the layer allows, else 64x64) + data gradient (problem 1: 64x128
// tiles - measured best inside the pair on MI355X, 350 -> 338 us per minibatch against 64x64 - or 64x64 for
// layers narrower than 128) of one layer in one launch
template <int BM0, int BN0, int BM1, int BN1>
void launch_pair_tiles(const Params& pw, const Params& px, hipStream_t s, int prec) {
  const int t0 = tiles_of<BM0, BN0>(pw), n0 = t0 * pw.nets * pw.splits;
  const int t1 = tiles_of<BM1, BN1>(px), n1 = t1 * px.nets * px.splits;
  constexpr size_t lds0 = gemm::smem_bytes<BM0, BN0, false, false>();
  constexpr size_t lds1 = gemm::smem_bytes<BM1, BN1, true, false>();
  constexpr size_t lds = lds0 > lds1 ? lds0 : lds1;
  if (prec == 2)
    gemm::gemm_pair_kernel<BM0, BN0, false, false, gemm::EPI_PARTIAL, BM1, BN1, true, false, gemm::EPI_MUL_DELU, 2>
        <<<dim3(n0 + n1), dim3(256), lds, s>>>(pw, px, t0, n0, t1);
  else if (prec == 1)
    gemm::gemm_pair_kernel<BM0, BN0, false, false, gemm::EPI_PARTIAL, BM1, BN1, true, false, gemm::EPI_MUL_DELU, 1>
        <<<dim3(n0 + n1), dim3(256), lds, s>>>(pw, px, t0, n0, t1);
  else
    gemm::gemm_pair_kernel<BM0, BN0, false, false, gemm::EPI_PARTIAL, BM1, BN1, true, false, gemm::EPI_MUL_DELU>
        <<<dim3(n0 + n1), dim3(256), lds, s>>>(pw, px, t0, n0, t1);
}

// Minibatches of at most kSmallRows rows leave every CU with one or two workgroups: each wave is alone on its SIMD and
// every contraction slab costs a full global round trip.  There the weight gradient runs on 64x64 tiles (4x the
// workgroups of the 128x128 choice: 2048 rows, 256x512 layer: 35 -> 27 us for the pair) and the forward GEMMs walk
// the contraction in 64-wide slabs (4x fewer round trips; 112 -> 100 us per optimiser step together; measured with
// CATPPO_DW_SMALL_TILE / CATPPO_SMALL_BK, which remain as switches).
void launch_dw_dx_pair(const Params& pw, const Params& px, hipStream_t s, int prec) {
  static const int small_tile = env_int("CATPPO_DW_SMALL_TILE", 1);
  const bool big = pw.I >= 128 && pw.J >= 128 && pw.kc_per_split >= 256 &&   // launch_gemm_auto's rule for EPI_PARTIAL
                   !(px.I <= kSmallRows && small_tile);
  const bool wide = px.J >= 128;
  if (big && wide) launch_pair_tiles<128, 128, 64, 128>(pw, px, s, prec);
  else if (big) launch_pair_tiles<128, 128, 64, 64>(pw, px, s, prec);
  else if (wide) launch_pair_tiles<64, 64, 64, 128>(pw, px, s, prec);
  else launch_pair_tiles<64, 64, 64, 64>(pw, px, s, prec);
}

// hidden-layer forward for `nets` networks starting at net index net0
void forward_hidden(const catppo_mlp_shape* sh, const catppo_mlp_layout& L, const float* params, const float* x,
                    int64_t M, const MlpWs& w, int net0, int nets, hipStream_t s, int n_layers = -1) {
  if (n_layers < 0) n_layers = sh->n_hidden;
  for (int l = 0; l < n_layers; ++l) {
    Params p{};
    p.xcd_legacy = xcd_legacy();
    p.nets = nets;
    p.splits = 1;
    p.I = (int)M;
    p.J = sh->hidden[l];
    p.Kc = L.in_dim[l];
    p.lda = L.in_dim[l];
    p.ldb = L.in_dim[l];
    p.ldc = sh->hidden[l];
    for (int n = 0; n < nets; ++n) {
      const int net = net0 + n;
      p.op[n].A = l == 0 ? x : w.H[net][l - 1];
      p.op[n].B = params + L.off_w[net][l];
      p.op[n].bias = params + L.off_b[net][l];
      p.op[n].C = w.H[net][l];
    }
    launch_gemm_auto<true, true, gemm::EPI_BIAS_ELU>(p, s, sh->mfma_bf16);
  }
}

// ------------------------------------------------------------------------------- wave helpers
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int m = 32; m >= 1; m >>= 1) v += __shfl_xor(v, m, 64);
  return v;
}
__device__ __forceinline__ double wave_sum_d(double v) {
#pragma unroll
  for (int m = 32; m >= 1; m >>= 1) v += __shfl_xor(v, m, 64);
  return v;
}

// CPL consecutive floats per lane as ONE store instruction (rows are 16-B aligned: HL % 64 == 0)
template <int CPL>
__device__ __forceinline__ void store_vec(float* p, const float (&v)[CPL]) {
  if constexpr (CPL == 1) {
    p[0] = v[0];
  } else if constexpr (CPL == 2) {
    *reinterpret_cast<float2*>(p) = make_float2(v[0], v[1]);
  } else {
#pragma unroll
    for (int q = 0; q < CPL / 4; ++q)
      reinterpret_cast<float4*>(p)[q] = make_float4(v[4 * q], v[4 * q + 1], v[4 * q + 2], v[4 * q + 3]);
  }
}

// the same store written through to memory (global stores only): activation-sized outputs that the NEXT launch
// reads should not sit dirty in L2 until the kernel boundary flushes them (see gemm_f32.h epilogue)
template <int CPL>
__device__ __forceinline__ void store_vec_wt(float* p, const float (&v)[CPL]) {
  using f4v = __attribute__((ext_vector_type(4))) float;
  using f2v = __attribute__((ext_vector_type(2))) float;
  if constexpr (CPL == 1) {
    asm volatile("global_store_dword %0, %1, off sc1\n\ts_nop 0" ::"v"(p), "v"(v[0]) : "memory");
  } else if constexpr (CPL == 2) {
    f2v o;
    o.x = v[0], o.y = v[1];
    asm volatile("global_store_dwordx2 %0, %1, off sc1\n\ts_nop 0" ::"v"(p), "v"(o) : "memory");
  } else {
#pragma unroll
    for (int q = 0; q < CPL / 4; ++q) {
      f4v o;
      o.x = v[4 * q], o.y = v[4 * q + 1], o.z = v[4 * q + 2], o.w = v[4 * q + 3];
      float* dst = p + 4 * q;
      asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 2" ::"v"(dst), "v"(o) : "memory");
    }
  }
}

template <int CPL>
__device__ __forceinline__ void load_vec(const float* p, float (&v)[CPL]) {
  if constexpr (CPL == 1) {
    v[0] = p[0];
  } else if constexpr (CPL == 2) {
    const float2 t = *reinterpret_cast<const float2*>(p);
    v[0] = t.x, v[1] = t.y;
  } else {
#pragma unroll
    for (int q = 0; q < CPL / 4; ++q) {
      const float4 t = reinterpret_cast<const float4*>(p)[q];
      v[4 * q] = t.x, v[4 * q + 1] = t.y, v[4 * q + 2] = t.z, v[4 * q + 3] = t.w;
    }
  }
}

// Sixteen per-lane partial values -> their 64-lane totals with 17 cross-lane exchanges instead of
// 16 x 6: every butterfly step halves the number of live values (the lane keeps the half selected by
// its own bit and hands the other half to its partner).  Afterwards the total of value j sits in the
// four lanes l with slot(l) == j, slot(l) = 8*bit5 + 4*bit4 + 2*bit3 + bit2.
__device__ __forceinline__ float reduce16(float (&v)[16], int lane) {
  float a[8], b[4], c[2];
  const bool h5 = lane & 32, h4 = lane & 16, h3 = lane & 8, h2 = lane & 4;
#pragma unroll
  for (int j = 0; j < 8; ++j) a[j] = (h5 ? v[j + 8] : v[j]) + __shfl_xor(h5 ? v[j] : v[j + 8], 32, 64);
#pragma unroll
  for (int j = 0; j < 4; ++j) b[j] = (h4 ? a[j + 4] : a[j]) + __shfl_xor(h4 ? a[j] : a[j + 4], 16, 64);
#pragma unroll
  for (int j = 0; j < 2; ++j) c[j] = (h3 ? b[j + 2] : b[j]) + __shfl_xor(h3 ? b[j] : b[j + 2], 8, 64);
  float s = (h2 ? c[1] : c[0]) + __shfl_xor(h2 ? c[0] : c[1], 4, 64);
  s += __shfl_xor(s, 2, 64);
  s += __shfl_xor(s, 1, 64);
  return s;
}
__host__ __device__ constexpr int slot_lane(int j) {   // first lane holding the total of value j
  return ((j >> 3) & 1) << 5 | ((j >> 2) & 1) << 4 | ((j >> 1) & 1) << 3 | (j & 1) << 2;
}
__device__ __forceinline__ float lane_bcast(float x, int src_lane) {   // src_lane is a compile-time constant
  return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(x), src_lane));
}

// ------------------------------------------------------------------------------- rollout head
// one wave per row: lane owns CPL = HL/64 columns of the last hidden activation; the A+1 dot products
// of a row are reduced together (reduce16), after which the lanes of slot k own action dimension k.
template <int CPL>
__global__ __launch_bounds__(256) void head_act_kernel(const float* __restrict__ Hc, const float* __restrict__ Ha,
                                                       const float* __restrict__ W4c, const float* __restrict__ b4c,
                                                       const float* __restrict__ W4a, const float* __restrict__ b4a,
                                                       const float* __restrict__ logstd,
                                                       const float* __restrict__ eps,
                                                       const float* __restrict__ given, int64_t M, int A,
                                                       float* __restrict__ action, float* __restrict__ logprob,
                                                       void* __restrict__ value_out, int value_f16,
                                                       const catppo_iter_state* __restrict__ rng_state, int rng_step,
                                                       float* __restrict__ eps_out) {
  constexpr int HL = CPL * 64;
  constexpr int VS = 15;
  extern __shared__ __attribute__((aligned(16))) float lds[];   // [16*HL] actor head weights, rows >= A zero
  const int lane = threadIdx.x & 63;
  const int64_t wave_id = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  const int64_t n_waves = (int64_t)gridDim.x * 4;
  for (int o = threadIdx.x; o < 16 * HL; o += 256) lds[o] = (W4a != nullptr && o < A * HL) ? W4a[o] : 0.0f;
  __syncthreads();
  float wc[CPL];
#pragma unroll
  for (int c = 0; c < CPL; ++c) wc[c] = W4c[lane * CPL + c];
  const float bc = b4c[0];
  const int slot = ((lane >> 5) & 1) * 8 + ((lane >> 4) & 1) * 4 + ((lane >> 3) & 1) * 2 + ((lane >> 2) & 1);
  const bool mine = slot < A;
  const float sd = mine ? expf(logstd[slot]) : 1.0f;
  const float var = sd * sd, lsd = logf(sd);
  const float ba = mine ? b4a[slot] : 0.0f;
  // on-device action noise: Philox4x32-10 keyed by the run's seed, counter {env, quad, step, iteration}
  uint32_t rk0 = 0, rk1 = 0, rit = 0;
  if (rng_state != nullptr) {
    const uint64_t sd64 = rng_state->seed;
    rk0 = (uint32_t)sd64, rk1 = (uint32_t)(sd64 >> 32), rit = (uint32_t)rng_state->iteration;
  }
  for (int64_t i = wave_id; i < M; i += n_waves) {
    float part[16];
    float dc = 0.0f;
#pragma unroll
    for (int c = 0; c < CPL; ++c) dc = fmaf(Hc[i * HL + lane * CPL + c], wc[c], dc);
    part[VS] = dc;
    if (Ha != nullptr) {
      float ha[CPL];
#pragma unroll
      for (int c = 0; c < CPL; ++c) ha[c] = Ha[i * HL + lane * CPL + c];
#pragma unroll
      for (int k = 0; k < VS; ++k) {
        float d = 0.0f, wk[CPL];
        load_vec<CPL>(lds + k * HL + lane * CPL, wk);
#pragma unroll
        for (int c = 0; c < CPL; ++c) d = fmaf(ha[c], wk[c], d);
        part[k] = d;
      }
    } else {
#pragma unroll
      for (int k = 0; k < VS; ++k) part[k] = 0.0f;
    }
    const float tot = reduce16(part, lane);
    const float v = lane_bcast(tot, slot_lane(VS)) + bc;
    if (Ha != nullptr) {
      const float mu = tot + ba;
      float a = mu;
      if (mine && given != nullptr) {
        a = given[i * A + slot];
      } else if (mine && rng_state != nullptr) {
        const rng::u32x4 blk = rng::philox4x32_10(rng::u32x4{(uint32_t)i, (uint32_t)(slot >> 2), (uint32_t)rng_step, rit},
                                                  rk0, rk1);
        const float e = rng::box_muller_pick(blk, slot & 3);
        a = mu + sd * e;
        if (eps_out != nullptr && (lane & 3) == 0) eps_out[i * A + slot] = e;
      } else if (mine && eps != nullptr) {
        a = mu + sd * eps[i * A + slot];   // Normal.sample(): loc + scale*N(0,1)
      }
      const float diff = a - mu;
      const float term = mine ? (-(diff * diff) / (2.0f * var) - lsd - kHalfLog2Pi) : 0.0f;
      float lp = 0.0f;
#pragma unroll
      for (int k = 0; k < VS; ++k) lp += lane_bcast(term, slot_lane(k));
      if (mine && (lane & 3) == 0) action[i * A + slot] = a;
      if (lane == 0) logprob[i] = lp;
    }
    if (lane == 0) {
      if (value_f16) reinterpret_cast<_Float16*>(value_out)[i] = (_Float16)v;   // fp16 rollout plane (RNE)
      else reinterpret_cast<float*>(value_out)[i] = v;
    }
  }
}

// ------------------------------------------------------------------------------- fused small-batch forward
// The whole policy / value forward of ONE network for 32 rows in one workgroup: every hidden layer and the head.
// At rollout size (4096 envs) the layer-wise path is 3 GEMM launches + head_act = ~48 us for 2.4 GFLOP (0.31 of the
// fp32-MFMA peak): every launch is one round of small workgroups whose prologue / epilogue / boundary nothing overlaps.
// Here M/32 x 2 workgroups (one per CU at 4096 rows) keep their activation tile in LDS from layer to layer and stream
// the weights (L2 resident: every CU reads the same slabs) through a three-slot LDS ring:
//   iteration s:  MFMAs of slab s on fragments already in registers | ds_read the fragments of slab s+1 (slot written
//                 one barrier ago) | ds_write slab s+2 from the staging registers | global_load slab s+3 | barrier
// so the matrix pipe only ever waits for the barrier itself.  One wave per SIMD can keep the fp32 MFMA pipe full
// (64 cycles per v_mfma_f32_32x32x2_f32, ~15 issue slots behind each), which is why 1 workgroup per CU is enough here.
// Contraction order = gemm_body's (slab, 8-k block, lane half, step): results are bit-identical to the layer-wise path.
#ifdef FUSED_TL   // tools/fused_fwd_timeline.py: thread 0 of every workgroup stamps the wall clock at the phase boundaries
__device__ unsigned long long* g_fftl;    // [2 nets][1024 workgroups][16 stamps]
#define FF_TL(i) do { if (threadIdx.x == 0 && g_fftl) { g_fftl[(blockIdx.y * 1024 + blockIdx.x) * 16 + (i)] = wall_clock64(); \
      if ((i) == 2 || (i) == 3) g_fftl[(blockIdx.y * 1024 + blockIdx.x) * 16 + 8 + (i)] = clock64(); } } while (0)
// shader-clock (s_memtime) stamps of thread 0 next to the wall-clock ones: [2 * 1024 * 16 + workgroup * 4 + i]
#define FF_CK(i) do { if (threadIdx.x == 0 && g_fftl) { g_fftl[2 * 1024 * 16 + (blockIdx.y * 1024 + blockIdx.x) * 4 + (i)] = clock64(); \
      g_fftl[2 * 1024 * 16 + (blockIdx.y * 1024 + blockIdx.x) * 4 + 2 + (i)] = wall_clock64(); } } while (0)
#else
#define FF_TL(i) do { } while (0)
#define FF_CK(i) do { } while (0)
#endif
constexpr int kFR = 32;            // rows per workgroup
constexpr int kFT = 512;           // threads per workgroup: eight waves (one 32-column strip of a 256-column chunk each)
constexpr int kFWS = 20;           // floats per weight-slab row in LDS (16 k + 4 pad: conflict-free ds_read_b128)
constexpr int kFRing = 3 * 256 * kFWS;

struct FusedFwdArgs {
  const float* x;                  // [M, Dp]
  const float* params;
  int64_t M;
  int Dp, n_hidden;
  int hidden[CATPPO_MAX_HIDDEN];
  int64_t off_w[2][CATPPO_MAX_HIDDEN + 1], off_b[2][CATPPO_MAX_HIDDEN + 1];
  int net0;                        // network of blockIdx.y == 0 (0 critic, 1 actor)
  int ld0, ld1;                    // row strides (floats) of the two LDS activation tiles
  float* Hout[2][CATPPO_MAX_HIDDEN];   // [net][layer] global copy of the activations (training) or null
  // head (rollout), as head_act_kernel
  const float *logstd, *eps, *given;
  int A;
  float *action, *logprob;
  void* value_out;
  int value_f16;
  const catppo_iter_state* rng_state;
  int rng_step;
  float* eps_out;
  int do_head;
  int nets_per_wg;                 // rows_fwd_kernel: 2 = one workgroup walks both networks (grid.y == 1), 1 = grid.y == nets
  int store_policy;                // rows_fwd_kernel activation stores: 0 all write-through (sc1), 1 write-through only for the
                                   // last layer of the last network a workgroup walks (the rest may sit in L2: they have the
                                   // rest of the launch to drain), 2 none
};

// Heads of the fused forward on the 32-row tile in LDS.  head_act_kernel gives every row a whole wave (the launch has
// thousands of waves to hide the Philox / Box-Muller / log-prob latency behind); a fused workgroup has four waves and
// 32 rows, so the wave-per-row form costs 8 serial rows of ~2500 dependent cycles each (8 us of a 45 us kernel).  Here
// the work is spread over items: actor = (row, action slot) with the 16 slots of a row in 16 adjacent lanes (two items
// per thread), critic = (row, eighth of the contraction) with 8 lanes per row.
template <int HL>
__device__ __forceinline__ void fused_head(const FusedFwdArgs& a, const float* __restrict__ hs, int ld, int net,
                                           int64_t r0, float* __restrict__ wlds) {
  constexpr int WL = HL + 4;                       // padded weight rows: 16 slots read the same column without conflicts
  const int tid = threadIdx.x;
  const int nl = a.n_hidden, A = a.A;
  const float* W4 = a.params + a.off_w[net][nl];
  const float* b4 = a.params + a.off_b[net][nl];
  const int n_out = net == 1 ? A : 1;
  for (int o = tid; o < 16 * HL; o += kFT) {
    const int k = o / HL, c = o - k * HL;
    wlds[k * WL + c] = k < n_out ? W4[o] : 0.0f;
  }
  __syncthreads();
  if (net == 0) {
    const int r = tid >> 4, part = tid & 15;       // 16 lanes per row, HL / 16 columns each
    const int64_t i = r0 + r;
    const float* hp = hs + r * ld + part * (HL / 16);
    const float* wp = wlds + part * (HL / 16);
    float d0 = 0.0f, d1 = 0.0f;
#pragma unroll
    for (int c = 0; c < HL / 16; c += 8) {
      const float4 h0 = *reinterpret_cast<const float4*>(hp + c), h1 = *reinterpret_cast<const float4*>(hp + c + 4);
      const float4 w0 = *reinterpret_cast<const float4*>(wp + c), w1 = *reinterpret_cast<const float4*>(wp + c + 4);
      d0 = fmaf(h0.x, w0.x, d0), d0 = fmaf(h0.y, w0.y, d0), d0 = fmaf(h0.z, w0.z, d0), d0 = fmaf(h0.w, w0.w, d0);
      d1 = fmaf(h1.x, w1.x, d1), d1 = fmaf(h1.y, w1.y, d1), d1 = fmaf(h1.z, w1.z, d1), d1 = fmaf(h1.w, w1.w, d1);
    }
    float d = d0 + d1;
    d += __shfl_xor(d, 1, 64);
    d += __shfl_xor(d, 2, 64);
    d += __shfl_xor(d, 4, 64);
    d += __shfl_xor(d, 8, 64);
    const float v = d + b4[0];
    if (part == 0 && i < a.M) {
      if (a.value_f16) reinterpret_cast<_Float16*>(a.value_out)[i] = (_Float16)v;
      else reinterpret_cast<float*>(a.value_out)[i] = v;
    }
    return;
  }
  const int k = tid & 15;                          // action slot of this thread (both items)
  const bool kin = k < A;
  const float sd = kin ? expf(a.logstd[k]) : 1.0f;
  const float var = sd * sd, lsd = logf(sd);
  const float ba = kin ? b4[k] : 0.0f;
  uint32_t rk0 = 0, rk1 = 0, rit = 0;
  if (a.rng_state != nullptr) {
    const uint64_t sd64 = a.rng_state->seed;
    rk0 = (uint32_t)sd64, rk1 = (uint32_t)(sd64 >> 32), rit = (uint32_t)a.rng_state->iteration;
  }
  {
    const int r = tid >> 4;                        // one (row, slot) item per thread
    const int64_t i = r0 + r;
    const bool mine = kin && i < a.M;
    const float* hp = hs + r * ld;
    const float* wp = wlds + k * WL;
    float d0 = 0.0f, d1 = 0.0f, d2 = 0.0f, d3 = 0.0f;
#pragma unroll 4
    for (int c = 0; c < HL; c += 16) {
      const float4 h0 = *reinterpret_cast<const float4*>(hp + c), h1 = *reinterpret_cast<const float4*>(hp + c + 4);
      const float4 h2 = *reinterpret_cast<const float4*>(hp + c + 8), h3 = *reinterpret_cast<const float4*>(hp + c + 12);
      const float4 w0 = *reinterpret_cast<const float4*>(wp + c), w1 = *reinterpret_cast<const float4*>(wp + c + 4);
      const float4 w2 = *reinterpret_cast<const float4*>(wp + c + 8), w3 = *reinterpret_cast<const float4*>(wp + c + 12);
      d0 = fmaf(h0.x, w0.x, d0), d0 = fmaf(h0.y, w0.y, d0), d0 = fmaf(h0.z, w0.z, d0), d0 = fmaf(h0.w, w0.w, d0);
      d1 = fmaf(h1.x, w1.x, d1), d1 = fmaf(h1.y, w1.y, d1), d1 = fmaf(h1.z, w1.z, d1), d1 = fmaf(h1.w, w1.w, d1);
      d2 = fmaf(h2.x, w2.x, d2), d2 = fmaf(h2.y, w2.y, d2), d2 = fmaf(h2.z, w2.z, d2), d2 = fmaf(h2.w, w2.w, d2);
      d3 = fmaf(h3.x, w3.x, d3), d3 = fmaf(h3.y, w3.y, d3), d3 = fmaf(h3.z, w3.z, d3), d3 = fmaf(h3.w, w3.w, d3);
    }
    const float mu = ((d0 + d1) + (d2 + d3)) + ba;
    float act = mu;
    if (mine && a.given != nullptr) {
      act = a.given[i * A + k];
    } else if (mine && a.rng_state != nullptr) {
      const rng::u32x4 blk = rng::philox4x32_10(rng::u32x4{(uint32_t)i, (uint32_t)(k >> 2), (uint32_t)a.rng_step, rit},
                                                rk0, rk1);
      const float e = rng::box_muller_pick(blk, k & 3);
      act = mu + sd * e;
      if (a.eps_out != nullptr) a.eps_out[i * A + k] = e;
    } else if (mine && a.eps != nullptr) {
      act = mu + sd * a.eps[i * A + k];             // Normal.sample(): loc + scale * N(0,1)
    }
    const float diff = act - mu;
    float lp = kin ? (-(diff * diff) / (2.0f * var) - lsd - kHalfLog2Pi) : 0.0f;
    lp += __shfl_xor(lp, 1, 64);                   // the 16 slots of a row sit in 16 adjacent lanes
    lp += __shfl_xor(lp, 2, 64);
    lp += __shfl_xor(lp, 4, 64);
    lp += __shfl_xor(lp, 8, 64);
    if (mine) a.action[i * A + k] = act;
    if (k == 0 && i < a.M) a.logprob[i] = lp;
  }
}

// one chunk of NC (256 or 128) output columns of one layer for the workgroup's 32 rows: out[:, c0 + ...] = elu(in . W^T
// + b).  Eight waves: wave w owns columns [32 w, 32 w + 32) of the chunk (a 128-column chunk occupies waves 0-3 only).
// W = the chunk's first weight row.  What the measurements of round 3 left standing (tools/fused_fwd_timeline.py with the
// -DFUSED_EXP_* switches, tools/mfma_rate_probe.hip):
//  * no workgroup barrier inside the contraction: every wave streams the weight rows of ITS OWN 32 columns (2 KB per
//    16-k slab: lane -> row lane / 4 (+16), k quad lane % 4, four lanes per 64-B row segment) through a wave-private
//    three-slot LDS ring; the only LDS hand-off is from a wave to itself (LDS operations of one wave execute in
//    order), the activation tile is read-only during a layer, and the eight waves drift instead of meeting per slab;
//  * the eight MFMAs of a slab are issued BACK TO BACK and everything else (fragments of the next slab, ring <- the
//    staged slab, the next request) in one block behind them: the probe shows one wave with ONE accumulator sustaining
//    142 TFLOP/s of v_mfma_f32_32x32x2_f32 when nothing sits between the MFMAs, so neither a second accumulator nor a
//    second wave per SIMD is needed for the matrix pipe (both were tried: no change);
//  * the loop body is guard free (~22 instructions per slab): the first version guarded every stage of every slab and
//    copied prefetched fragments - 185 instructions per slab and wave, ISSUE bound at 2070 cycles per slab; the
//    pipeline now simply runs past the end (the last iterations stage up to three slabs nobody multiplies: weight rows
//    are followed by more parameters in the flat buffer, the fragments read past K stay inside the LDS allocation);
//  * deeper weight prefetch (three staging register sets, inline-asm loads with exact vmcnt) changed nothing - the
//    requests are L2 hits that arrive within a slab - and was removed again.
// One accumulator, contraction order = gemm_body's (slab, 8-k block, lane half, step): bit-identical to the layer-wise path.
// the first two weight slabs of the NEXT chunk, requested while the current chunk multiplies (a chunk's own prologue is
// two serial memory round trips, ~1.5 us of a ~12 us layer, with nothing to overlap them inside the chunk)
struct FusedPre {
  float4 p0, p1, p2, p3;      // slab 0 rows (lane/4, lane/4 + 16), slab 1 likewise
  bool valid;                 // wave-uniform
};

template <int NC>
__device__ __forceinline__ void fused_chunk(const float* __restrict__ in, const int ldin, float* __restrict__ out,
                                            const int ldout, float* __restrict__ ring, const float* __restrict__ W,
                                            const float* __restrict__ bias_c, const int c0, const int K,
                                            const float* __restrict__ Wnext, const int Knext, const int NCnext,
                                            FusedPre& pre) {
  using gemm::f32x16;
  const int tid = threadIdx.x, lane = tid & 63, l31 = lane & 31, h = lane >> 5;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const bool want_next = Wnext != nullptr && wave * 32 < NCnext;     // wave-uniform
  if (NC == 128 && wave >= 4) {                              // wave-uniform: nothing to multiply in a 128-column chunk
    pre.valid = false;
    if (want_next) {
      const float* np_ = Wnext + (int64_t)(wave * 32 + (lane >> 2)) * Knext + 4 * (lane & 3);
      pre.p0 = *reinterpret_cast<const float4*>(np_);
      pre.p1 = *reinterpret_cast<const float4*>(np_ + (int64_t)16 * Knext);
      pre.p2 = *reinterpret_cast<const float4*>(np_ + 16);
      pre.p3 = *reinterpret_cast<const float4*>(np_ + (int64_t)16 * Knext + 16);
      pre.valid = true;
    }
    return;
  }
  const int n_slabs = K / 16;
  const float bias = bias_c[wave * 32 + l31];                // requested before the contraction, used after it
  float* const wring = ring + wave * (3 * 32 * kFWS);
  const float* gp = W + (int64_t)(wave * 32 + (lane >> 2)) * K + 4 * (lane & 3);     // this lane's element of slab 0
  const int64_t gq = (int64_t)16 * K;                        // 64 lanes = 16 rows further per load
  float* const s0 = wring + (lane >> 2) * kFWS + 4 * (lane & 3);
  float4 w0, w1;                                             // staging registers: one slab in flight
#define FUSED_GL()                                                                        \
  do {                                                                                    \
    w0 = *reinterpret_cast<const float4*>(gp);                                            \
    w1 = *reinterpret_cast<const float4*>(gp + gq);                                       \
    gp += 16;                                                                             \
  } while (0)
#define FUSED_ST(slot_)                                                                   \
  do {                                                                                    \
    float* dp = s0 + (slot_) * (32 * kFWS);                                               \
    *reinterpret_cast<float4*>(dp) = w0;                                                  \
    *reinterpret_cast<float4*>(dp + 16 * kFWS) = w1;                                      \
  } while (0)
  float4 a0, a1, b0, b1;          // fragments of the slab being multiplied: A / B of its two 8-k blocks
  const float* ap = in + l31 * ldin + 4 * h;                                   // A fragments: k advances 16 per slab
  const float* const bp = wring + l31 * kFWS + 4 * h;                           // B fragments inside a ring slot
  f32x16 acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.0f;
#ifdef FUSED_EXP_NOMFMA      // timing experiments (tools/fused_fwd_timeline.py): wrong results, never in the product build
#define FUSED_MM(av, bv) acc[0] += (av) * (bv)
#else
#define FUSED_MM(av, bv) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, acc, 0, 0, 0)
#endif
#ifdef FUSED_EXP_NOLOAD
#define FUSED_DO_LOAD 0
#else
#define FUSED_DO_LOAD 1
#endif
  // one slab whose ring slot is SL (static, the loop is unrolled by three): its MFMAs, then the fragments of the next
  // slab (slot SL+1) into the same registers, ring slot SL+2 <- the staged slab, request of the slab after that
#define FUSED_SLAB(SL)                                                                    \
  do {                                                                                    \
    FUSED_MM(a0.x, b0.x);                                                                 \
    FUSED_MM(a0.y, b0.y);                                                                 \
    FUSED_MM(a0.z, b0.z);                                                                 \
    FUSED_MM(a0.w, b0.w);                                                                 \
    FUSED_MM(a1.x, b1.x);                                                                 \
    FUSED_MM(a1.y, b1.y);                                                                 \
    FUSED_MM(a1.z, b1.z);                                                                 \
    FUSED_MM(a1.w, b1.w);                                                                 \
    __builtin_amdgcn_sched_barrier(0);                                                    \
    ap += 16;                                                                             \
    a0 = *reinterpret_cast<const float4*>(ap);                                            \
    a1 = *reinterpret_cast<const float4*>(ap + 8);                                        \
    b0 = *reinterpret_cast<const float4*>(bp + (((SL) + 1) % 3) * (32 * kFWS));           \
    b1 = *reinterpret_cast<const float4*>(bp + (((SL) + 1) % 3) * (32 * kFWS) + 8);       \
    if (FUSED_DO_LOAD) {                                                                  \
      FUSED_ST(((SL) + 2) % 3);                                                           \
      FUSED_GL();                                                                         \
    }                                                                                     \
    __builtin_amdgcn_sched_barrier(0);                                                    \
  } while (0)
  // prologue: slabs 0 and 1 into the ring (already in registers when the previous chunk requested them), slab 2 requested
  if (pre.valid) {
    w0 = pre.p0, w1 = pre.p1;
    FUSED_ST(0);
    w0 = pre.p2, w1 = pre.p3;
    FUSED_ST(1);
    gp += 32;
    FUSED_GL();
  } else {
    FUSED_GL();
    FUSED_ST(0);
    FUSED_GL();
    FUSED_ST(1);
    FUSED_GL();
  }
  pre.valid = false;
  if (want_next) {            // lands while this chunk multiplies; stored by the next chunk's prologue
    const float* np_ = Wnext + (int64_t)(wave * 32 + (lane >> 2)) * Knext + 4 * (lane & 3);
    pre.p0 = *reinterpret_cast<const float4*>(np_);
    pre.p1 = *reinterpret_cast<const float4*>(np_ + (int64_t)16 * Knext);
    pre.p2 = *reinterpret_cast<const float4*>(np_ + 16);
    pre.p3 = *reinterpret_cast<const float4*>(np_ + (int64_t)16 * Knext + 16);
    pre.valid = true;
  }
  a0 = *reinterpret_cast<const float4*>(ap);
  a1 = *reinterpret_cast<const float4*>(ap + 8);
  b0 = *reinterpret_cast<const float4*>(bp);
  b1 = *reinterpret_cast<const float4*>(bp + 8);
  int s = 0;
  for (; s + 3 <= n_slabs; s += 3) {
    FUSED_SLAB(0);              // slab s   : ring slot 2 <- slab s+2, request s+3
    FUSED_SLAB(1);              // slab s+1 : ring slot 0 <- slab s+3, request s+4
    FUSED_SLAB(2);              // slab s+2 : ring slot 1 <- slab s+4, request s+5
  }
  if (s < n_slabs) FUSED_SLAB(0);
  if (s + 1 < n_slabs) FUSED_SLAB(1);
#undef FUSED_SLAB
#undef FUSED_DO_LOAD
#undef FUSED_GL
#undef FUSED_ST
#undef FUSED_MM
  // bias + ELU -> output tile.  acc[r] of a lane: row (r&3) + 8 (r>>2) + 4 h, column l31
  const int cc = wave * 32 + l31;
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int row = (r & 3) + 8 * (r >> 2) + 4 * h;
    out[row * ldout + c0 + cc] = gemm::elu_f(acc[r] + bias);
  }
}

__global__ __launch_bounds__(kFT) void fused_fwd_kernel(const FusedFwdArgs a) {
  using gemm::f32x16;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* act0 = smem;
  float* act1 = act0 + kFR * a.ld0;
  float* ring = act1 + kFR * a.ld1;                       // [3][256][kFWS]
  const int net = a.net0 + blockIdx.y;
  const int64_t r0 = (int64_t)blockIdx.x * kFR;
  const int tid = threadIdx.x, lane = tid & 63, l31 = lane & 31, h = lane >> 5;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);

  FF_TL(0);
  // touch this wave's first weight rows of layer 0 now: the observation tile below costs one memory round trip anyway,
  // and the contraction's own first requests then find the lines close by instead of paying a second, serial one
  float4 warm0, warm1;
  {
    const float* w0p = a.params + a.off_w[net][0] + (int64_t)(wave * 32 + (lane >> 2)) * a.Dp + 4 * (lane & 3);
    warm0 = *reinterpret_cast<const float4*>(w0p);
    warm1 = *reinterpret_cast<const float4*>(w0p + (int64_t)16 * a.Dp);
  }
  {   // observation tile -> act0 (rows past M are zero: their results are never stored)
    const int q4 = a.Dp / 4;
    for (int f = tid; f < kFR * q4; f += kFT) {
      const int r = f / q4, q = f - r * q4;
      float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
      if (r0 + r < a.M) v = *reinterpret_cast<const float4*>(a.x + (r0 + r) * a.Dp + 4 * q);
      *reinterpret_cast<float4*>(act0 + r * a.ld0 + 4 * q) = v;
    }
  }
  __syncthreads();
  asm volatile("" ::"v"(warm0.x), "v"(warm1.x));       // keeps the two warm-up loads
  FF_TL(1);

  float* in = act0;
  float* out = act1;
  int ldin = a.ld0, ldout = a.ld1;
  int K = a.Dp;
  FusedPre pre;
  pre.valid = false;
  for (int l = 0; l < a.n_hidden; ++l) {
    const int N = a.hidden[l];
    const float* Wl = a.params + a.off_w[net][l];
    const float* bl = a.params + a.off_b[net][l];
    for (int c0 = 0; c0 < N; c0 += 256) {
      // the chunk after this one (same layer or the first of the next layer): its first weight rows, contraction width
      const float* Wn = nullptr;
      int Kn = 0, NCn = 0;
      if (c0 + 256 < N) {
        Wn = Wl + (int64_t)(c0 + 256) * K, Kn = K, NCn = (N - c0 - 256) >= 256 ? 256 : 128;
      } else if (l + 1 < a.n_hidden) {
        Wn = a.params + a.off_w[net][l + 1], Kn = N, NCn = a.hidden[l + 1] >= 256 ? 256 : 128;
      }
      if (N - c0 >= 256) fused_chunk<256>(in, ldin, out, ldout, ring, Wl + (int64_t)c0 * K, bl + c0, c0, K, Wn, Kn, NCn, pre);
      else fused_chunk<128>(in, ldin, out, ldout, ring, Wl + (int64_t)c0 * K, bl + c0, c0, K, Wn, Kn, NCn, pre);   // 128 columns left
    }
    __syncthreads();
    float* hg = a.Hout[blockIdx.y][l];
    if (hg != nullptr) {     // training: the activations also go to memory (backward reads them)
      const int q4 = N / 4;
      for (int f = tid; f < kFR * q4; f += kFT) {
        const int r = f / q4, q = f - r * q4;
        if (r0 + r < a.M) {
          const float4 v = *reinterpret_cast<const float4*>(out + r * ldout + 4 * q);
          const float o[4] = {v.x, v.y, v.z, v.w};
          store_vec_wt<4>(hg + (r0 + r) * N + 4 * q, o);
        }
      }
    }
    float* t = in;
    in = out, out = t;
    const int tl = ldin;
    ldin = ldout, ldout = tl;
    K = N;
    FF_TL(2 + l);
  }
  if (!a.do_head) return;
  // `in` now holds the last hidden activations [32][HL]
  switch (K) {
    case 128: fused_head<128>(a, in, ldin, net, r0, ring); break;
    case 256: fused_head<256>(a, in, ldin, net, r0, ring); break;
    case 512: fused_head<512>(a, in, ldin, net, r0, ring); break;
    default: break;
  }
  FF_TL(8);
}

// ------------------------------------------------------------------------------- row-resident forward (round 4)
// fwd_rows.h: R rows of activations stay in ONE LDS tile from layer to layer (in place), weights stream through
// wave-private rings in full 128-byte lines.  R = 64: the hidden layers below the last one of a training minibatch
// (activations also stored for the backward; the last layer + heads stay with fwd_head_kernel).  R = 32: the whole
// rollout forward incl. heads (same role as fused_fwd_kernel).  Every layer handled here is 256 wide.
#include "fwd_rows.h"

// TRAIN: activations of every layer go to memory (no head).  !TRAIN: rollout, heads at the end.  NETS: networks a
// workgroup walks.  NL: layers.  All compile-time, and both loops below fully unrolled: the compiler's s_waitcnt
// bookkeeping merges the states of a loop's entry and back edge conservatively, and a wait shared by "no stores in
// flight" (first layer) and "eight activation stores younger than the load I need" (later layers) would come out as
// vmcnt(0) - i.e. every layer would wait for the store tail of the one before.
template <int R, bool TRAIN, int NETS, int NL>
__global__ __launch_bounds__(rowsfwd::kThreads) void rows_fwd_kernel(const FusedFwdArgs a) {
  using gemm::f32x16;
  using u32x4 = __attribute__((ext_vector_type(4))) unsigned int;
  constexpr int T = R / 32;
  constexpr int XQ = 8;                                       // float4 of the observation tile per thread (Dp <= 256)
  constexpr int HQ = R * (rowsfwd::kWidth / 4) / rowsfwd::kThreads;     // float4 of an activation tile per thread
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* tile = smem;                                         // [R][ld]
  const int ld = a.ld0;
  const int tid = threadIdx.x, lane = tid & 63, l31 = lane & 31, h = lane >> 5;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  float* wring = smem + R * ld + wave * rowsfwd::kRingWave;    // this wave's two weight slots
  const int64_t r0 = (int64_t)blockIdx.x * R;
  constexpr int nets_here = NETS;
  // Observation rows and activation rows go through buffer descriptors: a row past M is out of range - the load returns
  // zeros, the store is dropped - so neither needs a branch, and the compiler can COUNT them (it cannot count loads /
  // stores under a divergent branch or inside inline asm; every later wait then becomes vmcnt(0) and stalls on the
  // activation stores of the layer before)
  const int q4 = a.Dp / 4;
  const __amdgpu_buffer_rsrc_t xrs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.x), 0,
                                                                       (int)(a.M * a.Dp * 4), 0x00020000);
  uint32_t xoff[XQ], xlds[XQ];
#pragma unroll
  for (int j = 0; j < XQ; ++j) {
    const int f = tid + j * rowsfwd::kThreads;
    const int r = f / q4, q = f - r * q4;
    const bool on = f < R * q4;
    xoff[j] = on ? (uint32_t)(((r0 + r) * a.Dp + 4 * q) * 4) : 0xffffffffu;
    xlds[j] = on ? (uint32_t)(r * ld + 4 * q) : 0xffffffffu;
  }
  u32x4 xr[XQ];
  auto x_request = [&]() {
#pragma unroll
    for (int j = 0; j < XQ; ++j) xr[j] = __builtin_amdgcn_raw_buffer_load_b128(xrs, xoff[j], 0, 0);
  };
  auto x_to_tile = [&]() {
#pragma unroll
    for (int j = 0; j < XQ; ++j)
      if (xlds[j] != 0xffffffffu) *reinterpret_cast<u32x4*>(tile + xlds[j]) = xr[j];
  };

  rowsfwd::Layer<R> ly;
  FF_TL(0);
  // prologue of the first network: weights of layer 0 and the observation tile requested together
  int net = a.net0 + (nets_here == 2 ? 0 : (int)blockIdx.y);
  float bias = a.params[a.off_b[net][0] + wave * 32 + l31];
  ly.stage(a.params + a.off_w[net][0], a.Dp, wave, lane);
  x_request();
#pragma unroll
  for (int ni = 0; ni < nets_here; ++ni) {
    const int slot_net = nets_here == 2 ? ni : (int)blockIdx.y;      // index into a.Hout
    if (ni > 0) __syncthreads();                              // the previous network's last tile has been read out
    x_to_tile();
    __syncthreads();
    FF_TL(1 + 8 * ni);
    ly.begin(wring, lane);
    int K = a.Dp;
#pragma unroll
    for (int l = 0; l < NL; ++l) {
      f32x16 acc[T];
      if (ni == 0 && l == NL - 1) FF_CK(0);
      ly.loop(tile, ld, wring, K, acc, lane);
      if (ni == 0 && l == NL - 1) FF_CK(1);
      FF_TL(2 + 8 * ni + 2 * l);
      // what comes next - layer l+1 of this network, or layer 0 of the next one - is requested NOW: first weight slabs,
      // bias, (next network) observation tile; all of it lands behind the two barriers and the tile write below
      const bool more_layers = l + 1 < NL;
      const bool more_nets = !more_layers && ni + 1 < nets_here;
      float bias_next = 0.0f;
      if (more_layers) {
        bias_next = a.params[a.off_b[net][l + 1] + wave * 32 + l31];
        ly.stage(a.params + a.off_w[net][l + 1], rowsfwd::kWidth, wave, lane);
      } else if (more_nets) {
        bias_next = a.params[a.off_b[net + 1][0] + wave * 32 + l31];
        ly.stage(a.params + a.off_w[net + 1][0], a.Dp, wave, lane);
        x_request();
      }
      __syncthreads();                                        // every wave is done reading the tile: overwrite it
      if (NL == 2 && ni == 0 && l == 1) FF_TL(6);
#pragma unroll
      for (int t = 0; t < T; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int row = 32 * t + (r & 3) + 8 * (r >> 2) + 4 * h;
          tile[row * ld + wave * 32 + l31] = gemm::elu_f(acc[t][r] + bias);
        }
      __syncthreads();
      if (NL == 2 && ni == 0 && l == 1) FF_TL(7);
      // ring <- the staged slabs BEFORE the activation stores: every load issued so far is then older than the stores
      if (more_layers) ly.begin(wring, lane);
      if (NL == 2 && ni == 0 && l == 0) FF_TL(14);
      if (TRAIN) {               // training: the activations also go to memory, under the next layer's contraction
        float* hg = a.Hout[slot_net][l];
        const __amdgpu_buffer_rsrc_t hrs =
            __builtin_amdgcn_make_buffer_rsrc(hg, 0, (int)(a.M * rowsfwd::kWidth * 4), 0x00020000);
        u32x4 hv[HQ];
#pragma unroll
        for (int j = 0; j < HQ; ++j) {
          const int f = tid + j * rowsfwd::kThreads;
          hv[j] = *reinterpret_cast<const u32x4*>(tile + (f >> 6) * ld + 4 * (f & 63));
        }
        if (NL == 2 && ni == 0 && l == 1) FF_TL(15);
#pragma unroll
        for (int j = 0; j < HQ; ++j) {
          const int f = tid + j * rowsfwd::kThreads;
          const uint32_t ho = (uint32_t)(((r0 + (f >> 6)) * rowsfwd::kWidth + 4 * (f & 63)) * 4);
          // write-through (sc1): rows the NEXT launch reads should not sit dirty in L2 until the kernel boundary flushes
          // them; rows written long before the end of this launch drain by themselves (store_policy)
          const bool wt = a.store_policy == 0 || (a.store_policy == 1 && l == NL - 1 && ni == nets_here - 1);
          if (wt) __builtin_amdgcn_raw_buffer_store_b128(hv[j], hrs, ho, 0, 16);
          else __builtin_amdgcn_raw_buffer_store_b128(hv[j], hrs, ho, 0, 0);
        }
      }
      bias = bias_next;
      K = rowsfwd::kWidth;
      FF_TL(3 + 8 * ni + 2 * l);
    }
    if (!TRAIN) {                // rollout: heads on the tile (the rings are free: head weights go there)
      fused_head<rowsfwd::kWidth>(a, tile, ld, net, r0, smem + R * ld);
    }
    ++net;
  }
}

#include "fwd_rows_wide.h"

#ifdef FUSED_TL
extern "C" int catppo_debug_fused_tl(void* buf) {     // timeline builds only: not part of include/catppo.h
  unsigned long long* pbuf = static_cast<unsigned long long*>(buf);
  return hipMemcpyToSymbol(HIP_SYMBOL(g_fftl), &pbuf, sizeof(pbuf)) == hipSuccess ? 0 : -1;
}
#endif

// ------------------------------------------------------------------------------- minibatch gather
// grid = (row chunks of one minibatch, minibatches).  Minibatch m = samples inds[m*M .. m*M + M_m) lands in the
// contiguous slices xmb[m*M ..], act[m*M ..], scal[4*m*M + {0,1,2,3}*M_m ..], adv_part[m][chunk][2].
__global__ __launch_bounds__(256) void ppo_gather_kernel(const float* __restrict__ b_obs, const float* __restrict__ b_act,
                                                         const float* __restrict__ b_logp,
                                                         const float* __restrict__ b_adv,
                                                         const float* __restrict__ b_ret,
                                                         const float* __restrict__ b_val,
                                                         const int64_t* __restrict__ inds, int64_t total, int64_t M,
                                                         int Dp, int A, float* __restrict__ xmb,
                                                         float* __restrict__ act, float* __restrict__ scal,
                                                         double* __restrict__ adv_part,
                                                         const catppo_iter_state* __restrict__ rng_state, int rng_epoch,
                                                         int adv_f16, int64_t* __restrict__ inds_out) {
  __shared__ int64_t s_idx[kGatherRows];
  const int64_t m0 = (int64_t)blockIdx.y * M;                     // first sample of this minibatch
  const int64_t Mm = (total - m0) < M ? (total - m0) : M;         // its size (the last one may be short)
  const int64_t r0 = (int64_t)blockIdx.x * kGatherRows;           // row chunk inside the minibatch
  if (r0 >= Mm) {
    if (threadIdx.x == 0) {
      adv_part[2 * ((int64_t)blockIdx.y * gridDim.x + blockIdx.x)] = 0.0;
      adv_part[2 * ((int64_t)blockIdx.y * gridDim.x + blockIdx.x) + 1] = 0.0;
    }
    return;
  }
  const int rows = (int)((Mm - r0) < kGatherRows ? (Mm - r0) : kGatherRows);
  xmb += m0 * Dp, act += m0 * A, scal += 4 * m0;
  adv_part += 2 * (int64_t)blockIdx.y * gridDim.x;
  if (threadIdx.x < rows) {
    int64_t src;
    if (rng_state != nullptr) {     // keyed bijection of [0,total): no index array, no sort (rng.h)
      rng::FeistelPerm perm;
      perm.init(rng_state->seed, rng_state->iteration, rng_epoch, total);
      src = perm(m0 + r0 + threadIdx.x);
      if (inds_out != nullptr) inds_out[m0 + r0 + threadIdx.x] = src;
    } else {
      src = inds[m0 + r0 + threadIdx.x];
    }
    s_idx[threadIdx.x] = src;
  }
  __syncthreads();
  const int q4 = Dp / 4;
  for (int f = threadIdx.x; f < rows * q4; f += 256) {
    const int r = f / q4, q = f - r * q4;
    reinterpret_cast<float4*>(xmb + (r0 + r) * Dp)[q] = reinterpret_cast<const float4*>(b_obs + s_idx[r] * Dp)[q];
  }
  for (int f = threadIdx.x; f < rows * A; f += 256) {
    const int r = f / A, k = f - r * A;
    act[(r0 + r) * A + k] = b_act[s_idx[r] * A + k];
  }
  if (threadIdx.x < 64) {   // wave 0: the four per-sample scalars + advantage moments
    double a1 = 0.0, a2 = 0.0;
    if (threadIdx.x < rows) {
      const int64_t src = s_idx[threadIdx.x], dst = r0 + threadIdx.x;
      const float adv = adv_f16 ? (float)reinterpret_cast<const _Float16*>(b_adv)[src] : b_adv[src];
      scal[0 * Mm + dst] = b_logp[src];
      scal[1 * Mm + dst] = adv;
      scal[2 * Mm + dst] = b_ret[src];
      scal[3 * Mm + dst] = b_val[src];
      a1 = (double)adv;
      a2 = a1 * a1;
    }
    a1 = wave_sum_d(a1);
    a2 = wave_sum_d(a2);
    if (threadIdx.x == 0) {
      adv_part[2 * blockIdx.x] = a1;
      adv_part[2 * blockIdx.x + 1] = a2;
    }
  }
}

// ------------------------------------------------------------------------------- heads + PPO loss + head backward
struct HeadArgs {
  const float *Hc, *Ha;        // [M, HL] last hidden activations (critic, actor)
  float *dZc, *dZa;            // [M, HL] out: gradient w.r.t. last hidden PRE-activations
  const float *W4c, *b4c, *W4a, *b4a, *logstd;
  const float *act, *oldlogp, *adv, *ret_n, *val_n;   // gathered minibatch
  const double* adv_part;      // [n_adv_part][2]
  int n_adv_part;
  const float* adv_stats;      // external {mean, std+1e-8} or null
  const float *vrms_mean, *vrms_var;
  float *part_w, *part_s;      // per-block partials
  int64_t M;
  int A;
  catppo_ppo_hparams hp;
};

// waves per block: a 32-row tile of a wide last layer (HL >= 256) fills the CU's LDS alone, so the block brings
// its own parallelism (16 waves x 2 rows at HL = 256); narrower layers co-reside 2-3 blocks per CU and do better
// with 8 x 4, and HL = 512 needs more than the 128 VGPRs a 1024-thread block may use
template <int CPL>
constexpr int head_waves() { return CPL == 4 ? 16 : 8; }   // CPL 8 needs > 128 VGPRs: 8 waves
constexpr int kHeadMaxBlocks = 512;  // = number of weight-gradient partials folded afterwards (2 blocks per CU)

// Heads + PPO loss + backward through the heads, one tile of 32 minibatch rows at a time:
//   phase 1 (wave per row)  last-hidden rows -> registers AND an LDS tile; A+1 dot products per row
//           (batched 16-value butterfly), log-prob, clipped losses, analytic d loss/d mu, d loss/d v;
//           dZ of the last hidden layer is stored; the per-row head gradients go to an LDS [32][16] tile
//   phase 2 (thread per weight column)  dW4 += G^T . H over the 32 rows of the tile from LDS; accumulators
//           stay in registers across the tiles of the block => ONE partial per block, no per-wave
//           reduction rounds
template <int CPL, int TRS = 0>   // TRS: row-tile override (16 for small minibatches: twice the workgroups)
__global__ __launch_bounds__(head_waves<CPL>() * 64, (CPL <= 4 ? 4 : 2)) void head_loss_kernel(const HeadArgs g) {
  constexpr int kHeadWaves = head_waves<CPL>();
  constexpr int TR = TRS ? TRS : (CPL == 8 ? 16 : kHeadRowsPerBlock);
  constexpr int kHeadRowsPerWave = TR / kHeadWaves;
  constexpr int HL = CPL * 64;
  constexpr int NT = kHeadWaves * 64;
  constexpr int NG = NT / HL >= 1 ? NT / HL : 1;       // phase-2 thread groups (HL <= 512)
  constexpr int KPG = 16 / NG;                         // head outputs per group (16 slots)
  constexpr int VS = 15;                               // slot of the critic output; actions use slots 0..A-1
  // ALL shared memory lives in the dynamic region: a static __shared__ object in front of it would
  // shift its base off 16-B alignment and every ds_read_b128 below would be replayed (64 cycles each)
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int A = g.A;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int NS = 2 * A + 1 + kHeadDiag;
  // every loop below runs over the 16 compile-time slots; unused slots carry zeros (weights, gradients)
  // so there is no data-dependent control flow inside the row loop
  float* s_wa = lds;                       // [16*HL]  actor head weights, rows >= A zero
  float* sHa = s_wa + 16 * HL;             // [TR*HL]  actor last-hidden tile
  float* sHc = sHa + TR * HL;              // [TR*HL]  critic last-hidden tile
  float* sG = sHc + TR * HL;               // [TR*16]  per-row head gradients: d mu_k (k<A), 0, ..., d v at VS
  float* ls = sG + TR * 16;                // [NS]     scalars: db4a[A], db4c, dlogstd[A], diag[8]
  float* s_adv = ls + 48;                  // [2]      advantage mean, std + 1e-8   (NS <= 2*15+1+8 = 39)

  for (int o = tid; o < 16 * HL; o += NT) s_wa[o] = o < A * HL ? g.W4a[o] : 0.0f;
  for (int o = tid; o < TR * 16; o += NT) sG[o] = 0.0f;
  // advantage statistics over the minibatch (ppo.py:314-318): mean, unbiased std
  if (wave == 0) {
    if (g.hp.norm_adv && g.adv_stats == nullptr) {
      double a1 = 0.0, a2 = 0.0;
      for (int b = lane; b < g.n_adv_part; b += 64) {
        a1 += g.adv_part[2 * b];
        a2 += g.adv_part[2 * b + 1];
      }
      a1 = wave_sum_d(a1);
      a2 = wave_sum_d(a2);
      if (lane == 0) {
        const double n = (double)g.M;
        const double mean = a1 / n;
        double var = (a2 - n * mean * mean) / (n - 1.0);   // NaN for n == 1, like torch.std()
        if (var < 0.0) var = 0.0;
        s_adv[0] = (float)mean;
        s_adv[1] = (float)sqrt(var) + 1e-8f;
      }
    } else if (lane == 0) {
      s_adv[0] = g.adv_stats ? g.adv_stats[0] : 0.0f;
      s_adv[1] = g.adv_stats ? g.adv_stats[1] : 1.0f;
    }
  }
  __syncthreads();
  const float adv_mean = s_adv[0], adv_den = s_adv[1];
  const float clipc = g.hp.clip_coef, invM = g.hp.inv_global_batch;
  const float vden = sqrtf(g.vrms_var[0] + 1e-8f), vmean = g.vrms_mean[0];
  const bool norm_adv = g.hp.norm_adv != 0, clip_vloss = g.hp.clip_vloss != 0;
  const float ent_coef_m = g.hp.ent_coef * invM, vf_half = g.hp.vf_coef * 0.5f;

  // after reduce16 the four lanes with slot(lane) == k hold the total of value k
  const int slot = ((lane >> 5) & 1) * 8 + ((lane >> 4) & 1) * 4 + ((lane >> 3) & 1) * 2 + ((lane >> 2) & 1);
  const bool mine = slot < A;
  const bool leader = mine && (lane & 3) == 0;    // one lane per action dim accumulates / publishes
  const float sd = mine ? expf(g.logstd[slot]) : 1.0f;
  const float var = sd * sd, lsd = logf(sd);
  const float ba = mine ? g.b4a[slot] : 0.0f;
  float ent_row = 0.0f;                           // entropy is state independent
  {
    const float e = mine ? kEntConst + lsd : 0.0f;
#pragma unroll
    for (int k = 0; k < VS; ++k) ent_row += lane_bcast(e, slot_lane(k));
  }
  float gls = 0.0f;                               // d loss / d logstd_k (leader lanes)
  float d_pg = 0.0f, d_v = 0.0f, d_ent = 0.0f, d_kl = 0.0f, d_okl = 0.0f, d_cf = 0.0f;
  float wc[CPL];
#pragma unroll
  for (int c = 0; c < CPL; ++c) wc[c] = g.W4c[lane * CPL + c];
  const float bc = g.b4c[0];

  // phase-2 ownership: weight column c2, slots [k0, k0+KPG)
  const int c2 = tid % HL, grp = tid / HL, k0 = grp * KPG;
  float acc[KPG];
#pragma unroll
  for (int kk = 0; kk < KPG; ++kk) acc[kk] = 0.0f;
  float accb = 0.0f;                              // bias gradients: threads 0..15 (one per slot)

  const int64_t n_tiles = (g.M + TR - 1) / TR;
  for (int64_t tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
    const int64_t row0 = tile * TR;
    const int rows = (int)((g.M - row0) < TR ? (g.M - row0) : TR);
    // ------------------------------------------------------------------ phase 1
    for (int rr = 0; rr < kHeadRowsPerWave; ++rr) {
      const int r = wave * kHeadRowsPerWave + rr;
      if (r >= rows) break;                        // wave-uniform
      const int64_t i = row0 + r;
      float hc[CPL], ha[CPL], part[16];
      load_vec<CPL>(g.Hc + i * HL + lane * CPL, hc);
      load_vec<CPL>(g.Ha + i * HL + lane * CPL, ha);
      const float a_taken = mine ? g.act[i * A + slot] : 0.0f;
      const float oldlogp = g.oldlogp[i], adv_raw = g.adv[i], R = g.ret_n[i], Vo = g.val_n[i];
      store_vec<CPL>(sHc + r * HL + lane * CPL, hc);
      store_vec<CPL>(sHa + r * HL + lane * CPL, ha);
#pragma unroll
      for (int k = 0; k < VS; ++k) {
        float d = 0.0f, wk[CPL];
        load_vec<CPL>(s_wa + k * HL + lane * CPL, wk);
#pragma unroll
        for (int c = 0; c < CPL; ++c) d = fmaf(ha[c], wk[c], d);
        part[k] = d;
      }
      {
        float d = 0.0f;
#pragma unroll
        for (int c = 0; c < CPL; ++c) d = fmaf(hc[c], wc[c], d);
        part[VS] = d;
      }
      const float tot = reduce16(part, lane);     // lanes of slot k: mu_k (k<A) / critic output (slot VS)
      const float mu = tot + ba;
      const float v = lane_bcast(tot, slot_lane(VS)) + bc;

      // ---- log-prob of the taken action
      const float diff = mine ? a_taken - mu : 0.0f;
      const float term = mine ? -(diff * diff) / (2.0f * var) - lsd - kHalfLog2Pi : 0.0f;
      float newlogp = 0.0f;
#pragma unroll
      for (int k = 0; k < VS; ++k) newlogp += lane_bcast(term, slot_lane(k));
      const float logratio = newlogp - oldlogp;
      const float ratio = expf(logratio);
      d_okl += -logratio;
      d_kl += (ratio - 1.0f) - logratio;
      d_cf += fabsf(ratio - 1.0f) > clipc ? 1.0f : 0.0f;

      const float adv = norm_adv ? (adv_raw - adv_mean) / adv_den : adv_raw;
      const float rc = ratio < 1.0f - clipc ? 1.0f - clipc : (ratio > 1.0f + clipc ? 1.0f + clipc : ratio);
      const float pg1 = -adv * ratio, pg2 = -adv * rc;
      const bool inside = ratio >= 1.0f - clipc && ratio <= 1.0f + clipc;
      // d max(pg1,pg2) / d ratio   (torch.max splits ties 1/2 : 1/2; clamp passes gradient inside only)
      const float dr_tie = 0.5f * -adv + (inside ? 0.5f * -adv : 0.0f);
      const float dr = pg1 > pg2 ? -adv : (pg1 < pg2 ? (inside ? -adv : 0.0f) : dr_tie);
      d_pg += pg1 > pg2 ? pg1 : pg2;
      const float g_logp = dr * ratio * invM;      // d loss / d newlogprob_i

      // ---- value head loss
      const float nv = (v - vmean) / vden;         // value_rms(newvalue, update=False)
      const float e1 = nv - R;
      const float vl1 = e1 * e1;
      const float dl = nv - Vo;
      const float cl = dl < -clipc ? -clipc : (dl > clipc ? clipc : dl);
      const float e2 = (Vo + cl) - R;
      const float vl2 = e2 * e2;
      const bool in2 = dl >= -clipc && dl <= clipc;
      const float dnv_c = vl1 > vl2 ? 2.0f * e1 : (vl1 < vl2 ? (in2 ? 2.0f * e2 : 0.0f) : e1 + (in2 ? e2 : 0.0f));
      const float vl = clip_vloss ? (vl1 > vl2 ? vl1 : vl2) : vl1;
      const float dnv = clip_vloss ? dnv_c : 2.0f * e1;
      d_v += 0.5f * vl;
      d_ent += ent_row;
      const float g_v = vf_half * dnv * invM / vden;   // d loss / d v_i

      // ---- backward through the heads
      const float gm = mine ? g_logp * diff / var : 0.0f;           // d loss / d mu_ik   (lanes of slot k)
      if (leader) {
        gls += g_logp * (diff * diff / var - 1.0f) - ent_coef_m;
        sG[r * 16 + slot] = gm;
      }
      if (lane == 63) sG[r * 16 + VS] = g_v;         // lane 63 has slot 15 = VS
      float dha[CPL];
#pragma unroll
      for (int c = 0; c < CPL; ++c) dha[c] = 0.0f;
#pragma unroll
      for (int k = 0; k < VS; ++k) {
        const float gmk = lane_bcast(gm, slot_lane(k));
        float wk[CPL];
        load_vec<CPL>(s_wa + k * HL + lane * CPL, wk);
#pragma unroll
        for (int c = 0; c < CPL; ++c) dha[c] = fmaf(gmk, wk[c], dha[c]);
      }
      float oa[CPL], oc[CPL];
#pragma unroll
      for (int c = 0; c < CPL; ++c) {
        oa[c] = dha[c] * (ha[c] > 0.0f ? 1.0f : ha[c] + 1.0f);
        oc[c] = (g_v * wc[c]) * (hc[c] > 0.0f ? 1.0f : hc[c] + 1.0f);
      }
      store_vec_wt<CPL>(g.dZa + i * HL + lane * CPL, oa);
      store_vec_wt<CPL>(g.dZc + i * HL + lane * CPL, oc);
    }
    __syncthreads();
    // ------------------------------------------------------------------ phase 2: dW4 += G^T . H
    for (int r = 0; r < rows; ++r) {
      const float ha2 = sHa[r * HL + c2], hc2 = sHc[r * HL + c2];
      float gk[KPG];
      load_vec<KPG>(sG + r * 16 + k0, gk);        // k0 is a multiple of KPG: one or two b128 broadcasts
#pragma unroll
      for (int kk = 0; kk < KPG; ++kk) acc[kk] = fmaf(gk[kk], (k0 + kk) == VS ? hc2 : ha2, acc[kk]);
    }
    if (tid < 16) {
      for (int r = 0; r < rows; ++r) accb += sG[r * 16 + tid];
    }
    __syncthreads();
  }

  // ---- per-block partials: weight gradients straight from the phase-2 registers, scalars through LDS
  float* pw = g.part_w + (int64_t)blockIdx.x * (A + 1) * HL;   // rows 0..A-1 = dW4a, row A = dW4c
#pragma unroll
  for (int kk = 0; kk < KPG; ++kk) {
    const int k = k0 + kk;
    if (k < A) pw[k * HL + c2] = acc[kk];
    else if (k == VS) pw[A * HL + c2] = acc[kk];
  }
  for (int w = 0; w < kHeadWaves; ++w) {           // fixed wave order => deterministic
    if (wave == w) {
      if (leader) ls[A + 1 + slot] = w == 0 ? gls : ls[A + 1 + slot] + gls;
      if (lane == 63) {
        float* dg = ls + 2 * A + 1;
        const float vals[kHeadDiag] = {d_pg, d_v, d_ent, 0.0f, d_kl, d_okl, d_cf, 0.0f};
#pragma unroll
        for (int q = 0; q < kHeadDiag; ++q) dg[q] = w == 0 ? vals[q] : dg[q] + vals[q];
      }
    }
    __syncthreads();
  }
  if (tid < A) ls[tid] = accb;                     // db4a[0..A-1]
  if (tid == VS) ls[A] = accb;                     // db4c
  __syncthreads();
  float* ps = g.part_s + (int64_t)blockIdx.x * NS;
  for (int o = tid; o < NS; o += NT) ps[o] = ls[o];
}

// ------------------------------------------------------------------------------- last hidden layer + heads + loss
// One launch instead of the last forward GEMM followed by head_loss_kernel (28 us at M = 16384 with no matrix work,
// 33 MB of last-layer activations written and read back): a workgroup owns 64 rows of ONE network over the full
// last-layer width, leaves H = elu(X W^T + b) in LDS (gemm::EPI_BIAS_ELU_LDS) and runs that network's head, its part
// of the PPO loss and the backward through the head on the tile.  The three head products are small GEMMs on the
// same fp32 MFMA (16 head outputs, rows past the real count zero):
//   A  Y[64,16]   = H[64,HL] . Wh^T          each wave a quarter of the contraction, quarters added in fixed order
//   -  row math   one thread per row: log-prob / ratio / clipped surrogate / d loss/d mu, or value loss / d loss/d v
//                 (the arithmetic of head_loss_kernel, ppo.py:299-345) -> G[64,16]
//   C  dWh[16,HL] = G^T . H                  contraction over the 64 rows; per-workgroup partial
//   B  dZ[64,HL]  = (G . Wh) * elu'(H)       written over H in LDS, every wave then streams out its own 32 x HL/2 region
// Partial rows [0, RB) belong to the actor workgroups, [RB, 2 RB) to the critic's (row layout of head_loss_kernel,
// each kind writes only its own entries; the fold reads them with separate base pointers).
// -DFWD_HEAD_TL (tools/fwd_head_timeline.py): thread 0 of every workgroup stamps the shader clock at the step boundaries
#ifdef FWD_HEAD_TL
__device__ unsigned long long* g_fhtl;    // [2 nets][1024 workgroups][8 stamps]
#define FH_TL(i) do { if (threadIdx.x == 0 && g_fhtl) g_fhtl[(blockIdx.z * 1024 + blockIdx.x) * 8 + (i)] = wall_clock64(); } while (0)
#else
#define FH_TL(i) do { } while (0)
#endif

template <int HL>
constexpr size_t fwd_head_lds_floats() { return (size_t)64 * (HL + gemm::kLdsTilePad) + 64 * 16 + 64 * 16 + 64 * 8 + 4; }

template <int HL, int PREC = 0>      // PREC: operand precision of the hidden-layer GEMM (gemm_body); the head products stay fp32
__global__ __launch_bounds__(256, 2) void fwd_head_kernel(const Params p, const HeadArgs g) {   // two workgroups per CU
  using gemm::f32x16;
  constexpr int BM = 64, LD = HL + gemm::kLdsTilePad;
  constexpr int KQ = HL / 4;            // contraction share of a wave in step A
  constexpr int TNB = HL / 64;          // 32-column tiles per wave in step B (waves 2 x 2)
  constexpr int TNC = HL / 128;         // 32-column tiles per wave in step C (waves 1 x 4)
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* Hs = smem;                     // [64][LD]   activated tile, later dZ
  float* sG = Hs + BM * LD;             // [64][16]   d loss / d head output k of row r (zero beyond the real outputs)
  float* sMu = sG + BM * 16;            // [64][16]   head outputs, later the per-row d loss / d logstd_k terms
  float* sD = sMu + BM * 16;            // [64][8]    per-row diagnostics {pg, v, ent, -, kl, old_kl, clipfrac, -}
  float* s_adv = sD + BM * 8;           // [2]        advantage mean, std + 1e-8
  const int net = blockIdx.z;           // 0 critic, 1 actor (Params::op order)
  const int tid = threadIdx.x, lane = tid & 63, l31 = lane & 31, h = lane >> 5;
  const int q = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int A = g.A;
  const int RB = gridDim.x;
  const int tile = gemm::xcd_tile_index(blockIdx.x, gridDim.x);
  const int64_t i0 = (int64_t)tile * BM;
  const int rows = (int)((g.M - i0) < BM ? (g.M - i0) : BM);
  const int NS = 2 * A + 1 + kHeadDiag;

  if (net == 1 && tid < 64) {           // advantage statistics over the minibatch (ppo.py:314-318): mean, unbiased std
    if (g.hp.norm_adv && g.adv_stats == nullptr) {
      double a1 = 0.0, a2 = 0.0;
      for (int b = lane; b < g.n_adv_part; b += 64) {
        a1 += g.adv_part[2 * b];
        a2 += g.adv_part[2 * b + 1];
      }
      a1 = wave_sum_d(a1);
      a2 = wave_sum_d(a2);
      if (lane == 0) {
        const double n = (double)g.M;
        const double mean = a1 / n;
        double var = (a2 - n * mean * mean) / (n - 1.0);   // NaN for n == 1, like torch.std()
        if (var < 0.0) var = 0.0;
        s_adv[0] = (float)mean;
        s_adv[1] = (float)sqrt(var) + 1e-8f;
      }
    } else if (lane == 0) {
      s_adv[0] = g.adv_stats ? g.adv_stats[0] : 0.0f;
      s_adv[1] = g.adv_stats ? g.adv_stats[1] : 1.0f;
    }
  }

  FH_TL(0);
  const float* Wh = net == 1 ? g.W4a : g.W4c;          // [KH][HL] head weights of this network
  const int KH = net == 1 ? A : 1;
  const float4 zero4 = make_float4(0.f, 0.f, 0.f, 0.f);
  const int wm = q >> 1, wn = q & 1;
  // Everything the epilogue reads from global memory is requested here, ahead of the main loop: the head weights in
  // the operand layouts of steps A and B (rows past KH zero) and the gathered scalars of the row this thread will
  // work on (row math: four threads per row, thread part pp owns the action dims pp, pp+4, pp+8, pp+12).
  float4 bw[KQ / 8];
#pragma unroll
  for (int kb = 0; kb < KQ / 8; ++kb)
    bw[kb] = l31 < KH ? *reinterpret_cast<const float4*>(Wh + l31 * HL + q * KQ + 8 * kb + 4 * h) : zero4;
  float bwB[TNB][2][4];
#pragma unroll
  for (int tn = 0; tn < TNB; ++tn)
#pragma unroll
    for (int blk = 0; blk < 2; ++blk)
#pragma unroll
      for (int sx = 0; sx < 4; ++sx) {
        const int kk = 8 * blk + 4 * h + sx;
        bwB[tn][blk][sx] = kk < KH ? Wh[kk * HL + wn * (HL / 2) + 32 * tn + l31] : 0.0f;
      }
  const int rr = tid >> 2, pp = tid & 3;               // row math: row, part
  const bool rvalid = rr < rows;
  const int64_t ri = i0 + (rvalid ? rr : 0);
  const float rs0 = net == 1 ? g.oldlogp[ri] : g.ret_n[ri];
  const float rs1 = net == 1 ? g.adv[ri] : g.val_n[ri];
  float ract[4], rls[4], rb[4];
#pragma unroll
  for (int kk = 0; kk < 4; ++kk) {
    const int k = pp + 4 * kk;
    const bool on = net == 1 && k < A;
    ract[kk] = on ? g.act[ri * A + k] : 0.0f;
    rls[kk] = on ? g.logstd[k] : 0.0f;
    rb[kk] = on ? g.b4a[k] : 0.0f;
  }
  const float rbc = g.b4c[0], rvv = g.vrms_var[0], rvm = g.vrms_mean[0];

  gemm::gemm_body<BM, HL, true, true, gemm::EPI_BIAS_ELU_LDS, gemm::BK, PREC, 1>(p, tile, blockIdx.z, smem);
  FH_TL(1);

  // ---- A: head outputs.  MFMA step (blk, s) of lane-half h contracts k = 8 blk + 4 h + s - the same permutation on
  //         both operands (gemm_body's K-contiguous fragments)
  {
    __syncthreads();                                     // H tile complete
    f32x16 c0, c1;
#pragma unroll
    for (int r = 0; r < 16; ++r) c0[r] = 0.0f, c1[r] = 0.0f;
#pragma unroll
    for (int kb = 0; kb < KQ / 8; ++kb) {
      const float4 a0 = *reinterpret_cast<const float4*>(Hs + l31 * LD + q * KQ + 8 * kb + 4 * h);
      const float4 a1 = *reinterpret_cast<const float4*>(Hs + (32 + l31) * LD + q * KQ + 8 * kb + 4 * h);
      c0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a0.x, bw[kb].x, c0, 0, 0, 0);
      c1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a1.x, bw[kb].x, c1, 0, 0, 0);
      c0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a0.y, bw[kb].y, c0, 0, 0, 0);
      c1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a1.y, bw[kb].y, c1, 0, 0, 0);
      c0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a0.z, bw[kb].z, c0, 0, 0, 0);
      c1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a1.z, bw[kb].z, c1, 0, 0, 0);
      c0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a0.w, bw[kb].w, c0, 0, 0, 0);
      c1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a1.w, bw[kb].w, c1, 0, 0, 0);
    }
    // the four contraction quarters in fixed order, two rounds: sMu = q0 + q1, sG = q2 + q3 (sG is free until the row
    // math writes it); the row math adds the two halves
    for (int w = 0; w < 2; ++w) {
      if ((q & 1) == w && l31 < 16) {
        float* half = (q >> 1) ? sG : sMu;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int row = (r & 3) + 8 * (r >> 2) + 4 * h;
          float* d0 = half + row * 16 + l31;
          float* d1 = half + (32 + row) * 16 + l31;
          *d0 = (w == 0 ? 0.0f : *d0) + c0[r];
          *d1 = (w == 0 ? 0.0f : *d1) + c1[r];
        }
      }
      __syncthreads();
    }
  }

  FH_TL(2);
  const float clipc = g.hp.clip_coef, invM = g.hp.inv_global_batch;
  // ---- row math: four threads per row
  {
    const int r = rr;
    float dg[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) dg[e] = 0.0f;
    float gm[4] = {0.f, 0.f, 0.f, 0.f}, gl[4] = {0.f, 0.f, 0.f, 0.f};
    if (net == 1) {
      float diff[4], var[4];
      float lp = 0.0f, en = 0.0f;
#pragma unroll
      for (int kk = 0; kk < 4; ++kk) {
        const int k = pp + 4 * kk;
        diff[kk] = 0.0f, var[kk] = 1.0f;
        if (k < A) {
          const float sd = expf(rls[kk]);
          const float lsd = logf(sd);
          var[kk] = sd * sd;
          const float mu = (sMu[r * 16 + k] + sG[r * 16 + k]) + rb[kk];
          diff[kk] = ract[kk] - mu;
          lp += -(diff[kk] * diff[kk]) / (2.0f * var[kk]) - lsd - kHalfLog2Pi;
          en += kEntConst + lsd;
        }
      }
      lp += __shfl_xor(lp, 1, 64), en += __shfl_xor(en, 1, 64);      // the four parts of a row sit in adjacent lanes
      lp += __shfl_xor(lp, 2, 64), en += __shfl_xor(en, 2, 64);
      if (rvalid) {
        const float adv_mean = s_adv[0], adv_den = s_adv[1];
        const bool norm_adv = g.hp.norm_adv != 0;
        const float ent_coef_m = g.hp.ent_coef * invM;
        const float logratio = lp - rs0;
        const float ratio = expf(logratio);
        dg[5] = -logratio;
        dg[4] = (ratio - 1.0f) - logratio;
        dg[6] = fabsf(ratio - 1.0f) > clipc ? 1.0f : 0.0f;
        const float adv = norm_adv ? (rs1 - adv_mean) / adv_den : rs1;
        const float rc = ratio < 1.0f - clipc ? 1.0f - clipc : (ratio > 1.0f + clipc ? 1.0f + clipc : ratio);
        const float pg1 = -adv * ratio, pg2 = -adv * rc;
        const bool inside = ratio >= 1.0f - clipc && ratio <= 1.0f + clipc;
        // d max(pg1,pg2) / d ratio   (torch.max splits ties 1/2 : 1/2; clamp passes gradient inside only)
        const float dr_tie = 0.5f * -adv + (inside ? 0.5f * -adv : 0.0f);
        const float dr = pg1 > pg2 ? -adv : (pg1 < pg2 ? (inside ? -adv : 0.0f) : dr_tie);
        dg[0] = pg1 > pg2 ? pg1 : pg2;
        dg[2] = en;
        const float g_logp = dr * ratio * invM;      // d loss / d newlogprob_i
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
          if (pp + 4 * kk < A) {
            gm[kk] = g_logp * diff[kk] / var[kk];                                    // d loss / d mu_ik
            gl[kk] = g_logp * (diff[kk] * diff[kk] / var[kk] - 1.0f) - ent_coef_m;   // row's share of d loss / d logstd_k
          }
        }
      }
    } else if (rvalid && pp == 0) {
      const bool clip_vloss = g.hp.clip_vloss != 0;
      const float vden = sqrtf(rvv + 1e-8f), vmean = rvm;
      const float vf_half = g.hp.vf_coef * 0.5f;
      const float R = rs0, Vo = rs1;
      const float v = (sMu[r * 16] + sG[r * 16]) + rbc;
      const float nv = (v - vmean) / vden;         // value_rms(newvalue, update=False)
      const float e1 = nv - R;
      const float vl1 = e1 * e1;
      const float dl = nv - Vo;
      const float cl = dl < -clipc ? -clipc : (dl > clipc ? clipc : dl);
      const float e2 = (Vo + cl) - R;
      const float vl2 = e2 * e2;
      const bool in2 = dl >= -clipc && dl <= clipc;
      const float dnv_c = vl1 > vl2 ? 2.0f * e1 : (vl1 < vl2 ? (in2 ? 2.0f * e2 : 0.0f) : e1 + (in2 ? e2 : 0.0f));
      const float vl = clip_vloss ? (vl1 > vl2 ? vl1 : vl2) : vl1;
      const float dnv = clip_vloss ? dnv_c : 2.0f * e1;
      dg[1] = 0.5f * vl;
      gm[0] = vf_half * dnv * invM / vden;         // d loss / d v_i  (slot 0)
    }
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) sG[r * 16 + pp + 4 * kk] = gm[kk], sMu[r * 16 + pp + 4 * kk] = gl[kk];
    if (pp == 0) {
#pragma unroll
      for (int e = 0; e < 8; ++e) sD[r * 8 + e] = dg[e];
    }
  }
  __syncthreads();

  FH_TL(3);
  // ---- C: head weight gradient of the tile, dWh[k][c] = sum_r G[r][k] H[r][c]; wave q owns TNC column tiles
  const int prow = net == 1 ? tile : RB + tile;
  {
    f32x16 cc[TNC];
#pragma unroll
    for (int t = 0; t < TNC; ++t)
#pragma unroll
      for (int r = 0; r < 16; ++r) cc[t][r] = 0.0f;
    const int col0 = q * (HL / 4);
#pragma unroll 8
    for (int s = 0; s < BM / 2; ++s) {
      const int r = 2 * s + h;
      const float a = l31 < 16 ? sG[r * 16 + l31] : 0.0f;
#pragma unroll
      for (int t = 0; t < TNC; ++t)
        cc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, Hs[r * LD + col0 + 32 * t + l31], cc[t], 0, 0, 0);
    }
    float* pw = g.part_w + (int64_t)prow * (A + 1) * HL + (net == 1 ? 0 : (int64_t)A * HL);   // rows 0..A-1 = dW4a, row A = dW4c
#pragma unroll
    for (int t = 0; t < TNC; ++t)
#pragma unroll
      for (int r = 0; r < 8; ++r) {                      // accumulator rows 0..15 = head outputs
        const int k = (r & 3) + 8 * (r >> 2) + 4 * h;
        if (k < KH) pw[k * HL + col0 + 32 * t + l31] = cc[t][r];
      }
  }
  __syncthreads();                                       // every read of H is done: step B overwrites it
  FH_TL(4);

  // ---- B: dZ = (G . Wh) * elu'(H), in place
  {
    f32x16 cb[TNB];
#pragma unroll
    for (int tn = 0; tn < TNB; ++tn)
#pragma unroll
      for (int r = 0; r < 16; ++r) cb[tn][r] = 0.0f;
#pragma unroll
    for (int blk = 0; blk < 2; ++blk) {
      const float4 a4 = *reinterpret_cast<const float4*>(sG + (32 * wm + l31) * 16 + 8 * blk + 4 * h);
      const float av[4] = {a4.x, a4.y, a4.z, a4.w};
#pragma unroll
      for (int s = 0; s < 4; ++s)
#pragma unroll
        for (int tn = 0; tn < TNB; ++tn)
          cb[tn] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[s], bwB[tn][blk][s], cb[tn], 0, 0, 0);
    }
#pragma unroll
    for (int tn = 0; tn < TNB; ++tn)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = 32 * wm + (r & 3) + 8 * (r >> 2) + 4 * h;
        float* hp = Hs + row * LD + wn * (HL / 2) + 32 * tn + l31;
        const float hv = *hp;
        *hp = cb[tn][r] * (hv > 0.0f ? 1.0f : hv + 1.0f);      // elu'(z) = 1 (z>0) | elu(z) + 1
      }
  }
  // every wave streams out its own 32 x HL/2 region of dZ (LDS operations of a wave execute in order: no workgroup
  // barrier between its in-place writes and these reads)
  __builtin_amdgcn_wave_barrier();
  FH_TL(5);
  {
    constexpr int C4 = HL / 8;                 // float4 chunks per region row
    constexpr int RPI = 64 / C4;               // region rows per store instruction
    const int c4 = lane % C4, r_in = lane / C4;
    float* dZ = net == 1 ? g.dZa : g.dZc;
#pragma unroll 4
    for (int rb = 0; rb < 32; rb += RPI) {
      const int r = 32 * wm + rb + r_in;
      if (r < rows) {
        const float4 v = *reinterpret_cast<const float4*>(Hs + r * LD + wn * (HL / 2) + 4 * c4);
        const float o[4] = {v.x, v.y, v.z, v.w};
        store_vec_wt<4>(dZ + (i0 + r) * HL + wn * (HL / 2) + 4 * c4, o);
      }
    }
  }
  // ---- scalars of the tile: bias / logstd gradients, diagnostics.  16 row groups of 4 rows, combined in fixed order
  //      through LDS (the H / dZ tile is free again once every row has been streamed out)
  __syncthreads();
  FH_TL(6);
  {
    float* red = Hs;                                    // [16 groups][40]: 16 db, 16 dlogstd, 8 diag
    const int k = tid & 15, grp = tid >> 4;
    float db = 0.0f, dl = 0.0f, dd = 0.0f;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int r = 4 * grp + j;
      db += sG[r * 16 + k], dl += sMu[r * 16 + k];
      if (k < 8) dd += sD[r * 8 + k];
    }
    red[grp * 40 + k] = db, red[grp * 40 + 16 + k] = dl;
    if (k < 8) red[grp * 40 + 32 + k] = dd;
  }
  __syncthreads();
  float* ps = g.part_s + (int64_t)prow * NS;
  if (tid < 40) {
    float v = 0.0f;
#pragma unroll
    for (int grp = 0; grp < 16; ++grp) v += Hs[grp * 40 + tid];
    if (tid < 16) {
      if (net == 1) { if (tid < A) ps[tid] = v; }          // db4a[k]
      else if (tid == 0) ps[A] = v;                        // db4c
    } else if (tid < 32) {
      if (net == 1 && tid - 16 < A) ps[A + 1 + tid - 16] = v;   // dlogstd[k]
    } else {
      ps[2 * A + 1 + tid - 32] = v;                        // diagnostics
    }
  }
  FH_TL(7);
}

#ifdef FWD_HEAD_TL
extern "C" int catppo_debug_fwd_head_tl(void* buf) {     // timeline builds only: not part of include/catppo.h
  unsigned long long* pbuf = static_cast<unsigned long long*>(buf);
  return hipMemcpyToSymbol(HIP_SYMBOL(g_fhtl), &pbuf, sizeof(pbuf)) == hipSuccess ? 0 : -1;
}
#endif

// ------------------------------------------------------------------------------- segmented partial reduction
// dst[e] (+)= scale * sum_{p<n_parts} src[p*stride + e]   in fixed order.  One launch handles every segment
// (all split-K weight/bias partials, the head partials and the diagnostics).
constexpr int kMaxSegs = 24;
struct Seg {
  const float* src;
  float* dst;
  int64_t count;
  int64_t stride;
  int n_parts;
  int mode;     // 0: dst = sum, 1: dst += sum * scale (diagnostics)
  float scale;
};
struct SegTable {
  int n;
  Seg s[kMaxSegs];
};

// block = EL lanes x G part-groups (EL*G = 256).  Thread (e,g) adds parts g, g+G, ... in order, the G group sums are
// then combined in LDS in fixed order => deterministic.  Aligned segments (every weight / bias partial): a lane owns FOUR
// consecutive elements (16-byte loads) and its parts are requested in batches of four or eight that are always full -
// a batch past the last part re-reads the last part and adds 0 - so the loads of a batch are in flight together
// whatever the split count (the unrolled loop of rounds 1-2 fell into its serial remainder for the 2-4 parts per thread
// of a small minibatch).  The order of the additions per element is unchanged: results are bit-identical.
template <int NB>
__device__ __forceinline__ float4 seg_sum4(const float* __restrict__ src, int64_t stride, int g, int G, int last) {
  float4 a = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
  for (int q0 = g; q0 <= last; q0 += NB * G) {
    float4 x[NB];
#pragma unroll
    for (int j = 0; j < NB; ++j) {
      const int q = q0 + j * G;
      x[j] = *reinterpret_cast<const float4*>(src + (int64_t)(q < last ? q : last) * stride);
    }
#pragma unroll
    for (int j = 0; j < NB; ++j) {
      const bool on = q0 + j * G <= last;
      a.x += on ? x[j].x : 0.0f, a.y += on ? x[j].y : 0.0f, a.z += on ? x[j].z : 0.0f, a.w += on ? x[j].w : 0.0f;
    }
  }
  return a;
}

// one workgroup's share of one segment: workgroup bx of nbx walks the segment's elements (sm: 1024 floats of LDS)
// Returns the fp64 sum of squares of the gradient elements THIS thread wrote (mode 0 only): the launches that fold the
// gradient can emit the squared-norm partials of the clip on the way (NormEmit below).
__device__ __forceinline__ double seg_reduce_body(const Seg sg, const int bx, const int nbx, float* __restrict__ sm,
                                                  const float ent_coef, const float vf_coef) {
  double ss = 0.0;
  // few wide partials (split-K): 4 part groups x 64 lanes; many narrow ones (head): 16 x 16
  const int G = sg.n_parts >= 128 ? 16 : 4;
  const int EL = 256 / G;
  const int el = threadIdx.x % EL, g = threadIdx.x / EL;
  const bool vec = sg.mode == 0 && (reinterpret_cast<uintptr_t>(sg.src) & 15) == 0 && sg.stride % 4 == 0 &&
                   sg.count % 4 == 0 && (reinterpret_cast<uintptr_t>(sg.dst) & 15) == 0;
  if (vec) {
    const int last = sg.n_parts - 1;
    const bool few = (sg.n_parts + G - 1) / G <= 4;         // parts per thread
    float4* sm4 = reinterpret_cast<float4*>(sm);
    for (int64_t e0 = (int64_t)bx * EL * 4; e0 < sg.count; e0 += (int64_t)nbx * EL * 4) {
      const int64_t e = e0 + el * 4;
      float4 a = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
      if (e < sg.count && g <= last) a = few ? seg_sum4<4>(sg.src + e, sg.stride, g, G, last)
                                              : seg_sum4<8>(sg.src + e, sg.stride, g, G, last);
      sm4[threadIdx.x] = a;
      __syncthreads();
      if (g == 0 && e < sg.count) {
        for (int gg = 1; gg < G; ++gg) {
          const float4 y = sm4[gg * EL + el];
          a.x += y.x, a.y += y.y, a.z += y.z, a.w += y.w;
        }
        *reinterpret_cast<float4*>(sg.dst + e) = a;
        ss += (double)a.x * (double)a.x;
        ss += (double)a.y * (double)a.y;
        ss += (double)a.z * (double)a.z;
        ss += (double)a.w * (double)a.w;
      }
      __syncthreads();
    }
    return ss;
  }
  for (int64_t e0 = (int64_t)bx * EL; e0 < sg.count; e0 += (int64_t)nbx * EL) {
    const int64_t e = e0 + el;
    float a = 0.0f;
    if (e < sg.count) {
#pragma unroll 8
      for (int p = g; p < sg.n_parts; p += G) a += sg.src[(int64_t)p * sg.stride + e];
    }
    sm[threadIdx.x] = a;
    __syncthreads();
    if (g == 0) {
      for (int gg = 1; gg < G; ++gg) a += sm[gg * EL + el];
    }
    __syncthreads();
    if (g == 0) sm[el] = a;          // combined sums, visible to the whole block
    __syncthreads();
    if (g == 0 && e < sg.count) {
      if (sg.mode == 0) {
        sg.dst[e] = a;
        ss += (double)a * (double)a;
      } else {
        // diagnostics block {pg, v, ent, loss, kl, old_kl, clipfrac, count} (count = 8 <= EL: one block)
        float v = a * sg.scale;
        if (e == 3) v = (sm[0] - ent_coef * sm[2] + sm[1] * vf_coef) * sg.scale;   // pg - ENT*entropy + v_loss*VF
        if (e == 7) v = 1.0f;                                                      // minibatches accumulated
        sg.dst[e] = sg.dst[e] + v;
      }
    }
    __syncthreads();
  }
  return ss;
}

// The clip of an optimiser step needs ||grad||^2 (cleanrl/ppo.py:354, clip_grad_norm_): a launch of its own that
// re-reads the gradient the fold launches have just written - 5 us per step for 1.2 MB.  With NormEmit.part set, every
// workgroup that folds a piece of the gradient also writes the fp64 sum of squares of that piece into its own slot
// (fixed slot per workgroup => the final sum has a fixed order), and one thread of the last fold launch advances the
// Adam step count and prepares the bias corrections (what sqnorm_partial_step_kernel does beside its loads).
// catppo_ppo_minibatch_step_packed then goes straight to the Adam launch.
struct NormEmit {
  double* part = nullptr;            // [kNormSlots]; nullptr: off
  catppo_iter_state* st = nullptr;
  double beta1 = 0.0, beta2 = 0.0;
  int n_slots = 0;                   // slots written so far by the launches of this step (host side)
};
static_assert(kNormSlots >= 256 * kMaxSegs, "one squared-norm slot per fold workgroup");

__device__ __forceinline__ void emit_norm_slot(double ss, double* __restrict__ slot, float* __restrict__ sm) {
  ss = wave_sum_d(ss);
  double* d = reinterpret_cast<double*>(sm);
  __syncthreads();                   // sm is free (seg_reduce_body ends behind a barrier; belt and braces)
  if ((threadIdx.x & 63) == 0) d[threadIdx.x >> 6] = ss;
  __syncthreads();
  if (threadIdx.x == 0) *slot = (d[0] + d[1]) + (d[2] + d[3]);
}

// torch.optim.Adam: bias_correction = 1 - beta ** step (Python doubles), step_size = lr / bias_correction1
__device__ __forceinline__ void adam_advance_step(catppo_iter_state* __restrict__ st, double beta1, double beta2) {
  const int64_t step_i = st->adam_step + 1;
  const double step = (double)step_i;
  const double bc1 = 1.0 - pow(beta1, step);
  const double bc2 = 1.0 - pow(beta2, step);
  st->adam_step = step_i;
  st->adam_step_size = (float)(st->lr / bc1);
  st->adam_bc2_sqrt = (float)sqrt(bc2);
}

__global__ __launch_bounds__(256) void seg_reduce_kernel(const SegTable t, float ent_coef, float vf_coef,
                                                         double* __restrict__ norm_slots, catppo_iter_state* st,
                                                         double beta1, double beta2) {
  __shared__ __attribute__((aligned(16))) float sm[1024];
  if (st != nullptr && blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 64) adam_advance_step(st, beta1, beta2);
  const double ss = seg_reduce_body(t.s[blockIdx.y], blockIdx.x, gridDim.x, sm, ent_coef, vf_coef);
  if (norm_slots != nullptr) emit_norm_slot(ss, norm_slots + blockIdx.y * gridDim.x + blockIdx.x, sm);
}

// The first layer's weight-gradient GEMM and the fold of every OTHER layer's partials in one launch (round 4).  dW_0 is
// the last GEMM of an optimiser step (it needs dZ_0, the output of the last paired launch) and a light one (0.8 GFLOP,
// 37 MB); the partials of the layers above it have been complete since their own launches.  Their fold (43 MB of
// streaming reads, no matrix work) used to wait behind it in a launch of its own; here its workgroups fill the CUs
// beside the GEMM's, the way the paired launches mix long and short workgroups.  Workgroups [0, n_gemm) run the GEMM
// (launch order first: they are resident from the start), the rest fold: kFoldX workgroups per segment.
constexpr int kFoldX = 256;
__global__ __launch_bounds__(256) void dw_fold_kernel(const gemm::Params p, const SegTable t, const int gemm_tiles,
                                                      const int n_gemm, float ent_coef, float vf_coef,
                                                      double* __restrict__ norm_slots) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int b = blockIdx.x;
  if (b < n_gemm) {
    const gemm::TileId id = gemm::xcd_tile_of(b, gemm_tiles, n_gemm / gemm_tiles, p.xcd_legacy);
    gemm::gemm_body<64, 64, false, false, gemm::EPI_PARTIAL>(p, id.tile, id.bz, smem);
  } else {
    const int f = b - n_gemm;
    const double ss = seg_reduce_body(t.s[f / kFoldX], f % kFoldX, kFoldX, smem, ent_coef, vf_coef);
    if (norm_slots != nullptr) emit_norm_slot(ss, norm_slots + f, smem);
  }
}

// ------------------------------------------------------------------------------- clip + Adam
// Elementwise tails of an optimiser step.  Both are a few hundred K elements behind a launch: what they cost is load
// round trips in sequence, so a thread takes FOUR consecutive elements per pass (16-byte accesses when the arrays are
// 16-byte aligned, as torch's are) instead of one element on each of four passes.
__device__ __forceinline__ double sqnorm_of_thread(const float* __restrict__ g, int64_t n) {
  double a = 0.0;
  const int64_t tid = (int64_t)blockIdx.x * 256 + threadIdx.x, nth = (int64_t)gridDim.x * 256;
  int64_t done = 0;
  if ((reinterpret_cast<uintptr_t>(g) & 15) == 0) {
    const int64_t n4 = n / 4;
    for (int64_t i = tid; i < n4; i += nth) {
      const float4 x = reinterpret_cast<const float4*>(g)[i];
      a += (double)x.x * (double)x.x;
      a += (double)x.y * (double)x.y;
      a += (double)x.z * (double)x.z;
      a += (double)x.w * (double)x.w;
    }
    done = n4 * 4;
  }
  for (int64_t e = done + tid; e < n; e += nth) {
    const double v = (double)g[e];
    a += v * v;
  }
  return a;
}

struct AdamCoef {
  float coef, one_m_b1, b2, one_m_b2, eps, step_size, bc2_sqrt;
};
// clip + Adam of one element: g <- g*coef; exp_avg.lerp_(g, 1-b1); exp_avg_sq.mul_(b2).addcmul_(g, g, 1-b2);
// param.addcdiv_(exp_avg, sqrt(exp_avg_sq)/sqrt(bc2) + eps, -step_size)
__device__ __forceinline__ void adam_elem(float& p, float& g, float& m, float& v, const AdamCoef& c) {
  const float gr = g * c.coef;
  g = gr;
  m = m + (gr - m) * c.one_m_b1;
  float vv = v * c.b2;
  vv = vv + c.one_m_b2 * gr * gr;
  v = vv;
  const float denom = sqrtf(vv) / c.bc2_sqrt + c.eps;
  p = p + (-c.step_size * m) / denom;
}
__device__ __forceinline__ void adam_all(float* __restrict__ p, float* __restrict__ g, float* __restrict__ m,
                                         float* __restrict__ v, int64_t n, const AdamCoef& c) {
  const int64_t tid = (int64_t)blockIdx.x * 256 + threadIdx.x, nth = (int64_t)gridDim.x * 256;
  int64_t done = 0;
  if (((reinterpret_cast<uintptr_t>(p) | reinterpret_cast<uintptr_t>(g) | reinterpret_cast<uintptr_t>(m) |
        reinterpret_cast<uintptr_t>(v)) & 15) == 0) {
    const int64_t n4 = n / 4;
    for (int64_t i = tid; i < n4; i += nth) {
      float4 P = reinterpret_cast<float4*>(p)[i], Gd = reinterpret_cast<float4*>(g)[i];
      float4 Mo = reinterpret_cast<float4*>(m)[i], V = reinterpret_cast<float4*>(v)[i];
      adam_elem(P.x, Gd.x, Mo.x, V.x, c);
      adam_elem(P.y, Gd.y, Mo.y, V.y, c);
      adam_elem(P.z, Gd.z, Mo.z, V.z, c);
      adam_elem(P.w, Gd.w, Mo.w, V.w, c);
      reinterpret_cast<float4*>(g)[i] = Gd;
      reinterpret_cast<float4*>(m)[i] = Mo;
      reinterpret_cast<float4*>(v)[i] = V;
      reinterpret_cast<float4*>(p)[i] = P;
    }
    done = n4 * 4;
  }
  for (int64_t e = done + tid; e < n; e += nth) adam_elem(p[e], g[e], m[e], v[e], c);
}

__global__ __launch_bounds__(256) void sqnorm_partial_kernel(const float* __restrict__ g, int64_t n,
                                                             double* __restrict__ part) {
  __shared__ double sm[4];
  double a = sqnorm_of_thread(g, n);
  a = wave_sum_d(a);
  if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6] = a;
  __syncthreads();
  if (threadIdx.x == 0) part[blockIdx.x] = (sm[0] + sm[1]) + (sm[2] + sm[3]);
}

__global__ __launch_bounds__(256) void clip_adam_kernel(float* __restrict__ p, float* __restrict__ g,
                                                        float* __restrict__ m, float* __restrict__ v, int64_t n,
                                                        const double* __restrict__ norm_part, int n_part,
                                                        float max_norm, float beta1, float beta2, float one_m_b1,
                                                        float one_m_b2, float eps, float step_size,
                                                        float bc2_sqrt) {
  __shared__ float s_coef;
  if (threadIdx.x < 64) {
    double a = 0.0;
    for (int b = threadIdx.x; b < n_part; b += 64) a += norm_part[b];
    a = wave_sum_d(a);
    if (threadIdx.x == 0) {
      const float total = (float)sqrt(a);
      const float c = max_norm / (total + 1e-6f);     // clip_grad_norm_: max_norm / (total_norm + 1e-6)
      s_coef = c > 1.0f ? 1.0f : c;                   //                  clamped to 1
    }
  }
  __syncthreads();
  const AdamCoef c{s_coef, one_m_b1, beta2, one_m_b2, eps, step_size, bc2_sqrt};
  adam_all(p, g, m, v, n, c);
}

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
  static const int force_nets = env_int("CATPPO_ROWS_NETS", 0);       // A/B: 1 = always one network per workgroup
  static const int store_policy = env_int("CATPPO_ROWS_STORE", 0);
  a.nets_per_wg = force_nets ? force_nets : (tiles >= n_cu ? 2 : 1);
  a.store_policy = store_policy;
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
  static const int force_nets = env_int("CATPPO_ROWS_NETS", 0);
  a.nets_per_wg = force_nets ? force_nets : (tiles >= n_cu ? 2 : 1);
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
      CATPPO_CHECK_LAUNCH(ctx);
      return CATPPO_OK;
    }
  }
  forward_hidden(shape, L, params, x, N, w, 0, critic_only ? 1 : 2, s);
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

extern "C" int catppo_policy_act(catppo_ctx* ctx, const catppo_mlp_shape* shape, const float* params,
                                 const float* x, int64_t N, const float* eps, const float* given_action,
                                 float* action, float* logprob, float* value, void* stream) {
  return policy_core(ctx, shape, params, x, N, eps, given_action, action, logprob, value, CATPPO_F32, nullptr, 0,
                     nullptr, false, stream, __func__);
}

extern "C" int catppo_policy_act_ex(catppo_ctx* ctx, const catppo_mlp_shape* shape, const float* params,
                                    const float* x, int64_t N, const float* eps, const float* given_action,
                                    float* action, float* logprob, void* value, int value_dtype, void* stream) {
  return policy_core(ctx, shape, params, x, N, eps, given_action, action, logprob, value, value_dtype, nullptr, 0,
                     nullptr, false, stream, __func__);
}

extern "C" int catppo_policy_act_rng(catppo_ctx* ctx, const catppo_mlp_shape* shape, const float* params,
                                     const float* x, int64_t N, const catppo_iter_state* state, int32_t step,
                                     float* eps_out, float* action, float* logprob, void* value, int value_dtype,
                                     void* stream) {
  CATPPO_CHECK_ARG(ctx, ctx != nullptr);
  CATPPO_CHECK_ARG(ctx, state != nullptr && step >= 0);
  return policy_core(ctx, shape, params, x, N, nullptr, nullptr, action, logprob, value, value_dtype, state, step,
                     eps_out, false, stream, __func__);
}

extern "C" int catppo_value(catppo_ctx* ctx, const catppo_mlp_shape* shape, const float* params, const float* x,
                            int64_t N, float* value, void* stream) {
  return policy_core(ctx, shape, params, x, N, nullptr, nullptr, nullptr, nullptr, value, CATPPO_F32, nullptr, 0,
                     nullptr, true, stream, __func__);
}

extern "C" int catppo_value_ex(catppo_ctx* ctx, const catppo_mlp_shape* shape, const float* params, const float* x,
                               int64_t N, void* value, int value_dtype, void* stream) {
  return policy_core(ctx, shape, params, x, N, nullptr, nullptr, nullptr, nullptr, value, value_dtype, nullptr, 0,
                     nullptr, true, stream, __func__);
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

extern "C" int catppo_ppo_gather(catppo_ctx* ctx, const catppo_mlp_shape* shape, const float* b_obs,
                                 const float* b_actions, const float* b_logprobs, const float* b_advantages,
                                 const float* b_returns_n, const float* b_values_n, const int64_t* inds,
                                 int64_t total, int64_t M, float* x_g, float* act_g, float* scal_g,
                                 double* adv_part_g, void* stream) {
  CATPPO_CHECK_ARG(ctx, ctx != nullptr);
  catppo_mlp_layout L;
  CATPPO_CHECK_ARG(ctx, shape && catppo_mlp_layout_of(shape, &L) == CATPPO_OK);
  CATPPO_CHECK_ARG(ctx, b_obs && b_actions && b_logprobs && b_advantages && b_returns_n && b_values_n && inds);
  CATPPO_CHECK_ARG(ctx, x_g && act_g && scal_g && adv_part_g && total >= 1 && M >= 1);
  const int64_t n_mb = cdiv64(total, M);
  CATPPO_CHECK_ARG(ctx, n_mb <= 65535);
  hipLaunchKernelGGL(ppo_gather_kernel, dim3((unsigned)cdiv64(M, kGatherRows), (unsigned)n_mb), dim3(256), 0,
                     static_cast<hipStream_t>(stream), b_obs, b_actions, b_logprobs, b_advantages, b_returns_n,
                     b_values_n, inds, total, M, L.obs_pad, shape->act_dim, x_g, act_g, scal_g, adv_part_g,
                     (const catppo_iter_state*)nullptr, 0, 0, (int64_t*)nullptr);
  CATPPO_CHECK_LAUNCH(ctx);
  return CATPPO_OK;
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
  if (fused_head) {
    // hidden layers below the last: ONE row-resident launch (fwd_rows.h) when they are all 256 wide and the minibatch has
    // enough 64-row tiles, else the layer-wise GEMM launches.  CATPPO_ROWS_FWD=0 keeps the latter (A/B).
    static const int rows_fwd_env = env_int("CATPPO_ROWS_FWD", 1);
    static const int rows_fwd_min = env_int("CATPPO_ROWS_FWD_MIN_ROWS", 8192);
    FusedFwdArgs ra{};
    size_t rlds = 0;
    if (rows_fwd_env && M >= rows_fwd_min && M <= (1 << 20) && nl - 1 <= 3 &&
        rows_fwd_plan(shape, L, nl - 1, 64, &ra, &rlds)) {
      ra.x = w.xmb, ra.params = params, ra.M = M, ra.net0 = 0, ra.do_head = 0;
      for (int net = 0; net < 2; ++net)
        for (int l = 0; l < nl - 1; ++l) ra.Hout[net][l] = w.H[net][l];
      rows_fwd_launch_train(ra, rlds, M, ctx->n_cu, s);
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
      }
      if (!done) forward_hidden(shape, L, params, w.xmb, M, w, 0, 2, s, nl - 1);
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
    }
    HeadArgs g{};
    g.dZc = w.dZ[0][nl - 1], g.dZa = w.dZ[1][nl - 1];
    g.W4c = params + L.off_w[0][nl], g.b4c = params + L.off_b[0][nl];
    g.W4a = params + L.off_w[1][nl], g.b4a = params + L.off_b[1][nl];
    g.logstd = params + L.off_logstd;
    g.act = w.act, g.oldlogp = w.scal, g.adv = w.scal + M, g.ret_n = w.scal + 2 * M, g.val_n = w.scal + 3 * M;
    g.adv_part = w.adv_part, g.n_adv_part = nbg;
    g.adv_stats = hp->adv_stats_external ? adv_stats : nullptr;
    g.vrms_mean = vrms_mean, g.vrms_var = vrms_var;
    g.part_w = w.head_w, g.part_s = w.head_s;
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
    const int pr = shape->mfma_bf16;
    if (HL == 256) {
      if (pr == 0) launch_fh(integral_constant<int, 256>{}, integral_constant<int, 0>{});
      else if (pr == 1) launch_fh(integral_constant<int, 256>{}, integral_constant<int, 1>{});
      else launch_fh(integral_constant<int, 256>{}, integral_constant<int, 2>{});
    } else {
      if (pr == 0) launch_fh(integral_constant<int, 128>{}, integral_constant<int, 0>{});
      else if (pr == 1) launch_fh(integral_constant<int, 128>{}, integral_constant<int, 1>{});
      else launch_fh(integral_constant<int, 128>{}, integral_constant<int, 2>{});
    }
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
    g.part_w = w.head_w, g.part_s = w.head_s;
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
  const bool overlap = !fork && ctx->grad_overlap && ctx->comm != nullptr;
  if (ne != nullptr && (fork || overlap))
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
  for (int l = nl - 1; l >= 0; --l) {
    const int out = shape->hidden[l], in = L.in_dim[l];
    // dZ_l is complete on the main stream here: fork
    if (fork) {
      CATPPO_HIP_OK(hipEventRecord(ctx->ev_fork[l], s));
      CATPPO_HIP_OK(hipStreamWaitEvent(side, ctx->ev_fork[l], 0));
    }
    // weight gradient: dW[out,in] = dZ^T . Xin      (contraction over the M rows)
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
      splits = 512 / (t64 > 0 ? t64 : 1);
      const int max128 = (int)(M / 128);
      if (splits > max128) splits = max128;
      if (splits > split_cap(out, in)) splits = split_cap(out, in);
      if (splits < 1) splits = 1;
      per = (int)cdiv64(M, splits);
      per = (per + gemm::BK - 1) / gemm::BK * gemm::BK;
    }
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
    const bool dw_with_fold = !pair && l == 0 && dw0_fold && !fork && !overlap && bf16 == 0 && segs.n > 0 &&
                              !(pw.I >= 128 && pw.J >= 128 && pw.kc_per_split >= 256);      // launch_gemm_auto's 128x128 rule
    if (dw_with_fold) {
      const int t64 = tiles_of<64, 64>(pw), n_gemm = t64 * pw.nets * pw.splits;
      constexpr size_t lds = gemm::smem_bytes<64, 64, false, false>();
      static_assert(lds >= 4096, "the fold workgroups use 1024 floats of the same allocation");
      hipLaunchKernelGGL(dw_fold_kernel, dim3((unsigned)(n_gemm + kFoldX * segs.n)), dim3(256), lds, s, pw, segs, t64, n_gemm,
                         hp->ent_coef, hp->vf_coef, ne ? ne->part + ne->n_slots : (double*)nullptr);
      CATPPO_CHECK_LAUNCH(ctx);
      if (ne) ne->n_slots += kFoldX * segs.n;
      segs.n = 0;        // folded; what is added below (this layer's own partials) goes to the final fold launch
    } else if (!pair) {
      launch_gemm_auto<false, false, gemm::EPI_PARTIAL>(pw, side, bf16);
      CATPPO_CHECK_LAUNCH(ctx);
    }
    for (int net = 0; net < 2; ++net) {
      add_seg(w.wpart[l] + (int64_t)net * out * in, grad + L.off_w[net][l], (int64_t)out * in,
              2 * (int64_t)out * in, splits, 0, 1.0f);
      add_seg(w.bpart[l] + (int64_t)net * splits * out, grad + L.off_b[net][l], out, out, splits, 0, 1.0f);
    }
    if (l == nl - 1) {
      // head partials + diagnostics ride along with the reduction launch
      const int NS = head_scalars(A);
      // fused head: partial rows [0, nbh) come from the actor workgroups, [nbh, 2 nbh) from the critic's
      const int64_t wrow = (int64_t)(A + 1) * HL;
      const float* cw = w.head_w + (fused_head ? (int64_t)nbh * wrow : 0);
      const float* cs = w.head_s + (fused_head ? (int64_t)nbh * NS : 0);
      add_seg(w.head_w, grad + L.off_w[1][nl], (int64_t)A * HL, wrow, nbh, 0, 1.0f);
      add_seg(cw + (int64_t)A * HL, grad + L.off_w[0][nl], HL, wrow, nbh, 0, 1.0f);
      add_seg(w.head_s, grad + L.off_b[1][nl], A, NS, nbh, 0, 1.0f);
      add_seg(cs + A, grad + L.off_b[0][nl], 1, NS, nbh, 0, 1.0f);
      add_seg(w.head_s + A + 1, grad + L.off_logstd, A, NS, nbh, 0, 1.0f);
      add_seg(w.head_s + 2 * A + 1, diag, kHeadDiag, NS, fused_head ? 2 * nbh : nbh, 1, hp->inv_global_batch);
    }
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
      if (pair)
        launch_dw_dx_pair(pw, px, s, bf16);
      else
        launch_gemm_auto<true, false, gemm::EPI_MUL_DELU>(px, s, bf16);
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
  CATPPO_CHECK_LAUNCH(ctx);
  if (ne) ne->n_slots += 256 * segs.n;
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

// ---- clip + Adam with the learning rate and the step count in device memory (catppo_iter_state) ----------------
namespace {
__global__ __launch_bounds__(256) void sqnorm_partial_step_kernel(const float* __restrict__ g, int64_t n,
                                                                  double* __restrict__ part,
                                                                  catppo_iter_state* __restrict__ st, double beta1,
                                                                  double beta2) {
  __shared__ double sm[4];
  // one lane of the launch advances the step count and prepares Adam's bias corrections for the NEXT launch
  // (clip_adam_dev_kernel) while everybody else is waiting for their gradient loads: two double-precision pow() that
  // used to sit at the head of every workgroup of the Adam launch.
  // torch.optim.Adam: bias_correction = 1 - beta ** step (Python doubles), step_size = lr / bias_correction1
  if (blockIdx.x == 0 && threadIdx.x == 64) adam_advance_step(st, beta1, beta2);
  double a = sqnorm_of_thread(g, n);
  a = wave_sum_d(a);
  if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6] = a;
  __syncthreads();
  if (threadIdx.x == 0) part[blockIdx.x] = (sm[0] + sm[1]) + (sm[2] + sm[3]);
}

__global__ __launch_bounds__(256) void clip_adam_dev_kernel(float* __restrict__ p, float* __restrict__ g,
                                                            float* __restrict__ m, float* __restrict__ v, int64_t n,
                                                            const double* __restrict__ norm_part, int n_part,
                                                            float max_norm, double beta1, double beta2, float eps,
                                                            const catppo_iter_state* __restrict__ st) {
  __shared__ float s_coef;
  if (n_part <= kNormBlocks) {
    if (threadIdx.x < 64) {
      double a = 0.0;
      for (int b = threadIdx.x; b < n_part; b += 64) a += norm_part[b];
      a = wave_sum_d(a);
      if (threadIdx.x == 0) {
        const float total = (float)sqrt(a);
        const float c = max_norm / (total + 1e-6f);
        s_coef = c > 1.0f ? 1.0f : c;
      }
    }
  } else {
    // the slots of the fold launches (one per fold workgroup, a few thousand): all four waves, eight requests in
    // flight per thread, fixed order
    __shared__ double s_w[4];
    double a0 = 0.0, a1 = 0.0, a2 = 0.0, a3 = 0.0;
    int b = threadIdx.x;
    for (; b + 7 * 256 < n_part; b += 8 * 256) {
      const double x0 = norm_part[b], x1 = norm_part[b + 256], x2 = norm_part[b + 512], x3 = norm_part[b + 768];
      const double x4 = norm_part[b + 1024], x5 = norm_part[b + 1280], x6 = norm_part[b + 1536], x7 = norm_part[b + 1792];
      a0 += x0, a1 += x1, a2 += x2, a3 += x3, a0 += x4, a1 += x5, a2 += x6, a3 += x7;
    }
    for (; b < n_part; b += 256) a0 += norm_part[b];
    double a = wave_sum_d((a0 + a1) + (a2 + a3));
    if ((threadIdx.x & 63) == 0) s_w[threadIdx.x >> 6] = a;
    __syncthreads();
    if (threadIdx.x == 0) {
      const float total = (float)sqrt((s_w[0] + s_w[1]) + (s_w[2] + s_w[3]));
      const float c = max_norm / (total + 1e-6f);
      s_coef = c > 1.0f ? 1.0f : c;
    }
  }
  const float step_size = st->adam_step_size, bc2_sqrt = st->adam_bc2_sqrt;   // the launch in front of this one wrote them
  __syncthreads();
  const AdamCoef c{s_coef, (float)(1.0 - beta1), (float)beta2, (float)(1.0 - beta2), eps, step_size, bc2_sqrt};
  adam_all(p, g, m, v, n, c);
}
}  // namespace

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
  CATPPO_CHECK_LAUNCH(ctx);
  return CATPPO_OK;
}

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
