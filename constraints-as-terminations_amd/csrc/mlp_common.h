// Shared pieces of the actor-critic MLP translation unit (mlp.hip): flat parameter layout, workspace carving, GEMM launch
// helpers (tile choice), wave-level helpers.  Included INSIDE mlp.hip's anonymous namespace - one translation unit, split by
// role for reading (round 5): mlp_common.h | mlp_forward.h | mlp_loss.h | mlp_backward.h | mlp_optim.h.
#pragma once


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
      // every weight matrix starts on a 128-byte line (round 6): with 16-byte alignment the rows of the hidden layers -
      // 1 or 2 KB each - all began 48 bytes into a line, so every coalesced 128-byte row segment of the row-resident kernels
      // straddled two lines (twice the tag look-ups, the second line re-fetched by the next slab's request)
      off = (off + 31) / 32 * 32;
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
  uint16_t* w16;           // [n_flat] bf16 copy of the flat parameters (hidden layers >= 1), bf16-stored mode ("act16", gemm_f32.h)
  uint16_t* w16t;          // [n_flat] the same matrices transposed ([in][out]): K-contiguous operand of the data gradient
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
  // upper bound of the head partial rows: head_loss blocks (16-row tiles at most), or step16's 16-row tiles x 2 networks
  const int64_t nbg = cdiv64(M, kGatherRows), nbh = M <= 8192 ? 2 * cdiv64(M, 16) : cdiv64(M, 16);
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
  if (s->mfma_bf16 == 1) {        // (the rollout forward uses the as-stored copy only; one layout keeps the carving simple)
    w->w16 = (uint16_t*)take(sizeof(uint16_t) * L.n_flat);
    w->w16t = (uint16_t*)take(sizeof(uint16_t) * L.n_flat);
  }
  w->bytes = used;
  return ok;
}

// ------------------------------------------------------------------------------- GEMM launch
constexpr int kSmallRows = 4096;    // see launch_dw_dx_pair

static int env_int(const char* name, int dflt) {
  const char* e = getenv(name);
  return e ? atoi(e) : dflt;
}
// Params::xcd_legacy = 1 is the round-2 workgroup -> tile order (gemm::xcd_tile_of); its A/B switch went with round 6's prune
static int xcd_legacy() { return 0; }

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
void launch_gemm(const Params& p, hipStream_t s, int prec) {
  dim3 grid(((p.J + BN - 1) / BN) * ((p.I + BM - 1) / BM), 1, p.nets * p.splits);   // 1-D tile index, see kernel
  const size_t lds = gemm::smem_bytes<BM, BN, A_KC, B_KC>();
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
  if (use_big) {
    launch_gemm<128, 128, A_KC, B_KC, EPI>(p, s, prec);
    return;
  }
  if constexpr (EPI != gemm::EPI_PARTIAL) {
    constexpr int small_bk = 64;      // see kSmallRows (32-wide slabs measured in between, round 2)
    if (prec == 0 && p.I <= kSmallRows && p.Kc % small_bk == 0 && p.Kc >= 2 * small_bk) {
      launch_gemm_bk<64, 64, A_KC, B_KC, EPI, 64>(p, s);
      return;
    }
  }
  launch_gemm<64, 64, A_KC, B_KC, EPI>(p, s, prec);
}

template <int BM, int BN>
constexpr int tiles_of(const Params& p) { return ((p.J + BN - 1) / BN) * ((p.I + BM - 1) / BM); }

// one operand-precision mode, explicitly (the bf16-stored modes 3 / 4 of gemm_f32.h)
template <int BM, int BN, bool A_KC, bool B_KC, int EPI, int PREC>
void launch_gemm_prec(const Params& p, hipStream_t s) {
  dim3 grid(tiles_of<BM, BN>(p), 1, p.nets * p.splits);
  constexpr size_t lds = gemm::smem_bytes<BM, BN, A_KC, B_KC>();
  gemm::gemm_f32_kernel<BM, BN, A_KC, B_KC, EPI, gemm::BK, PREC><<<grid, dim3(256), lds, s>>>(p);
}

// bf16-stored mode: weight gradient (both operands bf16-stored, IILoop16) + data gradient against the TRANSPOSED bf16 weight
// copy (K-contiguous x K-contiguous) in one launch; tile rules of launch_dw_dx_pair
template <int BM0, int BN0, int BM1, int BN1>
void launch_pair_tiles16(const Params& pw, const Params& px, hipStream_t s) {
  const int t0 = tiles_of<BM0, BN0>(pw), n0 = t0 * pw.nets * pw.splits;
  const int t1 = tiles_of<BM1, BN1>(px), n1 = t1 * px.nets * px.splits;
  constexpr size_t lds0 = gemm::smem_bytes<BM0, BN0, false, false>();
  constexpr size_t lds1 = gemm::smem_bytes<BM1, BN1, true, true>();
  constexpr size_t lds = lds0 > lds1 ? lds0 : lds1;
  gemm::gemm_pair_kernel<BM0, BN0, false, false, gemm::EPI_PARTIAL, BM1, BN1, true, true, gemm::EPI_MUL_DELU, 3>
      <<<dim3(n0 + n1), dim3(256), lds, s>>>(pw, px, t0, n0, t1);
}
void launch_dw_dx_pair16(const Params& pw, const Params& px, hipStream_t s, int n_cu) {
  const bool underfilled = tiles_of<128, 128>(pw) * pw.nets * pw.splits < n_cu;
  const bool big = pw.I >= 128 && pw.J >= 128 && pw.kc_per_split >= 256 && !underfilled;
  const bool wide = px.J >= 128;
  if (big && wide) launch_pair_tiles16<128, 128, 64, 128>(pw, px, s);
  else if (big) launch_pair_tiles16<128, 128, 64, 64>(pw, px, s);
  else if (wide) launch_pair_tiles16<64, 64, 64, 128>(pw, px, s);
  else launch_pair_tiles16<64, 64, 64, 64>(pw, px, s);
}

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
// switches that went with round 6's prune).
void launch_dw_dx_pair(const Params& pw, const Params& px, hipStream_t s, int prec, int n_cu = 256) {
  constexpr int small_tile = 1;
  // Round 5: a weight gradient whose 128x128 tiling yields fewer long workgroups than there are CUs (a 128-wide layer - the
  // reference's last hidden layer: 2 tiles x 2 networks x 32 splits = 128 on 256 CUs) takes 64x64 tiles instead (512
  // shorter workgroups, same splits, same contraction order per element: bit-identical): the reference's layer-2 pair
  // 60.1 -> 50.6 us, update phase 11.70 -> 11.51 ms (profiles/r5_ab_dw_fill.txt).  CATPPO_DW_FILL=0: the 128x128 tiling.
  static const int dw_fill = env_int("CATPPO_DW_FILL", 1);
  const bool underfilled = dw_fill && tiles_of<128, 128>(pw) * pw.nets * pw.splits < n_cu;      // (the device's CU count)
  const bool big = pw.I >= 128 && pw.J >= 128 && pw.kc_per_split >= 256 &&   // launch_gemm_auto's rule for EPI_PARTIAL
                   !(px.I <= kSmallRows && small_tile) && !underfilled;
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

// 8 consecutive values as bf16 (RNE), one 16-byte write-through store (bf16-stored activations, gemm_f32.h "act16")
__device__ __forceinline__ void store8_bf16_wt(uint16_t* p, const float (&v)[8]) {
  gemm::u32x4 pk;
#pragma unroll
  for (int j = 0; j < 4; ++j) pk[j] = gemm::pk_bf16(v[2 * j], v[2 * j + 1]);
  asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 2" ::"v"(p), "v"(pk) : "memory");
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
