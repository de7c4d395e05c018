// Small-minibatch optimiser step (round 6): forward of every layer, heads, PPO loss, head backward AND the data-gradient
// chain of the hidden layers for 16 minibatch rows of ONE network in one workgroup - one launch where the layer-wise
// path needs seven (three forward GEMMs, head_loss_kernel, two data-gradient GEMMs inside the pair launches, the first
// layer's own weight-gradient launch): cleanrl/ppo.py:300-352 for the 2048-row minibatches an 8-way env shard of
// BASELINE configs[2] runs (solo12/agents/clean_rl_ppo_cfg.py:20 sharded), for cfg1 and for every minibatch that leaves
// 64-row tiles with fewer workgroups than the chip has CUs.
//
// Why 16 rows: at 2048 rows x 2 networks a 16-row tile gives 256 workgroups - one per CU - where the 32 / 64-row
// row-resident kernels give 128 / 64.  The contraction runs on v_mfma_f32_16x16x4_f32: the probe of this round
// (tools/mfma16_probe.hip, profiles/r6_mfma16_probe.txt) shows the instruction is ONE fp32 FMA chain per output element
// over its four k in lane-group order (k index = lane / 16), exactly like v_mfma_f32_32x32x2_f32 over its two (lane / 32).
// A lane (c16 = lane % 16, g4 = lane / 16) fetches four consecutive k (k = 16 j + 4 g4 + e) of its operand row with one
// 16-byte access; two v_permlane32_swap per float4 turn them into the four operands of the unit in gemm_body's
// contraction order - per 8-k block 0,4,1,5,2,6,3,7 - so every hidden activation and every dZ is BIT-IDENTICAL to the
// layer-wise launches (tests/test_gpu_r6.py).
//
// No weight ring in LDS.  A 16 x 16 accumulator tile is 4 VGPRs, so the registers the 32-row kernels spend on
// accumulators hold the weight stream instead: every lane requests its operand rows straight from global memory (L2
// hits: all CUs walk the same 1.5 MB), STEP16_DEPTH_KB of weights per wave ahead of their use.
//   forward (W[n][k], k contiguous):  wave w owns columns [w CW, w CW + CW), CW = N / 8 = 16 T, interleaved over its T
//       tiles - column of (tile t, c16) = w CW + c16 T + t - so a lane's T outputs of a row are T consecutive floats
//       (one LDS write, one global store) and its T weight rows are consecutive rows of W;
//   data gradient (W[k][n], n contiguous): the same interleaving makes the lane's T operands of one k ONE T-float load
//       (four k rows x 16 lanes x 4 T bytes per instruction: full lines for T >= 2).
// LDS holds the observation tile and the activation tile of every layer (16 x (Dp + N0 + N1 + N2 + 32) floats: 62 KB for
// 512 / 256 / 128); the backward overwrites each activation tile with its dZ in place.  Activations and dZ leave for the
// weight-gradient launch straight from the accumulator registers (write-through, full lines).
//
// Order per workgroup:  X tile | L0 L1 L2 forward | head outputs (step A) | row math (one wave) | head weight gradient
// (C) + dZ of the last layer (B) | dX through W2 -> dZ1 | dX through W1 -> dZ0 | per-tile scalars.  Steps A / C / B and the
// row math follow fwd_head_kernel (mlp_loss.h) operation for operation: the forward diagnostics and every dZ are
// bit-identical to that launch; the head weight / bias partials cover 16 rows instead of 64 (another fold order).
#pragma once

namespace step16 {

using f4v = __attribute__((ext_vector_type(4))) float;
using f2v = __attribute__((ext_vector_type(2))) float;
using u32x4 = __attribute__((ext_vector_type(4))) unsigned int;
using u32x2 = __attribute__((ext_vector_type(2))) unsigned int;

constexpr int kThreads = 512;      // eight waves, two per SIMD
constexpr int kR = 16;             // rows per workgroup
constexpr int kPad = 8;            // tile row stride = width + 8 floats: the b128 lane groups of gfx950 hit 16 distinct 4-bank slots
#ifndef STEP16_DEPTH_KB
#define STEP16_DEPTH_KB 16
#endif
constexpr int kDepthKB = STEP16_DEPTH_KB;   // weight bytes a wave keeps in flight (KB; 64 lanes x 4 B: 4 VGPRs per KB)
constexpr int depth_of(int elem_kb, int n_elem) {   // pipeline elements of elem_kb KB each
  int d = kDepthKB / elem_kb;
  d = d < 2 ? 2 : d;
  return d < n_elem ? d : n_elem;
}

struct Args {
  const float* x;                  // [M, Dp] gathered observations
  const float* params;
  int64_t M, n_flat;
  int64_t off_w[2][CATPPO_MAX_HIDDEN + 1], off_b[2][CATPPO_MAX_HIDDEN + 1];
  float* H[2][CATPPO_MAX_HIDDEN];  // [net][layer] activations out (layers below the last)
  float* dZ[2][CATPPO_MAX_HIDDEN]; // [net][layer] pre-activation gradients out
  HeadArgs g;
};

template <int T> struct VecOf;
template <> struct VecOf<1> { using type = float; };
template <> struct VecOf<2> { using type = f2v; };
template <> struct VecOf<4> { using type = f4v; };

template <int T>
__device__ __forceinline__ float vget(const typename VecOf<T>::type& v, int t) {
  if constexpr (T == 1) return v;
  else return v[t];
}
template <int T>
__device__ __forceinline__ void vset(typename VecOf<T>::type& v, int t, float x) {
  if constexpr (T == 1) v = x;
  else v[t] = x;
}

// load T consecutive floats: descriptor (SGPRs) + per-lane byte offset (ONE VGPR per layer) + wave-uniform byte offset (an
// SGPR add per request).  With plain pointers every request of the unrolled contractions got its own 64-bit VGPR address
// (row offsets of 2 KB and more do not fit the instruction's immediate): ~40 live pointer pairs, 256 VGPRs and spills.
template <int T>
__device__ __forceinline__ typename VecOf<T>::type bload(__amdgpu_buffer_rsrc_t rs, uint32_t voff, uint32_t soff) {
  if constexpr (T == 4) return __builtin_bit_cast(f4v, __builtin_amdgcn_raw_buffer_load_b128(rs, voff, soff, 0));
  else if constexpr (T == 2) return __builtin_bit_cast(f2v, __builtin_amdgcn_raw_buffer_load_b64(rs, voff, soff, 0));
  else return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs, voff, soff, 0));
}

// store T consecutive floats through a buffer descriptor (a row past M: dropped), write-through
template <int T>
__device__ __forceinline__ void bstore(const typename VecOf<T>::type& v, __amdgpu_buffer_rsrc_t rs, uint32_t off) {
  if constexpr (T == 4) __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), rs, off, 0, 16);
  else if constexpr (T == 2) __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2, v), rs, off, 0, 16);
  else __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned int, v), rs, off, 0, 16);
}

// OPERAND-ORDER TILES.  Operand i of lane group g4 in a 16-k unit must carry k = kperm(i, g4) = 8 (i >> 1) + 2 (i & 1) +
// (g4 >> 1) + 4 (g4 & 1) (gemm_body's contraction order).  A tile / slot row that stores the 16 k of a unit in the order
//     position:  0 1 2 3 | 4 5 6  7  | 8 9 10 11 | 12 13 14 15
//     k       :  0 2 8 10| 4 6 12 14 | 1 3 9  11 | 5  7  13 15        pos(k) = b0 << 3 | b2 << 2 | b3 << 1 | b1
// hands lane group g4 its four operands with ONE ds_read_b128 of chunk g4 - no lane swaps.  (With k stored in natural order
// every fragment cost two v_permlane32_swap, ~13 cycles of the SIMD each: 12 per 32-k stage and wave at T = 2, 2.3 us of
// the 512-k layer - profiles/r6_step16_timeline_noswap.txt.)  Writers place values with pos(): four consecutive k
// (4 m .. 4 m + 3) are two adjacent pairs - (k, k + 2) at pos(k), (k + 1, k + 3) at pos(k) + 8 - i.e. two 8-byte stores.
__host__ __device__ constexpr int pos16(int k) { return ((k & 1) << 3) | (((k >> 2) & 1) << 2) | (((k >> 3) & 1) << 1) | ((k >> 1) & 1); }
// position (floats) of natural column c inside a row
__host__ __device__ constexpr int pcol(int c) { return (c & ~15) | pos16(c & 15); }

// a lane's T consecutive natural columns [col, col + T) of one tile row (col % T == 0): store / load, operand-order (PERM)
// or natural layout
template <int T, bool PERM>
__device__ __forceinline__ void tile_put(float* __restrict__ row, int col, const typename VecOf<T>::type& v) {
  if constexpr (!PERM) {
    *reinterpret_cast<typename VecOf<T>::type*>(row + col) = v;
  } else if constexpr (T == 4) {
    const int p = pcol(col);
    *reinterpret_cast<f2v*>(row + p) = f2v{v[0], v[2]};
    *reinterpret_cast<f2v*>(row + p + 8) = f2v{v[1], v[3]};
  } else if constexpr (T == 2) {
    const int p = pcol(col);
    row[p] = v[0];
    row[p + 8] = v[1];
  } else {
    row[pcol(col)] = v;
  }
}
template <int T, bool PERM>
__device__ __forceinline__ typename VecOf<T>::type tile_get(const float* __restrict__ row, int col) {
  using V = typename VecOf<T>::type;
  if constexpr (!PERM) {
    return *reinterpret_cast<const V*>(row + col);
  } else if constexpr (T == 4) {
    const int p = pcol(col);
    const f2v a = *reinterpret_cast<const f2v*>(row + p), b = *reinterpret_cast<const f2v*>(row + p + 8);
    return V{a[0], b[0], a[1], b[1]};
  } else if constexpr (T == 2) {
    const int p = pcol(col);
    return V{row[p], row[p + 8]};
  } else {
    return row[pcol(col)];
  }
}
// the four operands of a unit straight out of an operand-order row
__device__ __forceinline__ void ops_of(const f4v v, float (&o)[4]) { o[0] = v.x, o[1] = v.y, o[2] = v.z, o[3] = v.w; }

// The four MFMA operands of one 16-k unit out of a lane's float4 (k = 4 g4 + e): operand i of lane group g4 carries
//   i = 0: k {0,4,1,5}[g4]   i = 1: {2,6,3,7}   i = 2: {8,12,9,13}   i = 3: {10,14,11,15}
// (v_permlane32_swap a, b: a <- [a.lo | b.lo], b <- [a.hi | b.hi] over the two 32-lane halves)
__device__ __forceinline__ void prep(const f4v v, float (&o)[4]) {
#ifdef STEP16_EXP_NOSWAP        // timing experiment (tools/step16_timeline.py): wrong results, never in the product build
  o[0] = v.x, o[1] = v.y, o[2] = v.z, o[3] = v.w;
  return;
#endif
  const auto s0 = __builtin_amdgcn_permlane32_swap(__float_as_uint(v.x), __float_as_uint(v.y), false, false);
  const auto s1 = __builtin_amdgcn_permlane32_swap(__float_as_uint(v.z), __float_as_uint(v.w), false, false);
  o[0] = __uint_as_float(s0[0]), o[2] = __uint_as_float(s0[1]);
  o[1] = __uint_as_float(s1[0]), o[3] = __uint_as_float(s1[1]);
}
// first k row (inside a unit) lane group g4 supplies for operand i of an I-contiguous operand: same order as prep()
__device__ __forceinline__ int krow_of(int i, int g4) { return 8 * (i >> 1) + 2 * (i & 1) + (g4 >> 1) + 4 * (g4 & 1); }

#ifdef STEP16_TL   // tools/step16_timeline.py: lane 0 of every wave stamps the 100 MHz wall clock at the phase boundaries
__device__ unsigned long long* g_s16tl;    // [2 networks][1024 tiles][8 waves][16 stamps]
#define S16_TL(i) do { if ((threadIdx.x & 63) == 0 && g_s16tl && blockIdx.x < 1024) g_s16tl[((blockIdx.y * 1024 + blockIdx.x) * 8 + (threadIdx.x >> 6)) * 16 + (i)] = wall_clock64(); } while (0)
#else
#define S16_TL(i) do { } while (0)
#endif

// Two waves share a SIMD and run the same instruction stream from the same barrier.  Left alone they stay in phase: both
// in their LDS block (eight waves' ds_write_b128 / ds_read_b128 serialise on the CU's LDS: ~600 cycles per 32-k stage, the
// matrix pipe idle), then both in their MFMA block (2 x 512 cycles) - 1600 cycles per stage where the pipe needs 1024
// (profiles/r6_step16_timeline_waves.txt, tools/swap_probe.hip).  Raising the priority inside the MFMA block does not
// break the symmetry (both waves raise it at the same time: measured no change).  A PERMANENT difference does: waves
// 0-3 (one per SIMD: a workgroup's waves are dealt to the SIMDs cyclically) run at priority 2, waves 4-7 at 0 - the
// first wave of a SIMD runs its stages at its own pace and the second one's MFMA blocks fill the gaps its LDS / lane-swap
// blocks leave.  STEP16_PRIO_MODE: 0 off, 1 raised inside MFMA blocks, 2 (default) the permanent split.
#ifndef STEP16_PRIO_MODE
#define STEP16_PRIO_MODE 2
#endif
#if STEP16_PRIO_MODE == 1
#define STEP16_PRIO(p_) __builtin_amdgcn_s_setprio(p_)
#else
#define STEP16_PRIO(p_) do { } while (0)
#endif
#define STEP16_MFMA(a_, b_, c_) c_ = __builtin_amdgcn_mfma_f32_16x16x4f32(a_, b_, c_, 0, 0, 0)

// Weight streams.  Every layer object owns a register ring of DE pipeline elements (Fwd16 / Bwd: one 16-k unit; Fwd32: one
// 32-k stage) requested DE elements ahead of their use.  run() keeps the stream going ACROSS the layer boundary: once
// its own elements are all requested, every freed slot-time requests an element of the NEXT layer's ring (the rings are
// separate register arrays; what is live at any time is ~one ring), so the memory pipe neither idles through a layer's
// tail nor takes the next layer's first DE elements as one burst in front of a barrier.
struct NoNext {
  static constexpr int NE = 0, DE = 0;
  __device__ __forceinline__ void request(int, int) {}
};
// what a layer does behind the MFMAs of its element j
template <class Self, class Next>
__device__ __forceinline__ void stream_next(Self& me, Next& nx, int j) {
  if (j + Self::DE < Self::NE) {
    me.request(j + Self::DE, j % Self::DE);
  } else {
    const int idx = j + Self::DE - Self::NE;
    if (idx < Next::DE) nx.request(idx, idx);
  }
  __builtin_amdgcn_sched_barrier(0);        // end of the element's scheduling region
}
template <class Self, class Next>
__device__ __forceinline__ void stream_rest(Next& nx) {      // behind the last element: what the tail could not request
#pragma unroll
  for (int idx = Self::DE < Self::NE ? Self::DE : Self::NE; idx < Next::DE; ++idx) nx.request(idx, idx);
}

// ---- forward layer, 16-k units (any K % 16 == 0; the first layer): acc[t] = tile[:, :K] . W[w CW + c16 T + t, :K]^T
template <int T, int K>
struct Fwd16 {
  static constexpr int NE = K / 16;
  static constexpr int DE = depth_of(T, NE);
  static_assert(K % 16 == 0 && NE >= 1, "contraction in 16-k units");
  f4v ring[DE][T];
  __amdgpu_buffer_rsrc_t rs;
  uint32_t vo, so;                 // byte offsets into the flat parameter buffer: per lane | of this wave's first weight row
  __device__ __forceinline__ void init(__amdgpu_buffer_rsrc_t prs, int64_t w_off, int wave, int lane) {
    rs = prs;
    so = (uint32_t)(w_off + (int64_t)wave * 16 * T * K) * 4u;
    vo = (uint32_t)(((lane & 15) * T) * K + 4 * (lane >> 4)) * 4u;
  }
  __device__ __forceinline__ void request(int j, int slot) {
#pragma unroll
    for (int t = 0; t < T; ++t) ring[slot][t] = bload<4>(rs, vo, so + (uint32_t)(t * K + 16 * j) * 4u);
  }
  __device__ __forceinline__ void prefetch() {
#pragma unroll
    for (int j = 0; j < DE; ++j) request(j, j);
  }
  // Every element is one scheduling region (sched_barrier at its end): the MFMAs of element j, the operand preparation of
  // element j + 1 (LDS read, lane swaps: they issue behind the MFMAs) and one request.  Without the region boundaries the
  // compiler hoists the LDS reads and swaps of ALL elements of the unrolled contraction to its top (128+ VGPRs of
  // fragments, spills).
  template <class Next>
  __device__ __forceinline__ void run(const float* __restrict__ tile, const int ld, f4v (&acc)[T], int lane, Next& nx) {
    const float* ap = tile + (lane & 15) * ld + 4 * (lane >> 4);
    float a[2][4], b[2][T][4];
    ops_of(*reinterpret_cast<const f4v*>(ap), a[0]);          // the activation tile is in operand order
#pragma unroll
    for (int t = 0; t < T; ++t) prep(ring[0][t], b[0][t]);   // the weights come straight from memory: natural order
#pragma unroll
    for (int t = 0; t < T; ++t) acc[t] = f4v{0.f, 0.f, 0.f, 0.f};
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int j = 0; j < NE; ++j) {
      const int c = j & 1, n = c ^ 1;
      f4v af;
      if (j + 1 < NE) af = *reinterpret_cast<const f4v*>(ap + 16 * (j + 1));
      __builtin_amdgcn_sched_barrier(0);
      STEP16_PRIO(1);
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int t = 0; t < T; ++t) STEP16_MFMA(a[c][i], b[c][t][i], acc[t]);
      STEP16_PRIO(0);
      __builtin_amdgcn_sched_barrier(0);
      if (j + 1 < NE) {
        ops_of(af, a[n]);
#pragma unroll
        for (int t = 0; t < T; ++t) prep(ring[(j + 1) % DE][t], b[n][t]);
      }
      stream_next(*this, nx, j);
    }
    stream_rest<Fwd16, Next>(nx);
  }
};

// ---- forward layer, 32-k stages through a wave-private LDS slot (K % 32 == 0, T = 1 | 2).
// Fwd16's requests follow the MFMA operand layout: the 16 lanes of a quarter wave address 16 DIFFERENT weight rows, and
// the texture addresser looks up one 128-byte line per distinct row and quarter wave - 64 look-ups per 1 KB instruction
// where a coalesced one needs 8.  Measured (profiles/r6_step16_timeline.txt): the 512-k layer took 13-16 us against
// 8.2 us for the same bytes and MFMA count in the data gradient (whose requests are 256 contiguous bytes per quarter wave);
// ~1 look-up per clock and CU fits every layer's time.  So the weights are REQUESTED coalesced - lane -> (row lane / 8 +
// 8 i, 16-byte chunk lane % 8): eight lanes per 128-byte row segment, as fwd_rows.h - parked in registers DE stages
// ahead, and pass through ONE LDS slot per wave ([16 T rows][32 k + pad]) one stage ahead of their use: ds_write_b128 in
// the request layout, ds_read_b128 in the operand layout (row c16 T + t, k = 16 u + 4 g4 ..), prep() as for Fwd16.
// The slot belongs to one wave (LDS operations of a wave execute in order): no barrier.  Row stride S: T S = 8 mod 64
// floats, the operand reads of a b128 lane group then hit 16 distinct 4-bank slots (as the activation tiles).
template <int T, int K>
struct FwdL {
  static_assert(T == 1 || T == 2, "row stride rule");
  static constexpr int NE = K / 32;
  static constexpr int DE = depth_of(2 * T, NE);
  static constexpr int S = T == 2 ? 36 : 40;
  static constexpr int NLD = 2 * T;                  // requests per stage: 16 T rows / 8 rows per request
  static constexpr int kSlotFloats = 16 * T * S;
  static_assert(K % 32 == 0 && NE >= 1, "contraction in 32-k stages");
  f4v ring[DE][NLD];
  __amdgpu_buffer_rsrc_t rs;
  uint32_t vo, so;
  float* wr;                       // this lane's write position in the slot (row lane / 8, chunk lane % 8)
  const float* rd;                 // this lane's operand position (row c16 T, k = 4 g4)
  __device__ __forceinline__ void init(__amdgpu_buffer_rsrc_t prs, int64_t w_off, int wave, int lane, float* slot) {
    rs = prs;
    so = (uint32_t)(w_off + (int64_t)wave * 16 * T * K) * 4u;
    vo = (uint32_t)((lane >> 3) * K + 4 * (lane & 7)) * 4u;
    wr = slot + (lane >> 3) * S + 16 * ((lane & 7) >> 2) + pos16(4 * (lane & 3));   // operand order: k (4 q ..) -> two pairs
    rd = slot + (lane & 15) * T * S + 4 * (lane >> 4);
  }
  __device__ __forceinline__ void request(int j, int slot) {
#pragma unroll
    for (int i = 0; i < NLD; ++i) ring[slot][i] = bload<4>(rs, vo, so + (uint32_t)(8 * i * K + 32 * j) * 4u);
  }
  __device__ __forceinline__ void prefetch() {
#pragma unroll
    for (int j = 0; j < DE; ++j) request(j, j);
  }
  struct Raw {
    f4v a[2], b[2][T];
  };
  struct Ops {
    float a[2][4], b[2][T][4];
  };
  // stage j: registers -> slot -> operand fragments (+ the activation fragments of the stage)
  __device__ __forceinline__ void stage_in(Raw& r, const float* __restrict__ ap, int j) {
#ifdef STEP16_EXP_NOLDSW       // timing experiment: the slot is never written (the requests still have to arrive)
#pragma unroll
    for (int i = 0; i < NLD; ++i) asm volatile("" ::"v"(ring[j % DE][i]));
#else
#pragma unroll
    for (int i = 0; i < NLD; ++i) {
      const f4v v = ring[j % DE][i];
      *reinterpret_cast<f2v*>(wr + 8 * i * S) = f2v{v.x, v.z};
      *reinterpret_cast<f2v*>(wr + 8 * i * S + 8) = f2v{v.y, v.w};
    }
#endif
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      r.a[u] = *reinterpret_cast<const f4v*>(ap + 32 * j + 16 * u);
#pragma unroll
      for (int t = 0; t < T; ++t) r.b[u][t] = *reinterpret_cast<const f4v*>(rd + t * S + 16 * u);
    }
  }
  static __device__ __forceinline__ void make_ops(Ops& o, const Raw& r) {
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      ops_of(r.a[u], o.a[u]);
#pragma unroll
      for (int t = 0; t < T; ++t) ops_of(r.b[u][t], o.b[u][t]);
    }
  }
  template <class Next>
  __device__ __forceinline__ void run(const float* __restrict__ tile, const int ld, f4v (&acc)[T], int lane, Next& nx) {
    const float* ap = tile + (lane & 15) * ld + 4 * (lane >> 4);
    Ops ops[2];
    {
      Raw r;
      stage_in(r, ap, 0);
      make_ops(ops[0], r);
    }
#pragma unroll
    for (int t = 0; t < T; ++t) acc[t] = f4v{0.f, 0.f, 0.f, 0.f};
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int j = 0; j < NE; ++j) {                      // one scheduling region per stage, see Fwd16::run
      const int c = j & 1, n = c ^ 1;
      // ONE scheduling region per stage, its instructions interleaved by sched_group_barrier: behind every MFMA of stage j
      // (32 cycles of matrix pipe each) one LDS operation of stage j + 1 -
      //     4 T x (MFMA, ds_write_b64)    the staged weights into the slot, operand order
      //     2 T + 2 x (MFMA, ds_read_b128) the fragments back (LDS operations of a wave execute in order) = the operands
      // As separate blocks (LDS, MFMAs, swaps) the two waves of a SIMD stayed in phase - both stalled at the issue of their
      // ds_write_b128 (the VGPR -> LDS path is shared by two SIMDs: 13 cycles per store) with the matrix pipe idle, then
      // both in their MFMA block: 1600 cycles per stage pair where the pipe needs 1024 (r6_step16_timeline_waves.txt).
      Raw r;
      if (j + 1 < NE) stage_in(r, ap, j + 1);           // (every fragment of stage j was read in the previous region)
#pragma unroll
      for (int u = 0; u < 2; ++u)
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
          for (int t = 0; t < T; ++t) STEP16_MFMA(ops[c].a[u][i], ops[c].b[u][t][i], acc[t]);
      if (j + 1 < NE) make_ops(ops[n], r);
      if (j + 1 < NE) {
        constexpr int kMfma = 8 * T, kWr = 2 * NLD, kRd = 2 + 2 * T;
        static_assert(kWr + kRd <= kMfma + 2, "one LDS operation behind (nearly) every MFMA");
#pragma unroll
        for (int q = 0; q < kWr; ++q) {
          __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
          __builtin_amdgcn_sched_group_barrier(0x200, 1, 0);
        }
#pragma unroll
        for (int q = 0; q < kRd; ++q) {
          __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
          __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
        }
      }
      stream_next(*this, nx, j);
    }
    stream_rest<FwdL, Next>(nx);
  }
};

// ---- data gradient: acc[t] = dZtile[:, :K] . W[:K, w CW + c16 T + t]        (W row-major [K][N])
template <int T, int K, int N>
struct Bwd {
  using V = typename VecOf<T>::type;
  static constexpr int NE = K / 16;
  static constexpr int DE = depth_of(T, NE);
  V ring[DE][4];
  __amdgpu_buffer_rsrc_t rs;
  uint32_t vo, so;
  __device__ __forceinline__ void init(__amdgpu_buffer_rsrc_t prs, int64_t w_off, int wave, int lane) {
    rs = prs;
    so = (uint32_t)(w_off + wave * 16 * T) * 4u;
    vo = (uint32_t)(krow_of(0, lane >> 4) * N + (lane & 15) * T) * 4u;
  }
  __device__ __forceinline__ void request(int j, int slot) {
#pragma unroll
    for (int i = 0; i < 4; ++i)
      ring[slot][i] = bload<T>(rs, vo, so + (uint32_t)((16 * j + 8 * (i >> 1) + 2 * (i & 1)) * N) * 4u);
  }
  __device__ __forceinline__ void prefetch() {
#pragma unroll
    for (int j = 0; j < DE; ++j) request(j, j);
  }
  template <class Next>
  __device__ __forceinline__ void run(const float* __restrict__ tile, const int ld, f4v (&acc)[T], int lane, Next& nx) {
    const float* ap = tile + (lane & 15) * ld + 4 * (lane >> 4);
    float a[2][4];
    ops_of(*reinterpret_cast<const f4v*>(ap), a[0]);          // the dZ tile is in operand order
#pragma unroll
    for (int t = 0; t < T; ++t) acc[t] = f4v{0.f, 0.f, 0.f, 0.f};
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int j = 0; j < NE; ++j) {                      // one scheduling region per unit, see Fwd16::run
      const int c = j & 1, n = c ^ 1;
      f4v af;
      if (j + 1 < NE) af = *reinterpret_cast<const f4v*>(ap + 16 * (j + 1));
      __builtin_amdgcn_sched_barrier(0);
      STEP16_PRIO(1);
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int t = 0; t < T; ++t) STEP16_MFMA(a[c][i], vget<T>(ring[j % DE][i], t), acc[t]);
      STEP16_PRIO(0);
      __builtin_amdgcn_sched_barrier(0);
      if (j + 1 < NE) ops_of(af, a[n]);
      stream_next(*this, nx, j);
    }
    stream_rest<Bwd, Next>(nx);
  }
};

template <int N1, int N2>
constexpr int slot_floats() {      // one weight slot per wave, shared by the two LDS-staged layers
  constexpr int s1 = FwdL<N1 / 128, 32>::kSlotFloats, s2 = FwdL<N2 / 128, 32>::kSlotFloats;
  return s1 > s2 ? s1 : s2;
}
template <int DP, int N0, int N1, int N2>
constexpr size_t lds_floats() {
  return (size_t)kR * ((DP + kPad) + (N0 + kPad) + (N1 + kPad) + (N2 + kPad)) + 16 * 16 + 16 * 16 + 16 * 8 + 4 +
         8 * (size_t)slot_floats<N1, N2>();
}

// acc[t][r] of lane (c16, g4): row 4 g4 + r, column w CW + c16 T + t
// forward epilogue: elu(acc + bias) -> LDS tile, operand order (PERM: a contraction reads it) or natural (the head code
// reads the last layer's tile by column) (+ global copy, natural, for the weight-gradient launch)
template <int T, int N, bool PERM>
__device__ __forceinline__ void fwd_epilogue(const f4v (&acc)[T], const typename VecOf<T>::type bias, float* __restrict__ tile,
                                             float* hg, int64_t M, int64_t r0, int wave, int lane) {
  using V = typename VecOf<T>::type;
  const int c16 = lane & 15, g4 = lane >> 4, col = wave * 16 * T + c16 * T;
  constexpr int ld = N + kPad;
  V o[4];
#pragma unroll
  for (int r = 0; r < 4; ++r)
#pragma unroll
    for (int t = 0; t < T; ++t) vset<T>(o[r], t, gemm::elu_f(acc[t][r] + vget<T>(bias, t)));
#pragma unroll
  for (int r = 0; r < 4; ++r) tile_put<T, PERM>(tile + (4 * g4 + r) * ld, col, o[r]);
  if (hg != nullptr) {
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(hg, 0, (int)(M * N * 4), 0x00020000);
#pragma unroll
    for (int r = 0; r < 4; ++r) bstore<T>(o[r], rs, (uint32_t)(((r0 + 4 * g4 + r) * N + col) * 4));
  }
}
// data-gradient epilogue: dZ = acc * elu'(H) with H from the layer's LDS tile (PERM_IN: stored in operand order); dZ -> the
// same tile in operand order (TO_TILE: a further contraction reads it) and -> global memory.  A wave owns whole 16-column
// blocks, so a tile that changes layout in place (natural H -> operand-order dZ) only moves values among this wave's own
// lanes: every H value is read before the first dZ value is written (LDS operations of a wave execute in order).
template <int T, int N, bool TO_TILE, bool PERM_IN>
__device__ __forceinline__ void bwd_epilogue(const f4v (&acc)[T], float* __restrict__ tile, float* zg, int64_t M,
                                             int64_t r0, int wave, int lane) {
  using V = typename VecOf<T>::type;
  const int c16 = lane & 15, g4 = lane >> 4, col = wave * 16 * T + c16 * T;
  constexpr int ld = N + kPad;
  V hv[4], o[4];
#pragma unroll
  for (int r = 0; r < 4; ++r) hv[r] = tile_get<T, PERM_IN>(tile + (4 * g4 + r) * ld, col);
#pragma unroll
  for (int r = 0; r < 4; ++r)
#pragma unroll
    for (int t = 0; t < T; ++t) {
      const float hx = vget<T>(hv[r], t);
      vset<T>(o[r], t, acc[t][r] * (hx > 0.0f ? 1.0f : hx + 1.0f));      // elu'(z) = 1 (z > 0) | elu(z) + 1
    }
  if (TO_TILE) {
    if (!PERM_IN) __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int r = 0; r < 4; ++r) tile_put<T, true>(tile + (4 * g4 + r) * ld, col, o[r]);
  }
  const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(zg, 0, (int)(M * N * 4), 0x00020000);
#pragma unroll
  for (int r = 0; r < 4; ++r) bstore<T>(o[r], rs, (uint32_t)(((r0 + 4 * g4 + r) * N + col) * 4));
}

}  // namespace step16

// grid = (16-row tiles, 2 networks); blockIdx.y: 0 critic, 1 actor
template <int DP, int N0, int N1, int N2>
__global__ __launch_bounds__(step16::kThreads) void step16_kernel(const step16::Args a) {
  using namespace step16;
  constexpr int T0 = N0 / 128, T1 = N1 / 128, T2 = N2 / 128, HL = N2;
  static_assert(N0 % 128 == 0 && N1 % 128 == 0 && N2 % 128 == 0 && N0 <= 512 && N1 <= 256 && N2 <= 256, "eight waves x 16 T columns; LDS-staged layers T <= 2");
  static_assert(DP % 16 == 0 && DP <= 256, "observation tile");
  constexpr int ldx = DP + kPad, ld0 = N0 + kPad, ld1 = N1 + kPad, ld2 = N2 + kPad;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* tX = smem;                        // [16][ldx]
  float* t0 = tX + kR * ldx;               // [16][ld0]  H0, later dZ0 is NOT written back (nobody reads it)
  float* t1 = t0 + kR * ld0;               // [16][ld1]  H1, later dZ1
  float* t2 = t1 + kR * ld1;               // [16][ld2]  H2, later dZ2
  float* sMu = t2 + kR * ld2;              // [16][16]   head outputs (quarters 0 + 1), later per-row d loss / d logstd_k
  float* sG = sMu + 256;                   // [16][16]   head outputs (quarters 2 + 3), later d loss / d head output
  float* sD = sG + 256;                    // [16][8]    per-row diagnostics
  float* s_adv = sD + 128;                 // [2]
  const HeadArgs& g = a.g;
  const int net = blockIdx.y;              // 0 critic, 1 actor
  const int tile_i = blockIdx.x;
  const int64_t r0 = (int64_t)tile_i * kR;
  const int rows = (int)((a.M - r0) < kR ? (a.M - r0) : kR);
  const int tid = threadIdx.x, lane = tid & 63, c16 = lane & 15, g4 = lane >> 4;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  float* wslot = s_adv + 4 + wave * slot_floats<N1, N2>();   // this wave's weight slot (FwdL)
#if STEP16_PRIO_MODE == 2
  if (wave < 4) __builtin_amdgcn_s_setprio(2);
  else __builtin_amdgcn_s_setprio(0);
#endif
  const int A = g.A;
  const int RB = gridDim.x;
  const int NS = 2 * A + 1 + kHeadDiag;
  const float* P = a.params;

  S16_TL(0);
  // ---- requests in front of everything: layer-0 weights, the observation tile, biases, the row-math operands
  Fwd16<T0, DP> L0;
  FwdL<T1, N0> L1;
  FwdL<T2, N1> L2;
  Bwd<T1, N2, N1> B2;                                  // dZ1 = (dZ2 . W2) * elu'(H1)
  Bwd<T0, N1, N0> B1;                                  // dZ0 = (dZ1 . W1) * elu'(H0)
  NoNext nonext;
  const __amdgpu_buffer_rsrc_t prs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(P), 0, (int)(a.n_flat * 4), 0x00020000);
  L0.init(prs, a.off_w[net][0], wave, lane);
  L0.prefetch();
  L1.init(prs, a.off_w[net][1], wave, lane, wslot);
  L2.init(prs, a.off_w[net][2], wave, lane, wslot);
  B2.init(prs, a.off_w[net][2], wave, lane);
  B1.init(prs, a.off_w[net][1], wave, lane);
  {
    const __amdgpu_buffer_rsrc_t xrs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.x), 0, (int)(a.M * DP * 4), 0x00020000);
    constexpr int q4 = DP / 4, XQ = (kR * q4 + kThreads - 1) / kThreads;
    u32x4 xr[XQ];
#pragma unroll
    for (int j = 0; j < XQ; ++j) {
      const int f = tid + j * kThreads, r = f / q4, q = f - r * q4;
      xr[j] = __builtin_amdgcn_raw_buffer_load_b128(xrs, f < kR * q4 ? (uint32_t)(((r0 + r) * DP + 4 * q) * 4) : 0xffffffffu, 0, 0);
    }
#pragma unroll
    for (int j = 0; j < XQ; ++j) {
      const int f = tid + j * kThreads, r = f / q4, q = f - r * q4;
      if (f < kR * q4) {                             // operand order: (k, k + 2) at pos(k), (k + 1, k + 3) eight further
        float* xp = tX + r * ldx + pcol(4 * q);
        *reinterpret_cast<u32x2*>(xp) = u32x2{xr[j][0], xr[j][2]};
        *reinterpret_cast<u32x2*>(xp + 8) = u32x2{xr[j][1], xr[j][3]};
      }
    }
  }
  // advantage statistics over the minibatch (ppo.py:314-318: mean, unbiased std): wave 4 - idle while waves 0-3 compute
  // the head outputs - requests its partial sums now and reduces them there
  const bool adv_wave = net == 1 && wave == 4;
  double adv_a1 = 0.0, adv_a2 = 0.0;
  if (adv_wave && g.hp.norm_adv && g.adv_stats == nullptr && lane < g.n_adv_part) {
    adv_a1 = g.adv_part[2 * lane];
    adv_a2 = g.adv_part[2 * lane + 1];
  }
  const typename VecOf<T0>::type bias0 = *reinterpret_cast<const typename VecOf<T0>::type*>(P + a.off_b[net][0] + wave * 16 * T0 + c16 * T0);
  const typename VecOf<T1>::type bias1 = *reinterpret_cast<const typename VecOf<T1>::type*>(P + a.off_b[net][1] + wave * 16 * T1 + c16 * T1);
  const typename VecOf<T2>::type bias2 = *reinterpret_cast<const typename VecOf<T2>::type*>(P + a.off_b[net][2] + wave * 16 * T2 + c16 * T2);
  __syncthreads();
  S16_TL(1);

  // ---- forward
  f4v acc0[T0], acc1[T1], acc2[T2];
  L0.run(tX, ldx, acc0, lane, L1);
  S16_TL(2);
  fwd_epilogue<T0, N0, true>(acc0, bias0, t0, a.H[net][0], a.M, r0, wave, lane);
  __syncthreads();
  S16_TL(3);
  L1.run(t0, ld0, acc1, lane, L2);
  S16_TL(4);
  fwd_epilogue<T1, N1, true>(acc1, bias1, t1, a.H[net][1], a.M, r0, wave, lane);
  __syncthreads();
  S16_TL(5);
  // (requested here, one layer ahead of their use, instead of at kernel entry: 14 VGPRs less through layers 0 and 1)
  // row math (wave 0): four threads per row, thread part pp owns the action dims pp, pp + 4, pp + 8, pp + 12
  const int rr = (tid >> 2) & 15, pp = tid & 3;
  const bool rvalid = rr < rows;
  const int64_t ri = r0 + (rvalid ? rr : 0);
  float rs0 = 0.f, rs1 = 0.f, ract[4] = {0.f, 0.f, 0.f, 0.f}, rls[4] = {0.f, 0.f, 0.f, 0.f}, rb[4] = {0.f, 0.f, 0.f, 0.f};
  if (wave == 0) {
    rs0 = net == 1 ? g.oldlogp[ri] : g.ret_n[ri];
    rs1 = net == 1 ? g.adv[ri] : g.val_n[ri];
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) {
      const int k = pp + 4 * kk;
      const bool on = net == 1 && k < A;
      ract[kk] = on ? g.act[ri * A + k] : 0.0f;
      rls[kk] = on ? g.logstd[k] : 0.0f;
      rb[kk] = on ? g.b4a[k] : 0.0f;
    }
  }
  const float rbc = g.b4c[0], rvv = g.vrms_var[0], rvm = g.vrms_mean[0];
  L2.run(t1, ld1, acc2, lane, B2);
  S16_TL(6);
  // what the head steps and the first data gradient read from memory: requested before the last epilogue
  const float* Wh = net == 1 ? g.W4a : g.W4c;          // [KH][HL] head weights of this network
  const int KH = net == 1 ? A : 1;
  constexpr int KQ = HL / 4;                           // step A: contraction share of waves 0-3
  const f4v zero4 = {0.f, 0.f, 0.f, 0.f};
  f4v bwA[KQ / 16];
#pragma unroll
  for (int kb = 0; kb < KQ / 16; ++kb)
    bwA[kb] = (wave < 4 && c16 < KH) ? *reinterpret_cast<const f4v*>(Wh + c16 * HL + wave * KQ + 16 * kb + 4 * g4) : zero4;
  // step B operand: Wh[k][w CW + c16 T2 + t] for the unit's four k rows (rows past KH zero)
  typename VecOf<T2>::type bwB[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int k = krow_of(i, g4);
    typename VecOf<T2>::type z;
#pragma unroll
    for (int t = 0; t < T2; ++t) vset<T2>(z, t, 0.0f);
    bwB[i] = k < KH ? *reinterpret_cast<const typename VecOf<T2>::type*>(Wh + k * HL + wave * 16 * T2 + c16 * T2) : z;
  }
  fwd_epilogue<T2, N2, false>(acc2, bias2, t2, nullptr, a.M, r0, wave, lane);
  __syncthreads();
  S16_TL(7);

  // ---- A: head outputs Y[16][16] = H2 . Wh^T, waves 0-3 a quarter of the contraction each (fwd_head_kernel's order)
  {
    f4v c = zero4;                                     // rows 4 g4 + reg, head output c16
    if (wave < 4) {
#pragma unroll
      for (int kb = 0; kb < KQ / 16; ++kb) {
        const f4v av = *reinterpret_cast<const f4v*>(t2 + c16 * ld2 + wave * KQ + 16 * kb + 4 * g4);
        STEP16_MFMA(av.x, bwA[kb].x, c);
        STEP16_MFMA(av.y, bwA[kb].y, c);
        STEP16_MFMA(av.z, bwA[kb].z, c);
        STEP16_MFMA(av.w, bwA[kb].w, c);
      }
    }
    if (adv_wave) {
      if (g.hp.norm_adv && g.adv_stats == nullptr) {
        double a1 = adv_a1, a2 = adv_a2;
        for (int b = lane + 64; b < g.n_adv_part; b += 64) {
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
    // the four quarters in fixed order, two rounds: sMu = q0 + q1, sG = q2 + q3 (sG is free until the row math writes
    // it); the row math adds the two halves
    for (int w = 0; w < 2; ++w) {
      if (wave < 4 && (wave & 1) == w) {
        float* half = (wave >> 1) ? sG : sMu;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          float* d = half + (4 * g4 + r) * 16 + c16;
          *d = (w == 0 ? 0.0f : *d) + c[r];
        }
      }
      __syncthreads();
    }
  }

  S16_TL(8);
  // ---- row math (wave 0; the arithmetic of fwd_head_kernel / head_loss_kernel, ppo.py:299-345)
  const float clipc = g.hp.clip_coef, invM = g.hp.inv_global_batch;
  if (wave == 0) {
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
        if (g.branch_out != nullptr && pp == 0) g.branch_out[ri] = clip_code(ratio, 1.0f, clipc);
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
      if (g.branch_out != nullptr) g.branch_out[g.M + ri] = clip_code(dl, 0.0f, clipc) | ((vl1 > vl2 ? 1 : (vl1 < vl2 ? 2 : 0)) << 2);
      const bool in2 = dl >= -clipc && dl <= clipc;
      const float dnv_c = vl1 > vl2 ? 2.0f * e1 : (vl1 < vl2 ? (in2 ? 2.0f * e2 : 0.0f) : e1 + (in2 ? e2 : 0.0f));
      const float vl = clip_vloss ? (vl1 > vl2 ? vl1 : vl2) : vl1;
      const float dnv = clip_vloss ? dnv_c : 2.0f * e1;
      dg[1] = 0.5f * vl;
      gm[0] = vf_half * dnv * invM / vden;         // d loss / d v_i  (slot 0)
    }
    // (every read of sMu / sG above belongs to this wave, whose LDS operations execute in order)
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) sG[r * 16 + pp + 4 * kk] = gm[kk], sMu[r * 16 + pp + 4 * kk] = gl[kk];
    if (pp == 0) {
#pragma unroll
      for (int e = 0; e < 8; ++e) sD[r * 8 + e] = dg[e];
    }
  }
  __syncthreads();
  S16_TL(9);

  // ---- C: head weight gradient of the tile, dWh[k][c] = sum_r G[r][k] H2[r][c]; wave w owns columns [w CW, w CW + CW)
  const int prow = net == 1 ? tile_i : RB + tile_i;
  {
    f4v cc[T2];
#pragma unroll
    for (int t = 0; t < T2; ++t) cc[t] = zero4;
    const int col0 = wave * 16 * T2;
#pragma unroll
    for (int s = 0; s < kR / 4; ++s) {
      const int r = 4 * s + g4;
      const float av = sG[r * 16 + c16];
#pragma unroll
      for (int t = 0; t < T2; ++t) STEP16_MFMA(av, t2[r * ld2 + col0 + 16 * t + c16], cc[t]);
    }
    float* pw = g.part_w + (int64_t)prow * (A + 1) * HL + (net == 1 ? 0 : (int64_t)A * HL);   // rows 0..A-1 = dW4a, row A = dW4c
#pragma unroll
    for (int t = 0; t < T2; ++t)
#pragma unroll
      for (int r = 0; r < 4; ++r) {                      // accumulator row 4 g4 + r = head output
        const int k = 4 * g4 + r;
        if (k < KH) pw[k * HL + col0 + 16 * t + c16] = cc[t][r];
      }
  }
  // ---- B: dZ2 = (G . Wh) * elu'(H2), in place on the wave's own columns (step C above read the same columns: LDS
  //         operations of a wave execute in order)
  {
    f4v cb[T2];
#pragma unroll
    for (int t = 0; t < T2; ++t) cb[t] = zero4;
    float ga[4];
    prep(*reinterpret_cast<const f4v*>(sG + c16 * 16 + 4 * g4), ga);
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int t = 0; t < T2; ++t) STEP16_MFMA(ga[i], vget<T2>(bwB[i], t), cb[t]);
    __builtin_amdgcn_wave_barrier();
    bwd_epilogue<T2, N2, true, false>(cb, t2, a.dZ[net][2], a.M, r0, wave, lane);
  }
  __syncthreads();
  S16_TL(10);

  // ---- data gradients of the hidden layers
  f4v ax1[T1];
  B2.run(t2, ld2, ax1, lane, B1);
  S16_TL(11);
  bwd_epilogue<T1, N1, true, true>(ax1, t1, a.dZ[net][1], a.M, r0, wave, lane);
  __syncthreads();
  S16_TL(12);
  f4v ax0[T0];
  B1.run(t1, ld1, ax0, lane, nonext);
  S16_TL(13);
  bwd_epilogue<T0, N0, false, true>(ax0, t0, a.dZ[net][0], a.M, r0, wave, lane);

  // ---- scalars of the tile: bias / logstd gradients, diagnostics (rows in fixed order)
  float* ps = g.part_s + (int64_t)prow * NS;
  if (tid < 40) {
    float v = 0.0f;
    if (tid < 16) {
      for (int r = 0; r < kR; ++r) v += sG[r * 16 + tid];
      if (net == 1) { if (tid < A) ps[tid] = v; }          // db4a[k]
      else if (tid == 0) ps[A] = v;                        // db4c
    } else if (tid < 32) {
      for (int r = 0; r < kR; ++r) v += sMu[r * 16 + tid - 16];
      if (net == 1 && tid - 16 < A) ps[A + 1 + tid - 16] = v;   // dlogstd[k]
    } else {
      for (int r = 0; r < kR; ++r) v += sD[r * 8 + tid - 32];
      ps[2 * A + 1 + tid - 32] = v;                        // diagnostics
    }
  }
  S16_TL(14);
}

// Rollout forward on the same 16-row tiles (policy step + value, cleanrl/ppo.py:104-119,186-189; bootstrap value :251):
// the three layers of step16_kernel and the heads of the row-resident rollout kernels (fused_head: Philox action noise,
// log-prob, value).  For batches that leave the 32-row kernels with fewer workgroups than CUs - an env-sharded rank's 2048
// envs ran three layer-wise GEMM launches + head_act_kernel (36 us per env step).  grid = (16-row tiles, networks).
template <int DP, int N0, int N1, int N2>
__global__ __launch_bounds__(step16::kThreads) void step16_fwd_kernel(const FusedFwdArgs a) {
  using namespace step16;
  constexpr int T0 = N0 / 128, T1 = N1 / 128, T2 = N2 / 128;
  constexpr int ldx = DP + kPad, ld0 = N0 + kPad, ld1 = N1 + kPad, ld2 = N2 + kPad;
  static_assert(16 * (N2 + 4) <= kR * ld0, "the head weights are staged over the first activation tile");
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* tX = smem;
  float* t0 = tX + kR * ldx;
  float* t1 = t0 + kR * ld0;
  float* t2 = t1 + kR * ld1;
  const int net = a.net0 + blockIdx.y;
  const int64_t r0 = (int64_t)blockIdx.x * kR;
  const int tid = threadIdx.x, lane = tid & 63, c16 = lane & 15;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  float* wslot = t2 + kR * ld2 + wave * slot_floats<N1, N2>();
  const float* P = a.params;
  Fwd16<T0, DP> L0;
  FwdL<T1, N0> L1;
  FwdL<T2, N1> L2;
  NoNext nonext;
  const __amdgpu_buffer_rsrc_t prs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(P), 0, (int)(a.n_flat * 4), 0x00020000);
  L0.init(prs, a.off_w[net][0], wave, lane);
  L0.prefetch();
  L1.init(prs, a.off_w[net][1], wave, lane, wslot);
  L2.init(prs, a.off_w[net][2], wave, lane, wslot);
  {
    const __amdgpu_buffer_rsrc_t xrs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.x), 0, (int)(a.M * DP * 4), 0x00020000);
    constexpr int q4 = DP / 4, XQ = (kR * q4 + kThreads - 1) / kThreads;
    u32x4 xr[XQ];
#pragma unroll
    for (int j = 0; j < XQ; ++j) {
      const int f = tid + j * kThreads, r = f / q4, q = f - r * q4;
      xr[j] = __builtin_amdgcn_raw_buffer_load_b128(xrs, f < kR * q4 ? (uint32_t)(((r0 + r) * DP + 4 * q) * 4) : 0xffffffffu, 0, 0);
    }
#pragma unroll
    for (int j = 0; j < XQ; ++j) {
      const int f = tid + j * kThreads, r = f / q4, q = f - r * q4;
      if (f < kR * q4) {                             // operand order: (k, k + 2) at pos(k), (k + 1, k + 3) eight further
        float* xp = tX + r * ldx + pcol(4 * q);
        *reinterpret_cast<u32x2*>(xp) = u32x2{xr[j][0], xr[j][2]};
        *reinterpret_cast<u32x2*>(xp + 8) = u32x2{xr[j][1], xr[j][3]};
      }
    }
  }
  const typename VecOf<T0>::type bias0 = *reinterpret_cast<const typename VecOf<T0>::type*>(P + a.off_b[net][0] + wave * 16 * T0 + c16 * T0);
  const typename VecOf<T1>::type bias1 = *reinterpret_cast<const typename VecOf<T1>::type*>(P + a.off_b[net][1] + wave * 16 * T1 + c16 * T1);
  const typename VecOf<T2>::type bias2 = *reinterpret_cast<const typename VecOf<T2>::type*>(P + a.off_b[net][2] + wave * 16 * T2 + c16 * T2);
#if STEP16_PRIO_MODE == 2
  if (wave < 4) __builtin_amdgcn_s_setprio(2);
  else __builtin_amdgcn_s_setprio(0);
#endif
  __syncthreads();
  f4v acc0[T0], acc1[T1], acc2[T2];
  L0.run(tX, ldx, acc0, lane, L1);
  fwd_epilogue<T0, N0, true>(acc0, bias0, t0, nullptr, a.M, r0, wave, lane);
  __syncthreads();
  L1.run(t0, ld0, acc1, lane, L2);
  fwd_epilogue<T1, N1, true>(acc1, bias1, t1, nullptr, a.M, r0, wave, lane);
  __syncthreads();
  L2.run(t1, ld1, acc2, lane, nonext);
  fwd_epilogue<T2, N2, false>(acc2, bias2, t2, nullptr, a.M, r0, wave, lane);
  __syncthreads();
  if (a.do_head) fused_head<N2, kR>(a, t2, ld2, net, r0, t0);      // (the first activation tile is free: head weights go there)
}
#undef STEP16_MFMA
