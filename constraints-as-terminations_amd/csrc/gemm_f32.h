// fp32-input MFMA GEMM for the actor-critic MLP (gfx950, wave64).
//
//   C[i,j] = epilogue( sum_k A(i,k) * B(k,j) )        i<I, j<J, k<Kc
//
// v_mfma_f32_32x32x2_f32 (exact fp32, 64 FLOP/clk/SIMD = the 157 TF/s fp32 matrix peak; there
// is no TF32/xf32 on gfx950).  One 256-thread workgroup (4 waves as 2x2) owns a BMxBN tile,
// each wave a (BM/2)x(BN/2) sub-tile = TMxTN accumulators of 32x32.  K is walked in BK=16
// slabs, double-buffered in LDS, global->register->LDS staged so the next slab's loads are in
// flight during the MFMAs of the current one (one barrier per slab).
//
// Each operand is either "K-contiguous" (element (r,k) at P[r*ld+k]: activations X[m,:],
// weights W[n,:]) or "I-contiguous" (element (k,r) at P[k*ld+r]).  The three GEMMs of a layer:
//   forward     Y  = X  W^T     A=X  (K-contig)  B=W  (K-contig)
//   data grad   dX = dY W       A=dY (K-contig)  B=W  (I-contig)
//   weight grad dW = dY^T X     A=dY (I-contig)  B=X  (I-contig)   contraction = batch (split-K)
// so no operand is ever transposed in memory.  LDS images:
//   K-contig: [rows][BK] with row stride BK+4 floats; a lane fetches 4 consecutive k with ONE
//             ds_read_b128.  gfx950 serves a b128 read in four groups of 16 lanes - {0-3,12-15,20-27}, {4-11,16-19,
//             28-31} and the same +32 (MI355X_MICROARCH.md, LDS table), NOT 16 consecutive lanes - over 64 banks; the
//             rows of a group are distinct mod 16 and the 80-B row stride (20 banks, 20 = 4 x 5) maps them to 16
//             distinct 4-bank slots, so the READS are conflict free.  The image is WRITTEN with ds_write_b128 in groups
//             of 8 consecutive lanes over 32 banks: lanes 0-3 = row r (banks 20r .. 20r+15), lanes 4-7 = row r+1, whose
//             last quad wraps onto row r's first (20 + 12 = 32 = 0 mod 32) - a 2-way conflict per write.  That is what
//             SQ_LDS_BANK_CONFLICT sees in the K-contiguous kernels (0.27-0.32 conflict cycles per LDS-active cycle in
//             round 2's SQ pass; 0.03 with 64-k slabs, whose 272-B stride has no wrap); with 3-4 % of the wave cycles
//             issue-stalled on LDS it is not worth a layout that would break the 4-lanes-per-64-B-segment global load.
//             MFMA step s of lane-half h uses k = 8*blk + 4*h + s: a permutation of the contraction order shared by
//             A and B.
//   I-contig: [BK][rows]; a lane fetches element (k, row) with ds_read_b32 (two groups of 32 lanes over 32 banks), the
//             32 lanes of a half read 32 consecutive floats -> conflict-free.
#pragma once

#include "common.h"

#include <type_traits>

namespace gemm {

using f32x16 = __attribute__((ext_vector_type(16))) float;
using bf16x8 = __attribute__((ext_vector_type(8))) __bf16;

constexpr int BK = 16;             // default contraction slab; kernels take it as the BKT template parameter

#ifndef GEMM_PIPE
#define GEMM_PIPE 1                // 0: the round-2 main loop (A/B builds: tools/gpu_ab.sh)
#endif

#ifndef GEMM_STAGE_PARTIAL
#define GEMM_STAGE_PARTIAL 1       // 0: split-K partials stored with single-dword write-through stores (rounds 1-5) - A/B builds
#endif

#ifndef GEMM_ROW_SWIZZLE
#define GEMM_ROW_SWIZZLE 1         // 0: K-contiguous staging rows in thread order (2-way ds_write_b128 conflicts, rounds 1-5) - A/B builds
#endif

#ifndef GEMM_PACKED
#define GEMM_PACKED 1              // 0: bf16 / split-bf16 operands converted at every use (rounds 1-4) - A/B builds
#endif

enum Epilogue {
  EPI_BIAS_ELU = 0,  // C = elu(acc + bias[j])                       forward hidden layer
  EPI_MUL_DELU = 1,  // C = acc * elu'(aux[i,j]) (aux = activation)  data gradient
  EPI_PARTIAL = 2,   // C[split] = acc (+ column sums of A -> dbias) weight gradient, split-K
  EPI_BIAS_ELU_LDS = 3,  // elu(acc + bias[j]) left in LDS as a [BM][BN + kLdsTilePad] tile for a fused consumer
};
constexpr int kLdsTilePad = 4;    // row stride BN + 4 floats: a b128 read by 16 consecutive rows hits 16 distinct 16-B slots

struct Operands {
  const float* A;
  const float* B;
  float* C;
  const float* bias;  // EPI_BIAS_ELU: [J]
  const float* aux;   // EPI_MUL_DELU: [I, ldaux]
  float* dbias;       // EPI_PARTIAL: [splits, I] column sums of A over the split (may be null)
};

struct Params {
  Operands op[2];  // grouped launch: blockIdx.z % nets selects critic / actor
  int nets;
  int I, J, Kc;
  int lda, ldb, ldc, ldaux;
  int splits;            // EPI_PARTIAL: contraction split count (blockIdx.z / nets)
  int kc_per_split;      // multiple of BK
  int64_t c_split_stride;  // floats between consecutive split outputs
  int xcd_legacy;        // 1: round-2 workgroup order (tiles permuted inside one (net, split) only) - A/B switch
};

// ELU(alpha=1).  exp through the hardware exp2 (v_exp_f32, ~1 ulp on the (0,1] range that matters
// here): |error| < 2e-7 absolute on the activation, far inside the 1e-5 parity budget, and 10x
// fewer VALU instructions than expf in an epilogue that runs 64 of them per lane.
__device__ __forceinline__ float elu_f(float z) {
  const float e = __builtin_amdgcn_exp2f(z * 1.44269504088896340736f) - 1.0f;
  return z > 0.0f ? z : e;
}

#ifdef GEMM_TIMELINE   // tools/pair_timeline.hip: per-workgroup stamps of the main loop: [0] / [3] wall clock, [1] / [2] shader clock at its start / end
__device__ unsigned long long* g_tl;
#endif

// XCD-aware tile order (speed only): workgroup b runs on XCD b % 8, each XCD has a private L2.  Give every XCD a
// contiguous range of the tile sequence, so the tiles that share an operand row block hit the same L2 instead of
// fetching it once per XCD through the fabric.  Bijective for any workgroup count.
__device__ __forceinline__ int xcd_tile_index(int wg, int nwg) {
  const int q = nwg >> 3, r = nwg & 7, xcd = wg & 7;
  return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (wg >> 3);
}

// (tile, bz) of a workgroup of a launch whose grid is `tiles` x `nz` problems-slices (bz = net + nets * split), given its
// linear position `lin` in launch order.  The hardware deals workgroups to the XCDs round-robin over the WHOLE launch
// (lin % 8), so the unit that must stay together on one L2 is not "the tiles of this z-slice" but "consecutive logical
// indices": every XCD gets a contiguous range of (bz, tile) pairs, tile fastest.  Round 2 permuted the tile index
// inside one z-slice only; with few tiles per slice (a 256x256 weight gradient: 4 tiles x 64 (net, split) slices) the
// four tiles that read the SAME 512-row slabs of dZ and H then sat on four different XCDs and every slab was fetched
// from HBM twice (rocprofv3 FETCH_SIZE of the paired launch: 203 MB for 134 MB of operands).
struct TileId {
  int tile, bz;
};
__device__ __forceinline__ TileId xcd_tile_of(int lin, int tiles, int nz, int legacy) {
  if (legacy) return TileId{xcd_tile_index(lin % tiles, tiles), lin / tiles};
  const int l = xcd_tile_index(lin, tiles * nz);
  return TileId{l % tiles, l / tiles};
}


// ---- bf16 / split-bf16 operands, split ONCE per workgroup where the slab is staged (round 5) ----------------------------
// Rounds 1-4 kept fp32 LDS images for every precision and rounded / split each operand element to bf16 at EVERY use: a
// 128x128 tile's four waves each convert the fragments they read (every element twice), ~3.5 VALU operations per
// element and use for the split form - the bf16x3 main loops were VALU / LDS bound, not matrix bound.
// Here the thread that brings a float4 in from memory splits it - hi = bf16(x) (RNE), lo = bf16(x - hi) - and writes
// bf16 PLANES to LDS; the inner loop is LDS reads + MFMAs only:
//   K-contiguous operand   plane [rows][16 k] = 32 B rows, the two 16-byte halves of a row swapped in every second group
//                          of 8 rows (the b128 lane groups of gfx950 - {0-3, 12-15, 20-27}, ... - then hit 16 distinct
//                          16-byte slots: conflict free WITHOUT padding); a lane's fragment = ONE ds_read_b128 per plane
//   I-contiguous operand   plane [8 k-pairs][rows] of u32 = {bf16 x(k even, row), bf16 x(k odd, row)}: the staging thread
//                          loads the SAME four rows at k and k + 1 and packs the pairs (one ds_write_b128 per plane); a
//                          lane's fragment = four ds_read_b32 per plane (16 for the fp32 image)
// Slot e of lane half h of v_mfma_f32_32x32x16_bf16 carries k = 8 h + e for A and B alike.  The planes of a (128 + 128)-
// row slab pair take 128 B per row and buffer pair at most: inside the fp32 allocation of every launcher.
using bf16x2 = __attribute__((ext_vector_type(2))) __bf16;
using u32x4 = __attribute__((ext_vector_type(4))) uint32_t;
using u32x2 = __attribute__((ext_vector_type(2))) uint32_t;

__device__ __forceinline__ uint32_t pk_bf16(float a, float b) {
  bf16x2 v;
  v[0] = (__bf16)a, v[1] = (__bf16)b;
  return __builtin_bit_cast(uint32_t, v);
}
__device__ __forceinline__ float bf16_resid(float x) { return x - (float)(__bf16)x; }

template <int BM, int BN, bool A_KC, bool B_KC, int EPI, int PREC, int WAVES_M>
struct PkLoop {
  static constexpr int NP = PREC == 2 ? 2 : 1;
  static constexpr int WAVES_N = 4 / WAVES_M;
  static constexpr int WM = BM / WAVES_M, WN = BN / WAVES_N, TM = WM / 32, TN = WN / 32;
  static constexpr int A_PLANE = BM * 8, B_PLANE = BN * 8;          // u32 per plane and buffer (16 k x rows x 2 B)
  // float4 staging registers per thread and slab.  K-contiguous: rows x 4 quads over 256 threads; I-contiguous: one
  // (k-pair, row quad) job = two float4 (k even / odd), 2 x rows jobs over 256 threads
  template <int ROWS, bool KC> static constexpr int nreg() { return KC ? ROWS / 64 : (2 * ROWS + 255) / 256 * 2; }
  static constexpr int A_N = nreg<BM, A_KC>(), B_N = nreg<BN, B_KC>();
  static_assert((A_KC || BM <= 128) && (B_KC || BN <= 128), "one (k pair, row quad) job per thread for I-contiguous operands");
  static_assert(NP * 2 * (A_PLANE + B_PLANE) * 4 <= 2 * 4 * ((A_KC ? BM * 20 : 16 * BM) + (B_KC ? BN * 20 : 16 * BN)),
                "the bf16 planes must fit the fp32 slab allocation of the launchers");

  template <int ROWS, bool KC, int N>
  static __device__ __forceinline__ void gload(float4 (&r)[N], const float* __restrict__ P, int ld, int row0, int rows_valid,
                                               int k0, int k_end, bool fast, int tid) {
    if (KC) {
#pragma unroll
      for (int q = 0; q < N; ++q) {
        const int f = tid + q * 256, rr = f >> 2, kq = f & 3;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (fast || row0 + rr < rows_valid) v = *reinterpret_cast<const float4*>(P + (int64_t)(row0 + rr) * ld + k0 + 4 * kq);
        r[q] = v;
      }
    } else {
#pragma unroll
      for (int q = 0; q < N / 2; ++q) {
        const int j = tid + q * 256, kp = j / (ROWS / 4), iq = j % (ROWS / 4);
        float4 v0 = make_float4(0.f, 0.f, 0.f, 0.f), v1 = v0;
        if (2 * ROWS >= 256 || j < 2 * ROWS) {                        // (64-row tiles: jobs for the first two waves only)
          const int gk = k0 + 2 * kp, gi = row0 + 4 * iq;
          const bool in_i = fast || gi + 3 < rows_valid;
          if (in_i && (fast || gk < k_end)) v0 = *reinterpret_cast<const float4*>(P + (int64_t)gk * ld + gi);
          if (in_i && (fast || gk + 1 < k_end)) v1 = *reinterpret_cast<const float4*>(P + (int64_t)(gk + 1) * ld + gi);
        }
        r[2 * q] = v0, r[2 * q + 1] = v1;
      }
    }
  }

  template <int ROWS, bool KC, int N>
  static __device__ __forceinline__ void lstore(uint32_t* __restrict__ planes, const float4 (&r)[N], int tid) {
    if (KC) {
#pragma unroll
      for (int q = 0; q < N; ++q) {
        const int f = tid + q * 256, rr = f >> 2, kq = f & 3;
        const int o = rr * 8 + ((((kq >> 1) ^ (rr >> 3)) & 1) << 2) + ((kq & 1) << 1);
        const float4 v = r[q];
        u32x2 hi;
        hi[0] = pk_bf16(v.x, v.y), hi[1] = pk_bf16(v.z, v.w);
        *reinterpret_cast<u32x2*>(planes + o) = hi;
        if (NP == 2) {
          u32x2 lo;
          lo[0] = pk_bf16(bf16_resid(v.x), bf16_resid(v.y)), lo[1] = pk_bf16(bf16_resid(v.z), bf16_resid(v.w));
          *reinterpret_cast<u32x2*>(planes + ROWS * 8 + o) = lo;
        }
      }
    } else {
#pragma unroll
      for (int q = 0; q < N / 2; ++q) {
        const int j = tid + q * 256, kp = j / (ROWS / 4), iq = j % (ROWS / 4);
        if (2 * ROWS >= 256 || j < 2 * ROWS) {
          const float4 a = r[2 * q], b = r[2 * q + 1];
          u32x4 hi;
          hi[0] = pk_bf16(a.x, b.x), hi[1] = pk_bf16(a.y, b.y), hi[2] = pk_bf16(a.z, b.z), hi[3] = pk_bf16(a.w, b.w);
          *reinterpret_cast<u32x4*>(planes + kp * ROWS + 4 * iq) = hi;
          if (NP == 2) {
            u32x4 lo;
            lo[0] = pk_bf16(bf16_resid(a.x), bf16_resid(b.x)), lo[1] = pk_bf16(bf16_resid(a.y), bf16_resid(b.y));
            lo[2] = pk_bf16(bf16_resid(a.z), bf16_resid(b.z)), lo[3] = pk_bf16(bf16_resid(a.w), bf16_resid(b.w));
            *reinterpret_cast<u32x4*>(planes + ROWS * 8 + kp * ROWS + 4 * iq) = lo;
          }
        }
      }
    }
  }

  // the 8 bf16 (k = 8 h .. 8 h + 7) of operand row `row` out of one plane
  template <int ROWS, bool KC>
  static __device__ __forceinline__ bf16x8 frag(const uint32_t* __restrict__ plane, int row, int h) {
    u32x4 v;
    if (KC) {
      v = *reinterpret_cast<const u32x4*>(plane + row * 8 + (((h ^ (row >> 3)) & 1) << 2));
    } else {
#pragma unroll
      for (int i = 0; i < 4; ++i) v[i] = plane[(4 * h + i) * ROWS + row];
    }
    return __builtin_bit_cast(bf16x8, v);
  }

  static __device__ __forceinline__ void run(const Params& p, const Operands& op, const int i0, const int j0,
                                             const int k_begin, const int k_end, const bool do_db,
                                             float* __restrict__ smem, f32x16 (&acc)[TM][TN], float& dbsum) {
    uint32_t* As = reinterpret_cast<uint32_t*>(smem);                  // [2][NP][A_PLANE]
    uint32_t* Bs = As + 2 * NP * A_PLANE;                              // [2][NP][B_PLANE]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = WAVES_M == 2 ? wave >> 1 : 0, wn = WAVES_M == 2 ? wave & 1 : wave;
    const int l31 = lane & 31, h = lane >> 5;
    float4 ra[A_N], rb[B_N];
    const int n_slabs = (k_end - k_begin + 15) / 16;
    // every row / column of the tile in bounds and the contraction range a whole number of slabs: unguarded loads
    const bool fast = (i0 + BM <= p.I) && (j0 + BN <= p.J) && ((k_end - k_begin) % 16 == 0);
    float dbv[4] = {0.f, 0.f, 0.f, 0.f};                               // EPI_PARTIAL: exact fp32 column sums of A (bias gradient)
    auto load = [&](int s) {
      const int k0 = k_begin + 16 * s;
      gload<BM, A_KC>(ra, op.A, p.lda, i0, p.I, k0, k_end, fast, tid);
      gload<BN, B_KC>(rb, op.B, p.ldb, j0, p.J, k0, k_end, fast, tid);
      if (EPI == EPI_PARTIAL && !A_KC) {
        if (do_db) {
#pragma unroll
          for (int q = 0; q < A_N; ++q) dbv[0] += ra[q].x, dbv[1] += ra[q].y, dbv[2] += ra[q].z, dbv[3] += ra[q].w;
        }
      }
    };
    auto store = [&](int buf) {
      lstore<BM, A_KC>(As + buf * NP * A_PLANE, ra, tid);
      lstore<BN, B_KC>(Bs + buf * NP * B_PLANE, rb, tid);
    };
    if (n_slabs > 0) {
      load(0);
      store(0);
    }
    __syncthreads();
    for (int s = 0; s < n_slabs; ++s) {
      const int cur = s & 1;
      if (s + 1 < n_slabs) load(s + 1);                                // in flight under the MFMAs below
      const uint32_t* a = As + cur * NP * A_PLANE;
      const uint32_t* b = Bs + cur * NP * B_PLANE;
      bf16x8 pa[TM], pb[TN], la[TM], lb[TN];
#pragma unroll
      for (int t = 0; t < TM; ++t) {
        pa[t] = frag<BM, A_KC>(a, wm * WM + t * 32 + l31, h);
        if (NP == 2) la[t] = frag<BM, A_KC>(a + A_PLANE, wm * WM + t * 32 + l31, h);
      }
#pragma unroll
      for (int t = 0; t < TN; ++t) {
        pb[t] = frag<BN, B_KC>(b, wn * WN + t * 32 + l31, h);
        if (NP == 2) lb[t] = frag<BN, B_KC>(b + B_PLANE, wn * WN + t * 32 + l31, h);
      }
      if (NP == 2) {                                                   // small terms first (order of rounds 2-4)
#pragma unroll
        for (int tm = 0; tm < TM; ++tm)
#pragma unroll
          for (int tn = 0; tn < TN; ++tn) {
            acc[tm][tn] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(la[tm], pb[tn], acc[tm][tn], 0, 0, 0);
            acc[tm][tn] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(pa[tm], lb[tn], acc[tm][tn], 0, 0, 0);
          }
      }
#pragma unroll
      for (int tm = 0; tm < TM; ++tm)
#pragma unroll
        for (int tn = 0; tn < TN; ++tn)
          acc[tm][tn] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(pa[tm], pb[tn], acc[tm][tn], 0, 0, 0);
      if (s + 1 < n_slabs) store(cur ^ 1);
      __syncthreads();
    }
    if (EPI == EPI_PARTIAL && !A_KC) {
      // bias gradient: the column sums of A over this split, exact fp32, from the values the staging threads saw: thread
      // (k pair, row quad) holds the sums of its two k rows over all slabs; the 8 k-pair groups are added in fixed order
      if (do_db) {                                                     // workgroup-uniform
        if (2 * BM >= 256 || tid < 2 * BM) {
          const int kp = tid / (BM / 4), iq = tid % (BM / 4);
#pragma unroll
          for (int i = 0; i < 4; ++i) smem[kp * BM + 4 * iq + i] = dbv[i];
        }
        __syncthreads();
        if (tid < BM) {
          float sum = 0.0f;
#pragma unroll
          for (int kp = 0; kp < 8; ++kp) sum += smem[kp * BM + tid];
          dbsum = sum;
        }
        __syncthreads();
      }
    }
  }
};


// ---- bf16-STORED operands (round 6, "act16": BASELINE configs[4], operand precision 1) ---------------------------------------
// With bf16 matrix operands the optimiser-step group is HBM-bound (677 MB per 16384-row group at 5 TB/s)
// while every activation H and every dZ crossed HBM as fp32 only to be rounded to bf16 where a GEMM consumed it.  Storing them
// AS bf16 rounds at the producer instead of the consumer - the GEMM operands are the same bf16 values, every forward value
// agrees to fp32 rounding (the k of an instruction sit in other operand slots: another internal summation order); what differs
// beyond that: elu'(H) and the bias-gradient column sums see the rounded values - and halves the activation traffic.  Round 5 built
// this once with 8-byte loads (4 bf16 per lane) and found it slower; here every lane still moves 16 bytes per load:
//   K-contiguous operands (forward, data gradient): a [rows][K] bf16 matrix IS a [rows][K / 2] float matrix for the staging
//       code of gemm_body - same loads, same LDS image, same b128 fragment reads - whose 16-"float" slab is a 32-k slab; a
//       lane's float4 fragment (floats 8 blk + 4 h ..) is the 8 bf16 k = 16 blk + 8 h + e one v_mfma_f32_32x32x16_bf16 wants.
//       The host passes lda / ldb / Kc of such a problem in FLOAT units (PREC 3; weights come as bf16 copies, W and W^T).
//   I-contiguous operands (weight gradient, contraction over the minibatch rows): IILoop16 below - 32-k slabs, a thread loads
//       8 consecutive rows (16 B) at k and k + 1 and interleaves them into the k-pair plane [16 k pairs][rows] of u32 the
//       split-bf16 loop (PkLoop) reads: fragment = four ds_read_b32.  B may be an fp32 matrix (the observations of the first
//       layer: PREC 5): float4 of four rows at k, k + 1, packed.
// PREC:  3 = both operands bf16-stored (outputs / aux bf16 too: EPI_BIAS_ELU, EPI_MUL_DELU)   4 = fp32-stored operands rounded at
//        use (PREC 1's loop) with a bf16-stored OUTPUT (first layer forward)   5 = weight gradient with an fp32-stored B
//        6 = 3 with an fp32-stored output (last hidden layer of a rollout forward: the head kernel reads fp32).
template <int BM, int BN, int EPI, bool B16, int WAVES_M>
struct IILoop16 {
  static constexpr int WAVES_N = 4 / WAVES_M;
  static constexpr int WM = BM / WAVES_M, WN = BN / WAVES_N, TM = WM / 32, TN = WN / 32;
  static constexpr int A_PLANE = 16 * BM, B_PLANE = 16 * BN;         // u32 per buffer
  static_assert(BM <= 128 && BN <= 128 && (B16 || BN <= 64), "one staging job per thread and operand");
  static_assert(EPI == EPI_PARTIAL, "the weight-gradient loop");

  static __device__ __forceinline__ void run(const Params& p, const Operands& op, const int i0, const int j0,
                                             const int k_begin, const int k_end, const bool do_db,
                                             float* __restrict__ smem, f32x16 (&acc)[TM][TN], float& dbsum) {
    uint32_t* As = reinterpret_cast<uint32_t*>(smem);                  // [2][A_PLANE]
    uint32_t* Bs = As + 2 * A_PLANE;                                   // [2][B_PLANE]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = WAVES_M == 2 ? wave >> 1 : 0, wn = WAVES_M == 2 ? wave & 1 : wave;
    const int l31 = lane & 31, h = lane >> 5;
    const uint16_t* A16 = reinterpret_cast<const uint16_t*>(op.A);
    const uint16_t* B16p = reinterpret_cast<const uint16_t*>(op.B);
    const int n_slabs = (k_end - k_begin + 31) / 32;
    // staging jobs: A (and a bf16-stored B): (k pair kp, row octet io); fp32-stored B: (k pair, row quad)
    const bool a_job = tid < 2 * BM;
    const int a_kp = tid / (BM / 8), a_io = tid % (BM / 8);
    const bool b_job = B16 ? tid < 2 * BN : tid < 4 * BN;
    const int b_kp = B16 ? tid / (BN / 8) : tid / (BN / 4), b_io = B16 ? tid % (BN / 8) : tid % (BN / 4);
    const bool a_in = i0 + 8 * a_io + 7 < p.I;
    const bool b_in = B16 ? j0 + 8 * b_io + 7 < p.J : j0 + 4 * b_io + 3 < p.J;
    u32x4 ra0, ra1, rb0, rb1;
    float dbv[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};        // exact fp32 column sums of the (bf16) values of A this thread staged
    const u32x4 z4 = {0u, 0u, 0u, 0u};
    auto load = [&](int s) {
      const int k0 = k_begin + 32 * s;
      ra0 = z4, ra1 = z4, rb0 = z4, rb1 = z4;
      if (a_job && a_in) {
        const int gk = k0 + 2 * a_kp;
        const uint16_t* src = A16 + (int64_t)gk * p.lda + i0 + 8 * a_io;
        if (gk < k_end) ra0 = *reinterpret_cast<const u32x4*>(src);
        if (gk + 1 < k_end) ra1 = *reinterpret_cast<const u32x4*>(src + p.lda);
      }
      if (b_job && b_in) {
        const int gk = k0 + 2 * b_kp;
        if (B16) {
          const uint16_t* src = B16p + (int64_t)gk * p.ldb + j0 + 8 * b_io;
          if (gk < k_end) rb0 = *reinterpret_cast<const u32x4*>(src);
          if (gk + 1 < k_end) rb1 = *reinterpret_cast<const u32x4*>(src + p.ldb);
        } else {
          const float* src = op.B + (int64_t)gk * p.ldb + j0 + 4 * b_io;
          if (gk < k_end) rb0 = *reinterpret_cast<const u32x4*>(src);
          if (gk + 1 < k_end) rb1 = *reinterpret_cast<const u32x4*>(src + p.ldb);
        }
      }
      if (do_db) {                                                     // workgroup-uniform
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          dbv[2 * j] += __uint_as_float(ra0[j] << 16) + __uint_as_float(ra1[j] << 16);
          dbv[2 * j + 1] += __uint_as_float(ra0[j] & 0xffff0000u) + __uint_as_float(ra1[j] & 0xffff0000u);
        }
      }
    };
    // {x(k, r), x(k + 1, r)} per row: low half = even k
    auto interleave = [](const u32x4& e, const u32x4& o, u32x4& lo, u32x4& hi) {
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        lo[2 * j] = (e[j] & 0xffffu) | (o[j] << 16), lo[2 * j + 1] = (e[j] >> 16) | (o[j] & 0xffff0000u);
        hi[2 * j] = (e[j + 2] & 0xffffu) | (o[j + 2] << 16), hi[2 * j + 1] = (e[j + 2] >> 16) | (o[j + 2] & 0xffff0000u);
      }
    };
    auto store = [&](int buf) {
      if (a_job) {
        u32x4 lo, hi;
        interleave(ra0, ra1, lo, hi);
        uint32_t* d = As + buf * A_PLANE + a_kp * BM + 8 * a_io;
        *reinterpret_cast<u32x4*>(d) = lo, *reinterpret_cast<u32x4*>(d + 4) = hi;
      }
      if (b_job) {
        if (B16) {
          u32x4 lo, hi;
          interleave(rb0, rb1, lo, hi);
          uint32_t* d = Bs + buf * B_PLANE + b_kp * BN + 8 * b_io;
          *reinterpret_cast<u32x4*>(d) = lo, *reinterpret_cast<u32x4*>(d + 4) = hi;
        } else {
          u32x4 v;
#pragma unroll
          for (int j = 0; j < 4; ++j) v[j] = pk_bf16(__uint_as_float(rb0[j]), __uint_as_float(rb1[j]));
          *reinterpret_cast<u32x4*>(Bs + buf * B_PLANE + b_kp * BN + 4 * b_io) = v;
        }
      }
    };
    if (n_slabs > 0) {
      load(0);
      store(0);
    }
    __syncthreads();
    for (int s = 0; s < n_slabs; ++s) {
      const int cur = s & 1;
      if (s + 1 < n_slabs) load(s + 1);                                // in flight under the MFMAs below
      const uint32_t* a = As + cur * A_PLANE;
      const uint32_t* b = Bs + cur * B_PLANE;
#pragma unroll
      for (int m = 0; m < 2; ++m) {
        bf16x8 pa[TM], pb[TN];
#pragma unroll
        for (int t = 0; t < TM; ++t) {
          u32x4 v;
#pragma unroll
          for (int i = 0; i < 4; ++i) v[i] = a[(8 * m + 4 * h + i) * BM + wm * WM + t * 32 + l31];
          pa[t] = __builtin_bit_cast(bf16x8, v);
        }
#pragma unroll
        for (int t = 0; t < TN; ++t) {
          u32x4 v;
#pragma unroll
          for (int i = 0; i < 4; ++i) v[i] = b[(8 * m + 4 * h + i) * BN + wn * WN + t * 32 + l31];
          pb[t] = __builtin_bit_cast(bf16x8, v);
        }
#pragma unroll
        for (int tm = 0; tm < TM; ++tm)
#pragma unroll
          for (int tn = 0; tn < TN; ++tn)
            acc[tm][tn] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(pa[tm], pb[tn], acc[tm][tn], 0, 0, 0);
      }
      if (s + 1 < n_slabs) store(cur ^ 1);
      __syncthreads();
    }
    if (do_db) {                                                       // workgroup-uniform; the planes are free
      if (a_job) {
#pragma unroll
        for (int e = 0; e < 8; ++e) smem[a_kp * BM + 8 * a_io + e] = dbv[e];
      }
      __syncthreads();
      if (tid < BM) {
        float sum = 0.0f;
#pragma unroll
        for (int kp = 0; kp < 16; ++kp) sum += smem[kp * BM + tid];
        dbsum = sum;
      }
      __syncthreads();
    }
  }
};

// One workgroup's tile.  `wg` = tile index inside the problem (j fastest), `bz` = net + nets * split; both come from
// xcd_tile_of.
// WAVES_M: the four waves tile the workgroup's output as WAVES_M x (4 / WAVES_M); 2 x 2 by default, 1 x 4 for wide flat
// tiles (64 x 256: every wave keeps the 64x64 shape of the 128x128 configuration).
template <int BM, int BN, bool A_KC, bool B_KC, int EPI, int BKT = BK, int PREC = 0, int WAVES_M = 2>
__device__ __forceinline__ void gemm_body(const Params& p, const int wg, const int bz, float* __restrict__ smem) {
  constexpr int BK = BKT;                      // shadows gemm::BK inside the kernel
  constexpr int KC_STRIDE = BK + 4;            // floats, K-contig LDS row stride (80 B / 144 B: conflict-free b128 READS, see the header)
  constexpr int KQ = BK / 4;                   // float4 per K-contig row of a slab
  constexpr int WAVES_N = 4 / WAVES_M;
  constexpr int WM = BM / WAVES_M, WN = BN / WAVES_N;    // wave tile
  constexpr int TM = WM / 32, TN = WN / 32;              // 32x32 MFMA tiles per wave
  static_assert(WAVES_M == 1 || WAVES_M == 2, "wave layout");
  static_assert(TM >= 1 && TN >= 1, "tile too small for the wave layout");
  constexpr int A_TILE = A_KC ? BM * KC_STRIDE : BK * BM;
  constexpr int B_TILE = B_KC ? BN * KC_STRIDE : BK * BN;
  constexpr int A_LD4 = BM * BK / 4 / 256;   // float4 loads per thread per slab
  constexpr int B_LD4 = BN * BK / 4 / 256;
  static_assert(A_LD4 >= 1 && B_LD4 >= 1, "tile too small for 256 threads");

  float* As = smem;                 // [2][A_TILE]
  float* Bs = smem + 2 * A_TILE;    // [2][B_TILE]

  const int net = bz % p.nets;
  const int split = bz / p.nets;
  const Operands op = p.op[net];   // by value: pointers live in SGPRs for the whole kernel
  const int tiles_j = (p.J + BN - 1) / BN;
  const int i0 = (wg / tiles_j) * BM;
  const int j0 = (wg % tiles_j) * BN;
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int wm = WAVES_M == 2 ? wave >> 1 : 0, wn = WAVES_M == 2 ? wave & 1 : wave;
  const int l31 = lane & 31, h = lane >> 5;

  int k_begin = 0, k_end = p.Kc;
  if (EPI == EPI_PARTIAL) {
    k_begin = split * p.kc_per_split;
    k_end = k_begin + p.kc_per_split < p.Kc ? k_begin + p.kc_per_split : p.Kc;
  }

  float4 ra[A_LD4], rb[B_LD4];
  // K-contiguous staging, 16-k slabs: WHICH row of the tile a thread brings in.  Thread quad f / 4 takes row krow(f / 4) =
  // f / 4 with bits 0 and 2 exchanged, so the 8 consecutive lanes of one ds_write_b128 group hold rows r and r + 4 (80-B row
  // stride: 320 B apart = 16 banks of the 32 a b128 write sees - disjoint) instead of r and r + 1 (80 B apart: the second
  // row's last quad wrapped onto the first row's first, a 2-way conflict on every write: SQ_LDS_BANK_CONFLICT 0.27-0.34 per
  // LDS-active cycle through round 5).  The LDS image, the set of addresses of every global load instruction (a wave still
  // covers 16 whole rows) and therefore every result are unchanged.  GEMM_ROW_SWIZZLE=0: rounds 1-5 (A/B builds).
  auto krow = [](int r) { return (GEMM_ROW_SWIZZLE && KQ == 4) ? ((r & ~5) | ((r & 1) << 2) | ((r >> 2) & 1)) : r; };

  // Interior tiles (every row / column of the tile in bounds: all of them at the BASELINE shapes) take branch-free
  // loads through per-thread pointers that advance by one slab per iteration; edge tiles and the last, partial slab
  // of a contraction keep the guarded loads.  (The guarded form costs a saveexec / branch pair and a zero fill per
  // 16-B load: ~60 issue slots per slab that sat between the barrier and the first ds_read of every slab.)
  const bool interior = (i0 + BM <= p.I) && (j0 + BN <= p.J);
  // address = workgroup-uniform base (SGPR pair, advanced by one slab per iteration) + per-thread 32-bit byte offset:
  // the `saddr + voffset` form of global_load - one VGPR per 16-B load instead of a 64-bit pointer each
  const char* abase = reinterpret_cast<const char*>(A_KC ? op.A + (int64_t)i0 * p.lda : op.A + i0);
  const char* bbase = reinterpret_cast<const char*>(B_KC ? op.B + (int64_t)j0 * p.ldb : op.B + j0);
  uint32_t aoff[A_LD4], boff[B_LD4];
#pragma unroll
  for (int q = 0; q < A_LD4; ++q) {
    const int f = tid + q * 256;
    aoff[q] = 4u * (A_KC ? (uint32_t)krow(f / KQ) * (uint32_t)p.lda + 4u * (f % KQ)
                         : (uint32_t)(f / (BM / 4)) * (uint32_t)p.lda + 4u * (f % (BM / 4)));
  }
#pragma unroll
  for (int q = 0; q < B_LD4; ++q) {
    const int f = tid + q * 256;
    boff[q] = 4u * (B_KC ? (uint32_t)krow(f / KQ) * (uint32_t)p.ldb + 4u * (f % KQ)
                         : (uint32_t)(f / (BN / 4)) * (uint32_t)p.ldb + 4u * (f % (BN / 4)));
  }
  const int64_t a_step = 4 * (A_KC ? (int64_t)1 : (int64_t)p.lda), b_step = 4 * (B_KC ? (int64_t)1 : (int64_t)p.ldb);   // bytes per unit of k

  auto gload = [&](int k0, auto fast_c) {
    if constexpr (decltype(fast_c)::value) {     // whole tile and whole slab in bounds: straight-line loads
#pragma unroll
      for (int q = 0; q < A_LD4; ++q) ra[q] = *reinterpret_cast<const float4*>(abase + k0 * a_step + aoff[q]);
#pragma unroll
      for (int q = 0; q < B_LD4; ++q) rb[q] = *reinterpret_cast<const float4*>(bbase + k0 * b_step + boff[q]);
      return;
    }
#pragma unroll
    for (int q = 0; q < A_LD4; ++q) {
      const int f = tid + q * 256;
      float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
      if (A_KC) {
        const int r = krow(f / KQ), kq = f % KQ;
        const int gi = i0 + r;
        if (gi < p.I) v = *reinterpret_cast<const float4*>(op.A + (int64_t)gi * p.lda + k0 + 4 * kq);
      } else {
        const int kr = f / (BM / 4), iq = f % (BM / 4);
        const int gk = k0 + kr, gi = i0 + 4 * iq;
        if (gk < k_end && gi + 3 < p.I) v = *reinterpret_cast<const float4*>(op.A + (int64_t)gk * p.lda + gi);
      }
      ra[q] = v;
    }
#pragma unroll
    for (int q = 0; q < B_LD4; ++q) {
      const int f = tid + q * 256;
      float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
      if (B_KC) {
        const int r = krow(f / KQ), kq = f % KQ;
        const int gj = j0 + r;
        if (gj < p.J) v = *reinterpret_cast<const float4*>(op.B + (int64_t)gj * p.ldb + k0 + 4 * kq);
      } else {
        const int kr = f / (BN / 4), jq = f % (BN / 4);
        const int gk = k0 + kr, gj = j0 + 4 * jq;
        if (gk < k_end && gj + 3 < p.J) v = *reinterpret_cast<const float4*>(op.B + (int64_t)gk * p.ldb + gj);
      }
      rb[q] = v;
    }
  };

  auto lstore = [&](int buf) {
    float* a = As + buf * A_TILE;
    float* b = Bs + buf * B_TILE;
#pragma unroll
    for (int q = 0; q < A_LD4; ++q) {
      const int f = tid + q * 256;
      if (A_KC) {
        const int r = krow(f / KQ), kq = f % KQ;
        *reinterpret_cast<float4*>(a + r * KC_STRIDE + 4 * kq) = ra[q];
      } else {
        const int kr = f / (BM / 4), iq = f % (BM / 4);
        *reinterpret_cast<float4*>(a + kr * BM + 4 * iq) = ra[q];
      }
    }
#pragma unroll
    for (int q = 0; q < B_LD4; ++q) {
      const int f = tid + q * 256;
      if (B_KC) {
        const int r = krow(f / KQ), kq = f % KQ;
        *reinterpret_cast<float4*>(b + r * KC_STRIDE + 4 * kq) = rb[q];
      } else {
        const int kr = f / (BN / 4), jq = f % (BN / 4);
        *reinterpret_cast<float4*>(b + kr * BN + 4 * jq) = rb[q];
      }
    }
  };

  f32x16 acc[TM][TN];
#pragma unroll
  for (int a = 0; a < TM; ++a)
#pragma unroll
    for (int b = 0; b < TN; ++b)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.0f;

  float dbsum = 0.0f;  // EPI_PARTIAL: column sum of A for output row i0+tid (tid < BM)
  constexpr int DBC = (EPI == EPI_PARTIAL && !A_KC) ? 256 / BM : 1, DBU = (EPI == EPI_PARTIAL && !A_KC) ? 4 / DBC : 1;   // column copies, quarters per thread
  static_assert(!(EPI == EPI_PARTIAL && !A_KC) || BM == 64 || BM == 128, "bias-gradient quarters: 64- or 128-row weight-gradient tiles");
  [[maybe_unused]] float dbp[DBU] = {};
  [[maybe_unused]] const int db_col = tid % BM, db_c = tid / BM;
  const bool do_db = (EPI == EPI_PARTIAL) && op.dbias != nullptr && j0 == 0;

  const int n_slabs = (k_end - k_begin + BK - 1) / BK;
  // bf16 / split-bf16 operands: planes split at staging time, their own slab loop (PkLoop above)
  // Measured (profiles/r5_ab_packed_*.txt, interleaved A/B against the per-use conversion): split-bf16 pair launch 53.6 ->
  // 49.9 us, 128x128 forward 27.0 -> 25.1 us, minibatch group 209.2 -> 200.5 us.  NOT used for (a) plain bf16 operands:
  // one v_cvt_pk per two elements was never the bound there - the bf16 pair launch moves ~185 MB in 36.8 us = 5.0 TB/s, it
  // sits on the HBM roofline, and the planes' two-row loads cost it 1.4 us; (b) 64-row I-contiguous tiles (the first
  // layer's latency-bound weight gradient: only half the threads have a (k pair, row quad) job, 13.0 -> 19.0 us).
  constexpr bool kPacked = PREC == 2 && GEMM_PACKED && BK == 16 && (A_KC || BM >= 128) && (B_KC || BN >= 128);
  // bf16-STORED operands (act16, see IILoop16): the weight-gradient loop of its own / the K-contiguous loop on reinterpreted slabs
  constexpr bool kSrc16II = (PREC == 3 || PREC == 5) && !A_KC && !B_KC;
  constexpr bool kSrc16KK = (PREC == 3 || PREC == 6) && A_KC && B_KC;
  static_assert(PREC <= 2 || PREC == 4 || kSrc16II || kSrc16KK, "operand layouts of the bf16-stored modes");
  static_assert(PREC <= 2 || BK == 16, "bf16-stored modes: 16-float (32-k) slabs");
  constexpr bool OUT16 = (PREC == 3 || PREC == 4) && (EPI == EPI_BIAS_ELU || EPI == EPI_MUL_DELU);     // C (and aux) stored as bf16
  if constexpr (!kPacked && !kSrc16II) {
    if (n_slabs > 0) {
      gload(k_begin, std::false_type{});
      lstore(0);
    }
    __syncthreads();
  }

  // data gradient: the activation the epilogue multiplies with does not depend on the contraction; with one
  // accumulator tile per wave (16 values per lane) fetch it now so its HBM latency hides behind the main loop
  constexpr bool AUX_EARLY = EPI == EPI_MUL_DELU && TM * TN <= 2 && PREC != 3;
  float auxv[AUX_EARLY ? TM * TN : 1][16];
  if constexpr (AUX_EARLY) {
#pragma unroll
    for (int tm = 0; tm < TM; ++tm)
#pragma unroll
      for (int tn = 0; tn < TN; ++tn) {
        const int gj = j0 + wn * WN + tn * 32 + l31;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int gi = i0 + wm * WM + tm * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
          auxv[tm * TN + tn][r] = (gi < p.I && gj < p.J) ? op.aux[(int64_t)gi * p.ldaux + gj] : 0.0f;
        }
      }
  }

#ifdef GEMM_TIMELINE
  if (threadIdx.x == 0) {
    g_tl[(blockIdx.x + gridDim.x * blockIdx.z) * 4 + 0] = wall_clock64();
    g_tl[(blockIdx.x + gridDim.x * blockIdx.z) * 4 + 1] = clock64();
  }
#endif
  // Two copies of the main loop: interior tiles whose contraction range is a whole number of slabs run one with
  // straight-line global loads (no guard, no branch: the compiler otherwise merges the guarded and the unguarded load
  // sequences into one CFG and drains vmcnt between them); everything else runs the guarded copy.
  auto main_loop = [&](auto fast_c) {
  for (int s = 0; s < n_slabs; ++s) {
    const int cur = s & 1;
    constexpr bool kPiped = PREC == 0 && BK == 16 && GEMM_PIPE && A_KC && B_KC;     // see the pipelined slab below
    if (!kPiped && s + 1 < n_slabs) gload(k_begin + (s + 1) * BK, fast_c);
    const float* a = As + cur * A_TILE;
    const float* b = Bs + cur * B_TILE;

    if constexpr (EPI == EPI_PARTIAL && !A_KC) {
      if (do_db) {                                 // workgroup-uniform
        // Bias gradient = column sums of A over the split.  Round 6: ALL 256 threads take part - thread (column db_col, copy
        // db_c) sums its quarter(s) of the slab's k rows into its own running partial(s) - instead of the first BM threads
        // walking all BK rows while the other waves wait for them at the slab's barrier (16 dependent LDS reads + adds per
        // slab on the critical path of a workgroup whose matrix work is 8 instructions per wave).  The four quarter sums are
        // combined in a fixed order behind the loop: (q0 + q1) + (q2 + q3), the same for every tile height.  Update phase 8.957 ->
        // 8.936 ms at cfg2, 11.327 -> 11.266 ms at the reference shapes (three interleaved rounds, profiles/r6_ab_db_spread.txt).
#pragma unroll
        for (int u = 0; u < DBU; ++u) {
          const int g = db_c * DBU + u;
#pragma unroll
          for (int kk = 0; kk < BK / 4; ++kk) dbp[u] += a[(g * (BK / 4) + kk) * BM + db_col];
        }
      }
    }

    // fragment of 4 consecutive k (k = 8*blk + 4*h + 0..3) of operand rows `row` for every 32-row tile
    auto frag_a = [&](int blk, float (&af)[TM][4]) {
#pragma unroll
      for (int t = 0; t < TM; ++t) {
        const int row = wm * WM + t * 32 + l31;
        if (A_KC) {
          const float4 v = *reinterpret_cast<const float4*>(a + row * KC_STRIDE + 8 * blk + 4 * h);
          af[t][0] = v.x, af[t][1] = v.y, af[t][2] = v.z, af[t][3] = v.w;
        } else {
#pragma unroll
          for (int q = 0; q < 4; ++q) af[t][q] = a[(8 * blk + 4 * h + q) * BM + row];
        }
      }
    };
    auto frag_b = [&](int blk, float (&bf)[TN][4]) {
#pragma unroll
      for (int t = 0; t < TN; ++t) {
        const int col = wn * WN + t * 32 + l31;
        if (B_KC) {
          const float4 v = *reinterpret_cast<const float4*>(b + col * KC_STRIDE + 8 * blk + 4 * h);
          bf[t][0] = v.x, bf[t][1] = v.y, bf[t][2] = v.z, bf[t][3] = v.w;
        } else {
#pragma unroll
          for (int q = 0; q < 4; ++q) bf[t][q] = b[(8 * blk + 4 * h + q) * BN + col];
        }
      }
    };
    if constexpr (kPiped) {
      // Software-pipelined slab (fp32 MFMA, 16-k slabs).  v_mfma_f32_32x32x2_f32 occupies the matrix pipe for 64
      // cycles, so ~15 other instructions issue for free behind each one - but only if they sit BETWEEN the MFMAs in
      // program order (a wave issues in order).  The round-1/2 loop ran   barrier -> [guarded global loads of the next
      // slab: ~60 slots] -> 8 ds_read -> wait -> 32 MFMA -> vmcnt(0) -> 4 ds_write -> lgkmcnt(0) -> barrier : everything
      // outside the MFMA block is a gap in which this wave feeds nothing to the pipe (~800 of ~2900 cycles per slab;
      // the co-resident workgroup only covers it when the two happen to be out of phase).  Here:
      //   barrier -> 8 ds_read (whole slab) -> global loads (issue while the reads are in flight) -> 24 MFMA
      //           -> vmcnt(0) + 4 ds_write into the OTHER buffer -> 8 MFMA (cover the write latency) -> barrier
      // What stays exposed per slab: the barrier skew and one LDS read latency.
      float af[2][TM][4], bf[2][TN][4];
      frag_a(0, af[0]);
      frag_b(0, bf[0]);
      frag_a(1, af[1]);
      frag_b(1, bf[1]);
      __builtin_amdgcn_sched_barrier(0);
      if (s + 1 < n_slabs) gload(k_begin + (s + 1) * BK, fast_c);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int q = 0; q < 4; ++q)
#pragma unroll
        for (int tm = 0; tm < TM; ++tm)
#pragma unroll
          for (int tn = 0; tn < TN; ++tn)
            acc[tm][tn] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[0][tm][q], bf[0][tn][q], acc[tm][tn], 0, 0, 0);
#pragma unroll
      for (int q = 0; q < 2; ++q)
#pragma unroll
        for (int tm = 0; tm < TM; ++tm)
#pragma unroll
          for (int tn = 0; tn < TN; ++tn)
            acc[tm][tn] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[1][tm][q], bf[1][tn][q], acc[tm][tn], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
      if (s + 1 < n_slabs) lstore(cur ^ 1);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int q = 2; q < 4; ++q)
#pragma unroll
        for (int tm = 0; tm < TM; ++tm)
#pragma unroll
          for (int tn = 0; tn < TN; ++tn)
            acc[tm][tn] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[1][tm][q], bf[1][tn][q], acc[tm][tn], 0, 0, 0);
      __syncthreads();
      continue;
    } else if constexpr (kSrc16KK) {
      // bf16-stored K-contiguous operands: a lane's float4 IS the 8 bf16 (k = 16 blk + 8 h + e) of one MFMA operand
      using f32x4 = __attribute__((ext_vector_type(4))) float;
#pragma unroll
      for (int blk = 0; blk < BK / 8; ++blk) {
        float af[TM][4], bf[TN][4];
        frag_a(blk, af);
        frag_b(blk, bf);
        bf16x8 pa[TM], pb[TN];
#pragma unroll
        for (int t = 0; t < TM; ++t) pa[t] = __builtin_bit_cast(bf16x8, f32x4{af[t][0], af[t][1], af[t][2], af[t][3]});
#pragma unroll
        for (int t = 0; t < TN; ++t) pb[t] = __builtin_bit_cast(bf16x8, f32x4{bf[t][0], bf[t][1], bf[t][2], bf[t][3]});
#pragma unroll
        for (int tm = 0; tm < TM; ++tm)
#pragma unroll
          for (int tn = 0; tn < TN; ++tn)
            acc[tm][tn] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(pa[tm], pb[tn], acc[tm][tn], 0, 0, 0);
      }
    } else if constexpr (PREC == 0) {
#pragma unroll
      for (int blk = 0; blk < BK / 8; ++blk) {
        float af[TM][4], bf[TN][4];
        frag_a(blk, af);
        frag_b(blk, bf);
#pragma unroll
        for (int q = 0; q < 4; ++q)
#pragma unroll
          for (int tm = 0; tm < TM; ++tm)
#pragma unroll
            for (int tn = 0; tn < TN; ++tn)
              acc[tm][tn] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[tm][q], bf[tn][q], acc[tm][tn], 0, 0, 0);
      }
    } else {
      // bf16 operands, fp32 accumulation: the same fp32 LDS images, rounded to bf16 (RNE, v_cvt_pk_bf16_f32) on
      // the way into ONE v_mfma_f32_32x32x16_bf16 per 16 k.  Slot (h, i) of the instruction carries
      // k = 8*(i/4) + 4*h + i%4 for A and B alike, so the contraction is complete and each product exact.
      static_assert(PREC == 0 || BK % 16 == 0, "bf16 paths consume 16 k per MFMA");
#pragma unroll
      for (int kb = 0; kb < BK / 16; ++kb) {
        float a0[TM][4], a1[TM][4], b0[TN][4], b1[TN][4];
        frag_a(2 * kb, a0);
        frag_a(2 * kb + 1, a1);
        frag_b(2 * kb, b0);
        frag_b(2 * kb + 1, b1);
        bf16x8 pa[TM], pb[TN];
#pragma unroll
        for (int t = 0; t < TM; ++t)
#pragma unroll
          for (int q = 0; q < 4; ++q) pa[t][q] = (__bf16)a0[t][q], pa[t][4 + q] = (__bf16)a1[t][q];
#pragma unroll
        for (int t = 0; t < TN; ++t)
#pragma unroll
          for (int q = 0; q < 4; ++q) pb[t][q] = (__bf16)b0[t][q], pb[t][4 + q] = (__bf16)b1[t][q];
        if constexpr (PREC == 2) {
          // split-bf16 ("bf16x3"): x = hi + lo with hi = bf16(x), lo = bf16(x - hi): 16 mantissa bits per operand.
          //   a.b ~= a_hi.b_hi + a_hi.b_lo + a_lo.b_hi        (the dropped a_lo.b_lo term is 2^-16 relative)
          // three v_mfma_f32_32x32x16_bf16 (3 x 8 passes) replace eight v_mfma_f32_32x32x2_f32 (8 x 16 passes) per
          // 16 k: 5.3x less matrix-pipe time at ~16-bit operand precision - finer than the TF32 (10-bit) arithmetic the
          // reference itself enables for these GEMMs (scripts/clean_rl/train.py:86-87).  Small terms first.
          bf16x8 la[TM], lb[TN];
#pragma unroll
          for (int t = 0; t < TM; ++t)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
              la[t][q] = (__bf16)(a0[t][q] - (float)pa[t][q]);
              la[t][4 + q] = (__bf16)(a1[t][q] - (float)pa[t][4 + q]);
            }
#pragma unroll
          for (int t = 0; t < TN; ++t)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
              lb[t][q] = (__bf16)(b0[t][q] - (float)pb[t][q]);
              lb[t][4 + q] = (__bf16)(b1[t][q] - (float)pb[t][4 + q]);
            }
#pragma unroll
          for (int tm = 0; tm < TM; ++tm)
#pragma unroll
            for (int tn = 0; tn < TN; ++tn) {
              acc[tm][tn] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(la[tm], pb[tn], acc[tm][tn], 0, 0, 0);
              acc[tm][tn] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(pa[tm], lb[tn], acc[tm][tn], 0, 0, 0);
            }
        }
#pragma unroll
        for (int tm = 0; tm < TM; ++tm)
#pragma unroll
          for (int tn = 0; tn < TN; ++tn)
            acc[tm][tn] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(pa[tm], pb[tn], acc[tm][tn], 0, 0, 0);
      }
    }

    if (s + 1 < n_slabs) lstore(cur ^ 1);
    __syncthreads();
  }
  };
  constexpr bool kPipedLoop = PREC == 0 && BK == 16 && GEMM_PIPE && A_KC && B_KC;
  if constexpr (kPacked) {
    PkLoop<BM, BN, A_KC, B_KC, EPI, PREC, WAVES_M>::run(p, op, i0, j0, k_begin, k_end, do_db, smem, acc, dbsum);
  } else if constexpr (kSrc16II) {
    IILoop16<BM, BN, EPI, PREC == 3, WAVES_M>::run(p, op, i0, j0, k_begin, k_end, do_db, smem, acc, dbsum);
  } else {
    if (kPipedLoop && interior && (k_end - k_begin) % BK == 0) main_loop(std::true_type{});
    else main_loop(std::false_type{});
    if constexpr (EPI == EPI_PARTIAL && !A_KC) {
      if (do_db) {                                 // workgroup-uniform; the slab buffers are free behind the loop's last barrier
#pragma unroll
        for (int u = 0; u < DBU; ++u) smem[(db_c * DBU + u) * BM + db_col] = dbp[u];
        __syncthreads();
        if (tid < BM) dbsum = (smem[tid] + smem[BM + tid]) + (smem[2 * BM + tid] + smem[3 * BM + tid]);
        __syncthreads();
      }
    }
  }

#ifdef GEMM_TIMELINE
  if (threadIdx.x == 0) {
    g_tl[(blockIdx.x + gridDim.x * blockIdx.z) * 4 + 2] = clock64();
    g_tl[(blockIdx.x + gridDim.x * blockIdx.z) * 4 + 3] = wall_clock64();
  }
#endif
  // ---- epilogue.  acc[tm][tn][r] of lane: row = (r&3) + 8*(r>>2) + 4*h, col = l31 ----------
  if constexpr (EPI == EPI_BIAS_ELU_LDS) {
    // the activated tile stays on chip, over the slab buffers (free after the loop's last barrier); the caller
    // synchronises before reading it.  Rows past p.I hold elu(bias): the consumer masks them.
    constexpr int LD = BN + kLdsTilePad;
#pragma unroll
    for (int tm = 0; tm < TM; ++tm)
#pragma unroll
      for (int tn = 0; tn < TN; ++tn) {
        const int cj = wn * WN + tn * 32 + l31;
        const float bias = j0 + cj < p.J ? op.bias[j0 + cj] : 0.0f;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int lrow = wm * WM + tm * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
          smem[lrow * LD + cj] = elu_f(acc[tm][tn][r] + bias);
        }
      }
    return;
  }
  float* Cout = op.C;
  if (EPI == EPI_PARTIAL) Cout += (int64_t)split * p.c_split_stride;
  // Activation-sized outputs (forward, data gradient) leave through LDS: a lane owns one COLUMN of a 32x32
  // accumulator tile, so storing it directly takes 16 dword stores per tile and the store tail of a launch is
  // issue bound (MI355X guide: ~7 B/clk/CU).  Each wave transposes its tile through a private 32x36-float LDS patch
  // (the slab buffers are free after the main loop's last barrier) and writes rows with 4 dwordx4 stores instead.
  // Round 6: the split-K partials (EPI_PARTIAL) leave the same way.  Their 16 single-dword write-through stores per lane
  // and tile were the slow form of store on this chip (MI355X guide: a scalar sc1 store is one fabric write, ~6x the time
  // per byte of a dwordx4) - 15.7 MB of them per 2048-row optimiser step.  GEMM_STAGE_PARTIAL=0: the dword stores (A/B).
  constexpr int STG_LD = 36;                                     // 144-B rows: 16-B aligned, conflict-free b128 reads
  constexpr int QUART = (2 * (A_TILE + B_TILE)) / 4 > 32 * STG_LD ? (2 * (A_TILE + B_TILE)) / 4 : 32 * STG_LD;
  constexpr bool STAGED = EPI != EPI_PARTIAL || GEMM_STAGE_PARTIAL;      // (smem_bytes() guarantees four patches)
  float* stg = smem + wave * QUART;                              // this wave's patch inside the (free) slab buffers
#pragma unroll
  for (int tm = 0; tm < TM; ++tm) {
#pragma unroll
    for (int tn = 0; tn < TN; ++tn) {
      const int gj = j0 + wn * WN + tn * 32 + l31;
      float bias = 0.0f;
      if (EPI == EPI_BIAS_ELU) bias = gj < p.J ? op.bias[gj] : 0.0f;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int lrow = (r & 3) + 8 * (r >> 2) + 4 * h;
        const int gi = i0 + wm * WM + tm * 32 + lrow;
        const bool inb = gi < p.I && gj < p.J;
        float v = acc[tm][tn][r];
        if (EPI == EPI_BIAS_ELU) {
#ifdef GEMM_PROBE_NOELU
          v = v + bias;
#else
          v = elu_f(v + bias);
#endif
        } else if (EPI == EPI_MUL_DELU && PREC != 3) {   // (bf16-stored aux: multiplied in the row phase below, 8 columns per 16-B load)
          float hact;
          if constexpr (AUX_EARLY) {
            hact = auxv[tm * TN + tn][r];
          } else {
            hact = inb ? op.aux[(int64_t)gi * p.ldaux + gj] : 0.0f;
          }
          v = v * (hact > 0.0f ? 1.0f : hact + 1.0f);  // elu'(z) = 1 (z>0) | exp(z) = elu(z)+1
        }
        if constexpr (STAGED) {
          stg[lrow * STG_LD + l31] = v;
        } else {
#ifdef GEMM_PROBE_NOSTORE
          if (v == 12345.678f)
#endif
          if (inb) {
            float* dstp = Cout + (int64_t)gi * p.ldc + gj;
            if constexpr (EPI == EPI_PARTIAL) {   // split-K partials: 16.8 MB per launch, re-read by the fold launch
              asm volatile("global_store_dword %0, %1, off sc1\n\ts_nop 0" ::"v"(dstp), "v"(v) : "memory");
            } else {
              *dstp = v;
            }
          }
        }
      }
      if constexpr (STAGED && OUT16) {
        // bf16-stored output: a lane owns 8 consecutive columns of a row (two b128 reads of the patch, one 16-B store);
        // 4 lanes = one 64-B row segment, 16 rows per instruction.  Data gradient: the aux activations (bf16-stored) arrive
        // the same way and elu' is applied here.
        __builtin_amdgcn_wave_barrier();
        const int c8 = 8 * (lane & 3);
        const int gj8 = j0 + wn * WN + tn * 32 + c8;
        uint16_t* C16 = reinterpret_cast<uint16_t*>(Cout);
#pragma unroll
        for (int it = 0; it < 2; ++it) {
          const int lrow = it * 16 + (lane >> 2);
          const int gi = i0 + wm * WM + tm * 32 + lrow;
          const float4 q0 = *reinterpret_cast<const float4*>(stg + lrow * STG_LD + c8);
          const float4 q1 = *reinterpret_cast<const float4*>(stg + lrow * STG_LD + c8 + 4);
          float o[8] = {q0.x, q0.y, q0.z, q0.w, q1.x, q1.y, q1.z, q1.w};
          if (gi < p.I && gj8 + 7 < p.J) {
            if constexpr (EPI == EPI_MUL_DELU) {
              const u32x4 hx = *reinterpret_cast<const u32x4*>(reinterpret_cast<const uint16_t*>(op.aux) + (int64_t)gi * p.ldaux + gj8);
#pragma unroll
              for (int j = 0; j < 4; ++j) {
                const float h0 = __uint_as_float(hx[j] << 16), h1 = __uint_as_float(hx[j] & 0xffff0000u);
                o[2 * j] *= h0 > 0.0f ? 1.0f : h0 + 1.0f;
                o[2 * j + 1] *= h1 > 0.0f ? 1.0f : h1 + 1.0f;
              }
            }
            u32x4 pk;
#pragma unroll
            for (int j = 0; j < 4; ++j) pk[j] = pk_bf16(o[2 * j], o[2 * j + 1]);
            uint16_t* dstp = C16 + (int64_t)gi * p.ldc + gj8;
            asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 2" ::"v"(dstp), "v"(pk) : "memory");
          }
        }
        __builtin_amdgcn_wave_barrier();
      } else if constexpr (STAGED) {
        __builtin_amdgcn_wave_barrier();                         // LDS ops of a wave execute in order
        const int gj4 = j0 + wn * WN + tn * 32 + 4 * (lane & 7);  // 8 lanes x 16 B = one 128-B row segment
#pragma unroll
        for (int it = 0; it < 4; ++it) {
          const int lrow = it * 8 + (lane >> 3);
          const int gi = i0 + wm * WM + tm * 32 + lrow;
          const float4 q = *reinterpret_cast<const float4*>(stg + lrow * STG_LD + 4 * (lane & 7));
#ifdef GEMM_PROBE_NOSTORE
          if (q.x == 12345.678f)
#endif
          // write-through (non-temporal) row stores: tens of MB left dirty in L2 would be flushed at the kernel
          // boundary (MI355X guide, "boundary": + bytes / 6 TB/s) - stream them out while the tile is still computing
          if (gi < p.I && gj4 + 3 < p.J) {
            using f4v = __attribute__((ext_vector_type(4))) float;
            f4v o;
            o.x = q.x, o.y = q.y, o.z = q.z, o.w = q.w;
            float* dstp = Cout + (int64_t)gi * p.ldc + gj4;
            // (the compiler's hazard recognizer does not see a store inside inline asm: a VALU write of the four data
            // VGPRs right behind a >64-bit store needs a wait state, hence the s_nop)
            asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 2" ::"v"(dstp), "v"(o) : "memory");
          }
        }
        __builtin_amdgcn_wave_barrier();
      }
    }
  }
  if (EPI == EPI_PARTIAL) {
    if (do_db && tid < BM && i0 + tid < p.I) op.dbias[(int64_t)split * p.I + i0 + tid] = dbsum;
  }
}

template <int BM, int BN, bool A_KC, bool B_KC, int EPI, int BKT = BK, int PREC = 0>
__global__ __launch_bounds__(256) void gemm_f32_kernel(const Params p) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const gemm::TileId t = xcd_tile_of(blockIdx.x + gridDim.x * blockIdx.z, gridDim.x, gridDim.z, p.xcd_legacy);
  gemm_body<BM, BN, A_KC, B_KC, EPI, BKT, PREC>(p, t.tile, t.bz, smem);
}

// Two independent problems in ONE launch: the first n0 workgroups (in launch order) run problem 0, the rest
// problem 1.  Used for a layer's weight gradient (few long split-K workgroups, one wave per SIMD when alone on a
// CU) together with its data gradient (many short workgroups): the short ones fill the issue slots the long
// ones leave idle, and one launch boundary disappears.
template <int BM0, int BN0, bool A_KC0, bool B_KC0, int EPI0, int BM1, int BN1, bool A_KC1, bool B_KC1, int EPI1,
          int PREC = 0>
__global__ __launch_bounds__(256) void gemm_pair_kernel(const Params p0, const Params p1, const int tiles0,
                                                        const int n0, const int tiles1) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int b = blockIdx.x;
  if (b < n0) {
    const TileId t = xcd_tile_of(b, tiles0, n0 / tiles0, p0.xcd_legacy);
    gemm_body<BM0, BN0, A_KC0, B_KC0, EPI0, BK, PREC>(p0, t.tile, t.bz, smem);
  } else {
    // the XCD of workgroup b is b % 8 = (c + n0) % 8: grouping by c % 8 is the same partition for any n0
    const TileId t = xcd_tile_of(b - n0, tiles1, (gridDim.x - n0) / tiles1, p1.xcd_legacy);
    gemm_body<BM1, BN1, A_KC1, B_KC1, EPI1, BK, PREC>(p1, t.tile, t.bz, smem);
  }
}

template <int BM, int BN, bool A_KC, bool B_KC, int BKT = BK>
constexpr size_t smem_bytes() {
  constexpr size_t slabs = sizeof(float) * 2 *
         ((A_KC ? BM * (BKT + 4) : BKT * BM) + (B_KC ? BN * (BKT + 4) : BKT * BN));
  constexpr size_t patches = sizeof(float) * 4 * 32 * 36;       // the epilogue's four 32 x 36 transposition patches
  return slabs > patches ? slabs : patches;
}

}  // namespace gemm
