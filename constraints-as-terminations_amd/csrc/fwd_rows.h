// Row-resident forward of the hidden layers (round 4): a workgroup keeps the activations of R = 64 (update phase) or
// 32 (rollout) rows in ONE LDS tile from layer to layer, in place, and streams the weights.
//
// Replaces, per minibatch of the update phase, the first-layer forward launch and the 128x128-tile forward launch(es) of
// cleanrl/ppo.py:78-96,104-119 (13.8 + 41.1 us at cfg2, 0.37 / 0.66 of the fp32-MFMA peak: each a single round of
// workgroups whose prologue burst, 33 MB store tail and launch boundary nothing overlaps) by one launch in which the
// activation stores of layer l run under the contraction of layer l+1.
//
// What differs from round 3's fused_fwd_kernel (32 rows, two ping-pong tiles, three-slot rings of 16-k slabs), and why:
//   * 64 rows per workgroup, two accumulators per wave: the weight stream (the same bytes for any number of rows) is
//     amortised over twice the MFMAs.  Round 3 measured that kernel's slab loop at 7.3 us per layer with the MFMAs
//     REMOVED against 7.4 us of pure matrix-pipe time: the weight path (16 B/clk/CU of L2 hits, every CU asking for the
//     same lines at the same time) was as long as the arithmetic it was supposed to hide behind;
//   * weights are requested in full 128-byte lines: a load instruction covers 8 rows x 32 k (8 lanes x 16 B per row)
//     instead of 16 rows x 16 k (64-byte half lines whose other half the NEXT slab fetched again through a 32 KB L1
//     that eight waves x 32 rows of 128-B lines exactly fill);
//   * ONE activation tile, overwritten in place behind a barrier once every wave holds the layer's outputs in its
//     accumulators (66.5 KB instead of 133 KB for 64 rows x 256): the rings of 32-k slabs fit beside it.
// Contraction order per output element = gemm_body's K-contiguous order (k = 32 s + 8 blk + 4 h + q, ascending), one
// fp32 MFMA chain per element: bit-identical to the layer-wise launches.
//
// Layout in LDS: tile [R][ld] (ld = widest activation + 4 floats: rows 4 banks apart => conflict-free ds_read_b128 by 16
// rows), then eight wave-private rings of two slots [32 weight rows][36] (144-B rows: conflict-free b128 reads by row,
// ds_write_b128 of 8 lanes = one whole 128-B row segment).  A wave owns the 32 output columns [32 w, 32 w + 32) of the
// 256-column layer; the only cross-wave traffic is the activation tile (two barriers per layer).
#pragma once

namespace rowsfwd {

using gemm::f32x16;

constexpr int kThreads = 512;              // eight waves, two per SIMD
constexpr int kWS = 36;                    // floats per weight row in a ring slot: 32 k + 4 pad
constexpr int kSlot = 32 * kWS;            // one slot: the wave's 32 weight rows x 32 k
constexpr int kRingWave = 2 * kSlot;       // two slots per wave
constexpr int kWidth = 256;                // every layer computed here is 256 wide (8 waves x 32 columns)

template <int R>
constexpr size_t lds_bytes(int ld) { return sizeof(float) * ((size_t)R * ld + 8 * kRingWave); }

// One 256-wide layer for the workgroup's R rows: acc[t] (rows 32 t .. 32 t + 31) = in[:, :K] . W[32 w .. 32 w + 31, :K]^T.
//
// Weight pipeline of a wave (slabs of 32 k; ring slots A / B; ONE staging register set inside the loop):
//     stage()   requests slabs 0 and 1 (two register sets) - issued by the caller in FRONT of whatever separates two
//               layers (barriers, the in-place tile write), so that their latency hides there
//     begin()   slot A <- slab 0, slot B <- slab 1, slab 2 requested
//     loop()    slab s multiplies out of slot s & 1; its last block fetches the first fragments of slab s+1 (other
//               slot, written at least one slab ago); at its end the staged slab s+2 overwrites slot s & 1 (every read
//               of slab s has been issued: LDS operations of a wave execute in order) and slab s+3 is requested.
// Why two slabs deep: the caller streams the PREVIOUS layer's activation tile out to memory between begin() and loop()
// (write-through stores).  gfx9 counts loads and stores in ONE in-order counter (vmcnt), so a wait for a load issued
// after those stores also waits for the stores.  Here the first such load is slab 3, requested at the end of slab 0 and
// needed at the end of slab 1: the stores get two slabs (~3.7 us) to reach memory before anything waits behind them.
// (The first version requested slab 1 behind the stores and needed it half a slab later: every layer stalled on its
// predecessor's store tail, 57 us for what the matrix pipe does in 35.)
template <int R>
struct Layer {
  static constexpr int T = R / 32;
  const float* gp;                 // this lane's float4 of weight row (32 w + lane / 8), k quad lane % 8
  int64_t row8;                    // 8 weight rows further (floats)
  float4 st[4], st1[4];            // staged slabs: rows lane/8 + 8 j  (st1: only between stage() and begin())

  __device__ __forceinline__ void gload(float4 (&d)[4], int s) const {
#pragma unroll
    for (int j = 0; j < 4; ++j) d[j] = *reinterpret_cast<const float4*>(gp + j * row8 + 32 * s);
  }
  __device__ __forceinline__ void lstore(float* slot, const float4 (&d)[4], int lane) const {
    float* p = slot + (lane >> 3) * kWS + 4 * (lane & 7);
#pragma unroll
    for (int j = 0; j < 4; ++j) *reinterpret_cast<float4*>(p + 8 * j * kWS) = d[j];
  }
  __device__ __forceinline__ void stage(const float* W, int K, int wave, int lane) {
    gp = W + (int64_t)(wave * 32 + (lane >> 3)) * K + 4 * (lane & 7);
    row8 = (int64_t)8 * K;
    gload(st, 0);
    gload(st1, 1);       // (a slab past the end of a short contraction reads the following weight rows: never multiplied)
  }
  __device__ __forceinline__ void begin(float* __restrict__ wring, int lane) {
    lstore(wring, st, lane);
    lstore(wring + kSlot, st1, lane);
    gload(st, 2);
  }

  // one 32-k slab (NB = 4 blocks, or the 2-block tail of a contraction that is not a multiple of 32) out of ring slot
  // `bs`; fragments of its block 0 are already in fa[0] / fb[0].  NEXT: another slab follows (its block-0 fragments are
  // requested at the last block, from `bn`).  REFILL: the staged slab s+2 goes into THIS slab's slot `mine` at the end
  // and slab s+3 is requested (a request past the last slab reads the weight rows / the bias that follow: in bounds by
  // the plan's check, never multiplied).  No branch inside: the loop body is straight-line code.
  template <int NB, bool NEXT, bool REFILL>
  // (`bs`, `mine` and `bn` point into the SAME wave-private ring - a REFILL slab reads slot `bs` and then overwrites it
  // through `mine` - so none of them may be __restrict__: the order "last fragment read, then ds_write" is kept by the
  // sched_barriers below and by the in-order LDS issue of a wave, not by an aliasing promise)
  __device__ __forceinline__ void slab(const float* __restrict__ as, const int ld, const float* bs,
                                       float* mine, const float* bn, const int s,
                                       float4 (&fa)[2][T], float4 (&fb)[2], f32x16 (&acc)[T], const int lane) {
#pragma unroll
    for (int blk = 0; blk < NB; ++blk) {
      const int cur = blk & 1, nxt = cur ^ 1;
      // the NEXT block's fragments are requested before this block's MFMAs (the compiler otherwise sinks the reads to
      // just in front of their use: lgkmcnt wait -> MFMA, one exposed LDS latency per block)
      if (blk + 1 < NB) {
#pragma unroll
        for (int t = 0; t < T; ++t) fa[nxt][t] = *reinterpret_cast<const float4*>(as + 32 * t * ld + 8 * (blk + 1));
        fb[nxt] = *reinterpret_cast<const float4*>(bs + 8 * (blk + 1));
      } else if (NEXT) {
#pragma unroll
        for (int t = 0; t < T; ++t) fa[nxt][t] = *reinterpret_cast<const float4*>(as + 32 * t * ld + 32);
        fb[nxt] = *reinterpret_cast<const float4*>(bn);
      }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int t = 0; t < T; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[cur][t].x, fb[cur].x, acc[t], 0, 0, 0);
#pragma unroll
      for (int t = 0; t < T; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[cur][t].y, fb[cur].y, acc[t], 0, 0, 0);
      if (REFILL && blk == NB - 1) {
        // every fragment read of this slab has been issued: its slot takes the staged slab s+2 - between MFMAs, so that
        // the four ds_write_b128 and the four requests of slab s+3 issue behind the matrix pipe
        __builtin_amdgcn_sched_barrier(0);
        lstore(mine, st, lane);
        gload(st, s + 3);
        __builtin_amdgcn_sched_barrier(0);
      }
#pragma unroll
      for (int t = 0; t < T; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[cur][t].z, fb[cur].z, acc[t], 0, 0, 0);
#pragma unroll
      for (int t = 0; t < T; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[cur][t].w, fb[cur].w, acc[t], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
    }
  }

  // ZERO = false: the accumulators continue a contraction begun by an earlier call (fwd_rows_wide.h: a 512-wide input
  // consumed in 256-column chunks - the MFMA chain per element simply goes on, k ascending)
  template <bool ZERO = true>
  __device__ __forceinline__ void loop(const float* __restrict__ tile, const int ld, float* __restrict__ wring,
                                       const int K, f32x16 (&acc)[T], const int lane) {
    const int l31 = lane & 31, h = lane >> 5;
    const int nblk = K >> 3;                         // 8-k blocks of the contraction (K % 16 == 0)
    const int n_slabs = (nblk + 3) >> 2;             // 32-k slabs; the last one may hold 2 blocks
    if (ZERO) {
#pragma unroll
      for (int t = 0; t < T; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.0f;
    }
    const float* ap = tile + l31 * ld + 4 * h;       // A fragments: row l31 (+ 32 t), k = 32 s + 8 blk + 4 h ..
    float* const w0 = wring;
    float* const w1 = wring + kSlot;
    const float* const bp = wring + l31 * kWS + 4 * h;     // B fragments inside a slot
    float4 fa[2][T], fb[2];
#pragma unroll
    for (int t = 0; t < T; ++t) fa[0][t] = *reinterpret_cast<const float4*>(ap + 32 * t * ld);
    fb[0] = *reinterpret_cast<const float4*>(bp);
    int s = 0;
    if (n_slabs > 2) {
      // the first refilling slab stands outside the loop: its wait for the staged slab 2 (requested in begin(), i.e.
      // BEFORE the caller's activation stores) then gets an exact count, vmcnt(<stores>), instead of the vmcnt(0) a
      // wait shared with the later iterations would need - which would stall on those stores
      slab<4, true, true>(ap, ld, bp, w0, bp + kSlot, 0, fa, fb, acc, lane);
      s = 1;
    }
    for (; s + 2 < n_slabs; ++s) {                   // slabs with two successors: refill their slot with slab s+2
      const int o = (s & 1) * kSlot;
      slab<4, true, true>(ap + 32 * s, ld, bp + o, (s & 1) ? w1 : w0, bp + (kSlot - o), s, fa, fb, acc, lane);
    }
    if (s + 1 < n_slabs) {                           // the last but one: a successor, nothing left to stage
      const int o = (s & 1) * kSlot;
      slab<4, true, false>(ap + 32 * s, ld, bp + o, nullptr, bp + (kSlot - o), s, fa, fb, acc, lane);
      ++s;
    }
    const int o = (s & 1) * kSlot;
    if (nblk - 4 * s == 4) slab<4, false, false>(ap + 32 * s, ld, bp + o, nullptr, nullptr, s, fa, fb, acc, lane);
    else slab<2, false, false>(ap + 32 * s, ld, bp + o, nullptr, nullptr, s, fa, fb, acc, lane);
  }
};

}  // namespace rowsfwd
