// Row-resident forward for networks whose layers are NOT all 256 wide (round 5): the reference's own Agent is
// 45 -> 512 -> 256 -> 128 (cleanrl/ppo.py:78-96) and never took round 4's rows_fwd_kernel (fwd_rows.h: every layer 256
// wide, ONE activation tile overwritten in place).  Included by mlp.hip behind rows_fwd_kernel (uses FusedFwdArgs,
// fused_head, rowsfwd::Layer).
//
// What changes against rows_fwd_kernel, and why it still fits the LDS:
//   * a 512-wide first layer is never resident as a whole.  It is PRODUCED AND CONSUMED IN 256-COLUMN CHUNKS: chunk c of
//     H0 = elu(X W0[256 c .. 256 c + 255]^T + b0) is computed from the (small, persistent) observation tile, written to
//     the activation tile, streamed out for the backward, and contracted with W1[:, 256 c .. 256 c + 255] into the
//     layer-1 accumulators, which live across the chunks.  Per output element of layer 1 the contraction still walks
//     k = 0 .. 511 ascending in 32-k slabs - one fp32 MFMA chain, the order of gemm_body - so results stay bit-identical
//     to the layer-wise launches.  LDS: observation tile [R][Dp + 4] (Dp <= 64) + activation tile [R][260] + the eight
//     two-slot weight rings = 13.3 + 66.5 + 73.7 KB at R = 64 (a resident [64][516] tile alone would be 132 KB).
//   * a 128-wide layer occupies waves 0-3 (32 columns each); waves 4-7 skip its contraction and meet the others at the
//     barriers.  Widths are run-time values (wave-uniform branches), the layer COUNT and the chunk count are template
//     parameters: as in rows_fwd_kernel every loop over layers / chunks / networks is fully unrolled so that the
//     compiler can count the loads and stores in flight (s_waitcnt vmcnt(N) with exact N instead of vmcnt(0) behind the
//     activation stores of the previous phase).
//   * the first layer's weights (K = Dp <= 64: at most two 32-k slabs) go through their own staging registers `pre`,
//     requested one phase ahead - in FRONT of the activation stores of the phase before - and copied into the wave's ring
//     right before use; the long contractions keep rowsfwd::Layer's two-slabs-deep pipeline.
// Order of VMEM operations per chunk (what makes every wait hit a load that is OLDER than the stores in flight):
//     stage(L1 chunk c)  |  L0 chunk contraction  |  barrier, tile write, barrier  |  begin(): ring <- staged slabs,
//     request slab 2  |  request `pre` of the next chunk / next network  |  activation stores  |  L1 chunk contraction
#pragma once

namespace rowsfwd {

// `pre`: this wave's 32 weight rows of a short contraction (K <= 64), two slabs of 32 k: rows lane / 8 + 8 j.
// (Native vector type, not HIP's float4 struct: struct copies between global memory, this array and LDS became
// memcpy calls that kept the eight staging registers in SCRATCH memory - 144 B of private segment, 130 scratch
// instructions per workgroup walk - instead of VGPRs.)
using f4v = __attribute__((ext_vector_type(4))) float;
__device__ __forceinline__ void pre_request(f4v (&s0)[4], f4v (&s1)[4], const float* W, int ldw, int wave, int lane) {
  const float* gp = W + (int64_t)(wave * 32 + (lane >> 3)) * ldw + 4 * (lane & 7);
  const int64_t row8 = (int64_t)8 * ldw;
#pragma unroll
  for (int j = 0; j < 4; ++j) s0[j] = *reinterpret_cast<const f4v*>(gp + j * row8);
#pragma unroll
  for (int j = 0; j < 4; ++j) s1[j] = *reinterpret_cast<const f4v*>(gp + j * row8 + 32);   // (k past a 48-wide row: never multiplied)
}
__device__ __forceinline__ void pre_to_ring(const f4v (&s0)[4], const f4v (&s1)[4], float* wring, int lane) {
  float* p = wring + (lane >> 3) * kWS + 4 * (lane & 7);
#pragma unroll
  for (int j = 0; j < 4; ++j) *reinterpret_cast<f4v*>(p + 8 * j * kWS) = s0[j];
#pragma unroll
  for (int j = 0; j < 4; ++j) *reinterpret_cast<f4v*>(p + kSlot + 8 * j * kWS) = s1[j];
}

constexpr int kTileLd = kWidth + 4;     // activation tile row stride (floats)

template <int R>
constexpr size_t wide_lds_bytes(int ldx) { return sizeof(float) * ((size_t)R * ldx + (size_t)R * kTileLd + 8 * kRingWave); }

}  // namespace rowsfwd

// R / TRAIN / NETS as rows_fwd_kernel.  NL: hidden layers computed here (1..3).  NCH: 256-column chunks of layer 0
// (1: width 128 or 256, 2: width 512 - then NL >= 2).  Layer widths come from a.hidden[] (128 / 256; layer 0 also 512).
template <int R, bool TRAIN, int NETS, int NL, int NCH>
__global__ __launch_bounds__(rowsfwd::kThreads) void rows_fwd_wide_kernel(const FusedFwdArgs a) {
  using gemm::f32x16;
  using u32x4 = __attribute__((ext_vector_type(4))) unsigned int;
  static_assert(NCH == 1 || NL >= 2, "a 512-wide first layer is consumed chunk by chunk by the layer above it");
  constexpr int T = R / 32;
  constexpr int XQ = R * 16 / rowsfwd::kThreads;                      // float4 of the observation tile per thread (Dp <= 64)
  constexpr int HQ = R * (rowsfwd::kWidth / 4) / rowsfwd::kThreads;   // float4 of a 256-column activation tile per thread
  constexpr int LD = rowsfwd::kTileLd;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int ldx = a.ld1;
  float* tileX = smem;                                        // [R][ldx]   observations (persist across chunks and networks)
  float* tile = smem + R * ldx;                               // [R][LD]    activations of the current layer / chunk
  const int tid = threadIdx.x, lane = tid & 63, l31 = lane & 31, h = lane >> 5;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  float* wring = tile + R * LD + wave * rowsfwd::kRingWave;
  const int64_t r0 = (int64_t)blockIdx.x * R;
  const int W0 = a.hidden[0];
  // ---- observation tile: rows past M read zero through the buffer descriptor (no branch)
  const int q4 = a.Dp / 4;
  const __amdgpu_buffer_rsrc_t xrs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.x), 0,
                                                                       (int)(a.M * a.Dp * 4), 0x00020000);
  u32x4 xr[XQ];
  uint32_t xlds[XQ];
#pragma unroll
  for (int j = 0; j < XQ; ++j) {
    const int f = tid + j * rowsfwd::kThreads;
    const int r = f / q4, q = f - r * q4;
    const bool on = f < R * q4;
    xlds[j] = on ? (uint32_t)(r * ldx + 4 * q) : 0xffffffffu;
    xr[j] = __builtin_amdgcn_raw_buffer_load_b128(xrs, on ? (uint32_t)(((r0 + r) * a.Dp + 4 * q) * 4) : 0xffffffffu, 0, 0);
  }
  int net = a.net0 + (NETS == 2 ? 0 : (int)blockIdx.y);
  rowsfwd::Layer<R> ly;
  rowsfwd::f4v pre0[4], pre1[4];
  // a wave takes part in a layer (chunk) when its 32 columns exist there
  // (`pre` is requested and copied UNCONDITIONALLY - a wave without columns in the chunk fetches wave 0's rows and never
  // multiplies them: under a branch the compiler keeps the eight staging registers in scratch memory instead)
  const int wpre_first = wave * 32 < (W0 < 256 ? W0 : 256) ? wave : 0;
  rowsfwd::pre_request(pre0, pre1, a.params + a.off_w[net][0], a.Dp, wpre_first, lane);
#pragma unroll
  for (int j = 0; j < XQ; ++j)
    if (xlds[j] != 0xffffffffu) *reinterpret_cast<u32x4*>(tileX + xlds[j]) = xr[j];
  __syncthreads();

  // activation rows -> memory (TRAIN): HQ full-width requests per thread, the ones past `width` columns or past row M
  // are dropped by the descriptor / the offset (no branch: the compiler can count the stores)
  auto store_tile = [&](float* hg, int ldh, int col0, int width) {
    const __amdgpu_buffer_rsrc_t hrs = __builtin_amdgcn_make_buffer_rsrc(hg, 0, (int)(a.M * ldh * 4), 0x00020000);
    u32x4 hv[HQ];
#pragma unroll
    for (int j = 0; j < HQ; ++j) {
      const int f = tid + j * rowsfwd::kThreads;
      hv[j] = *reinterpret_cast<const u32x4*>(tile + (f >> 6) * LD + 4 * (f & 63));
    }
#pragma unroll
    for (int j = 0; j < HQ; ++j) {
      const int f = tid + j * rowsfwd::kThreads;
      const int c = 4 * (f & 63);
      const uint32_t ho = c < width ? (uint32_t)(((r0 + (f >> 6)) * ldh + col0 + c) * 4) : 0xffffffffu;
      __builtin_amdgcn_raw_buffer_store_b128(hv[j], hrs, ho, 0, 16);     // write-through (sc1), see rows_fwd_kernel
    }
  };
  auto write_tile = [&](const f32x16 (&acc)[T], float bias) {
#pragma unroll
    for (int t = 0; t < T; ++t)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = 32 * t + (r & 3) + 8 * (r >> 2) + 4 * h;
        tile[row * LD + wave * 32 + l31] = gemm::elu_f(acc[t][r] + bias);
      }
  };

#pragma unroll
  for (int ni = 0; ni < NETS; ++ni) {
    const int slot_net = NETS == 2 ? ni : (int)blockIdx.y;      // index into a.Hout
    const int W1 = NL >= 2 ? a.hidden[1] : 0;
    const bool on1 = NL >= 2 && wave * 32 < W1;
    f32x16 acc[T];                                            // accumulators of the layer ABOVE the one in the tile
    // ------------------------------------------------------------------ layer 0, chunk by chunk
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
      const int Wc = (W0 - 256 * c) < 256 ? (W0 - 256 * c) : 256;       // columns of this chunk
      const bool on0 = wave * 32 < Wc;
      float bias0 = 0.0f;
      f32x16 acc0[T];
      rowsfwd::pre_to_ring(pre0, pre1, wring, lane);
      if (on0) bias0 = a.params[a.off_b[net][0] + 256 * c + wave * 32 + l31];
      if (on1) ly.stage(a.params + a.off_w[net][1] + 256 * c, W0, wave, lane);      // layer 1's weights for this k chunk
      if (on0) ly.template loop<true>(tileX, ldx, wring, a.Dp, acc0, lane);
      __syncthreads();                                        // the tile is free: every wave is past the layer above's loop
      if (on0) write_tile(acc0, bias0);
      __syncthreads();
      if (on1) ly.begin(wring, lane);
      // the first layer's weights of the NEXT chunk / of the next network's first chunk: requested before the stores
      if (c + 1 < NCH) {
        const int wn = (W0 - 256 * (c + 1)) < 256 ? (W0 - 256 * (c + 1)) : 256;
        rowsfwd::pre_request(pre0, pre1, a.params + a.off_w[net][0] + (int64_t)256 * (c + 1) * a.Dp, a.Dp, wave * 32 < wn ? wave : 0, lane);
      } else if (ni + 1 < NETS) {
        rowsfwd::pre_request(pre0, pre1, a.params + a.off_w[net + 1][0], a.Dp, wpre_first, lane);
      }
      if (TRAIN) store_tile(a.Hout[slot_net][0], W0, 256 * c, Wc);
      if (on1) {
        if (c == 0) ly.template loop<true>(tile, LD, wring, Wc, acc, lane);
        else ly.template loop<false>(tile, LD, wring, Wc, acc, lane);
      }
    }
    // ------------------------------------------------------------------ layers 1 .. NL-1: in place on the tile
#pragma unroll
    for (int l = 1; l < NL; ++l) {
      const int Wl = a.hidden[l];
      const bool onl = wave * 32 < Wl;
      const bool more = l + 1 < NL;
      const bool onn = more && wave * 32 < a.hidden[more ? l + 1 : l];
      float bias = 0.0f;
      if (onl) bias = a.params[a.off_b[net][l] + wave * 32 + l31];
      if (onn) ly.stage(a.params + a.off_w[net][more ? l + 1 : l], Wl, wave, lane);
      __syncthreads();                                        // every wave is done reading the tile
      if (onl) write_tile(acc, bias);
      __syncthreads();
      if (onn) ly.begin(wring, lane);
      if (TRAIN) store_tile(a.Hout[slot_net][l], Wl, 0, Wl);
      if (onn) ly.template loop<true>(tile, LD, wring, Wl, acc, lane);
    }
    if (!TRAIN) {                // rollout: heads on the tile (the rings are free: head weights go there)
      float* wl = tile + R * LD;
      if (a.hidden[NL - 1] == 128) fused_head<128>(a, tile, LD, net, r0, wl);
      else fused_head<256>(a, tile, LD, net, r0, wl);
    }
    ++net;
  }
}
