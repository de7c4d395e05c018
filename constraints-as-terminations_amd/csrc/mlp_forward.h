// Forward kernels of the actor-critic MLP (cleanrl/ppo.py:78-123): the rollout head (head_act_kernel), the one-launch
// small-batch forward of round 3 (fused_fwd_kernel) and the row-resident forwards of rounds 4 / 5 (rows_fwd_kernel,
// fwd_rows.h; rows_fwd_wide_kernel, fwd_rows_wide.h).  Part of mlp.hip's translation unit (see mlp_common.h).
#pragma once

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
  int64_t n_flat;                  // step16_fwd_kernel: floats in the flat parameter buffer (buffer descriptor range)
  int store_policy;                // rows_fwd_kernel activation stores: 0 all write-through (sc1), 1 write-through only for the
                                   // last layer of the last network a workgroup walks (the rest may sit in L2: they have the
                                   // rest of the launch to drain), 2 none
};

// Heads of the fused forward on the 32-row tile in LDS.  head_act_kernel gives every row a whole wave (the launch has
// thousands of waves to hide the Philox / Box-Muller / log-prob latency behind); a fused workgroup has four waves and
// 32 rows, so the wave-per-row form costs 8 serial rows of ~2500 dependent cycles each (8 us of a 45 us kernel).  Here
// the work is spread over items: actor = (row, action slot) with the 16 slots of a row in 16 adjacent lanes (two items
// per thread), critic = (row, eighth of the contraction) with 8 lanes per row.
template <int HL, int ROWS = 32>      // ROWS: rows of the tile (32: fused_fwd / rows_fwd kernels; 16: step16_fwd_kernel - items of rows >= ROWS are masked)
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
    if (part == 0 && i < a.M && r < ROWS) {
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
    const bool mine = kin && i < a.M && r < ROWS;
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
    if (k == 0 && i < a.M && r < ROWS) a.logprob[i] = lp;
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
  const int tid = threadIdx.x, lane = tid & 63;
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
