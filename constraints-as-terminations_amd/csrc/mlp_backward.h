// Backward tail of a minibatch (cleanrl/ppo.py:352): the fixed-order fold of split-K / head partials into the flat gradient
// (seg_reduce_kernel) and the first layer's weight-gradient launch that carries the fold of the other layers (dw_fold_kernel).
// The per-layer weight + data gradient GEMMs are gemm_f32.h's gemm_pair_kernel.  Part of mlp.hip's translation unit.
#pragma once

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
template <int PREC = 0>      // operand precision of the GEMM workgroups (round 5: the bf16 / split-bf16 modes take this launch too)
__global__ __launch_bounds__(256) void dw_fold_kernel(const gemm::Params p, const SegTable t, const int gemm_tiles,
                                                      const int n_gemm, float ent_coef, float vf_coef,
                                                      double* __restrict__ norm_slots) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int b = blockIdx.x;
  if (b < n_gemm) {
    const gemm::TileId id = gemm::xcd_tile_of(b, gemm_tiles, n_gemm / gemm_tiles, p.xcd_legacy);
    gemm::gemm_body<64, 64, false, false, gemm::EPI_PARTIAL, gemm::BK, PREC>(p, id.tile, id.bz, smem);
  } else {
    const int f = b - n_gemm;
    const double ss = seg_reduce_body(t.s[f / kFoldX], f % kFoldX, kFoldX, smem, ent_coef, vf_coef);
    if (norm_slots != nullptr) emit_norm_slot(ss, norm_slots + f, smem);
  }
}

// Every hidden layer's split-K weight gradient in ONE launch (round 6, the small-minibatch step of step16.h: all dZ are in
// memory when the backward chain's launch ends, so nothing orders the weight gradients against each other).  All
// problems use gemm_body's 64x64 EPI_PARTIAL tile - one instruction stream, the problem picked by workgroup index; the
// partial sums per element are those of the per-layer launches (same splits, same slab order).
struct DwMulti {
  gemm::Params p[CATPPO_MAX_HIDDEN];
  int first[CATPPO_MAX_HIDDEN + 1];    // first workgroup of problem i (launch order); first[n] = grid size
  int tiles[CATPPO_MAX_HIDDEN];        // 64x64 tiles of one (network, split) slice
  int n;
};
__global__ __launch_bounds__(256) void dw_multi_kernel(const DwMulti m) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int b = blockIdx.x;
  int i = 0;
  while (i + 1 < m.n && b >= m.first[i + 1]) ++i;
  const gemm::TileId id = gemm::xcd_tile_of(b - m.first[i], m.tiles[i], (m.first[i + 1] - m.first[i]) / m.tiles[i], m.p[i].xcd_legacy);
  gemm::gemm_body<64, 64, false, false, gemm::EPI_PARTIAL>(m.p[i], id.tile, id.bz, smem);
}
