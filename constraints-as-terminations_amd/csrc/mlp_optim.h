// Optimiser kernels (cleanrl/ppo.py:353-354, clip_grad_norm_ + optimizer.step()): squared-norm partials, clip + Adam with
// host scalars (catppo_clip_adam) and with the learning rate / step count in device memory (catppo_iter_state).
// Part of mlp.hip's translation unit (see mlp_common.h).
#pragma once

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

// ------------------------------------------------------------------------------- bf16 weight copies (bf16-stored mode)
// W_l (l >= 1) of both networks as bf16, as stored ([out][in]: forward operand) and transposed ([in][out]: the data gradient's
// K-contiguous operand), once per optimiser step from the fp32 master parameters - 0.26 M elements at cfg5.
struct W16Segs {
  int n;
  int64_t off[2 * CATPPO_MAX_HIDDEN];
  int out[2 * CATPPO_MAX_HIDDEN], in[2 * CATPPO_MAX_HIDDEN];
  int64_t first[2 * CATPPO_MAX_HIDDEN + 1];
};
__device__ __forceinline__ void w16_convert_block(const float* __restrict__ params, uint16_t* __restrict__ w16,
                                                  uint16_t* __restrict__ w16t, const W16Segs& t, const int block) {
  const int64_t e = (int64_t)block * 256 + threadIdx.x;
  if (e >= t.first[t.n]) return;
  int i = 0;
  while (i + 1 < t.n && e >= t.first[i + 1]) ++i;
  const int64_t le = e - t.first[i];
  w16[t.off[i] + le] = __builtin_bit_cast(uint16_t, (__bf16)params[t.off[i] + le]);
  // transposed copy: consecutive threads WRITE consecutive elements ([in][out], r fastest) and read a column of W from L2
  const int c = (int)(le / t.out[i]), r = (int)(le % t.out[i]);
  w16t[t.off[i] + le] = __builtin_bit_cast(uint16_t, (__bf16)params[t.off[i] + (int64_t)r * t.in[i] + c]);
}

// The first layer's forward (fp32-stored observations and W_0, bf16-stored output: PREC 4) with the weight conversion riding in
// the same launch: workgroups [0, n_conv) convert (n_conv a multiple of 8: the XCD of a GEMM workgroup is that of its tile index),
// the rest run 64x64 tiles.  Nothing in this launch reads the copies; the next launch (layer 1) does.
__global__ __launch_bounds__(256) void fwd0_w16_kernel(const gemm::Params p, const int n_conv, const int tiles,
                                                       const float* __restrict__ params, uint16_t* __restrict__ w16,
                                                       uint16_t* __restrict__ w16t, const W16Segs t) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int b = blockIdx.x;
  if (b < n_conv) {
    w16_convert_block(params, w16, w16t, t, b);
    return;
  }
  const gemm::TileId id = gemm::xcd_tile_of(b - n_conv, tiles, (gridDim.x - n_conv) / tiles, p.xcd_legacy);
  gemm::gemm_body<64, 64, true, true, gemm::EPI_BIAS_ELU, gemm::BK, 4>(p, id.tile, id.bz, smem);
}

// Layer-wise forward of hidden layers 0 .. n_layers - 1 with bf16-STORED activations (gemm_f32.h "act16"), `nets` networks from
// network 0: the first layer's launch also makes the bf16 weight copies of layers 1 .. nl - 1 (stored + transposed) that
// every later launch of the step reads.  last_fp32: the last of these layers writes fp32 (a consumer that reads fp32
// activations follows: head_act_kernel of the rollout forward).  Returns the number of weights converted.
int64_t forward_hidden16(const catppo_mlp_shape* sh, const catppo_mlp_layout& L, const float* params, const float* x, int64_t M,
                         const MlpWs& w, int nets, hipStream_t s, int n_layers, bool last_fp32) {
  const int nl = sh->n_hidden;
  W16Segs ws16{};
  int64_t tot = 0;
  for (int net = 0; net < nets; ++net)
    for (int l = 1; l < nl; ++l) {
      const int i = ws16.n++;
      ws16.off[i] = L.off_w[net][l], ws16.out[i] = sh->hidden[l], ws16.in[i] = L.in_dim[l];
      ws16.first[i] = tot;
      tot += (int64_t)sh->hidden[l] * L.in_dim[l];
    }
  ws16.first[ws16.n] = tot;
  for (int l = 0; l < n_layers; ++l) {
    Params pf{};
    pf.xcd_legacy = xcd_legacy();
    pf.nets = nets, pf.splits = 1;
    pf.I = (int)M, pf.J = sh->hidden[l];
    pf.ldc = sh->hidden[l];                           // bf16 or fp32 elements
    for (int net = 0; net < nets; ++net) {
      pf.op[net].bias = params + L.off_b[net][l];
      pf.op[net].C = w.H[net][l];
      if (l == 0) {
        pf.op[net].A = x, pf.op[net].B = params + L.off_w[net][0];
      } else {
        pf.op[net].A = w.H[net][l - 1];
        pf.op[net].B = reinterpret_cast<const float*>(w.w16 + L.off_w[net][l]);
      }
    }
    const bool out32 = last_fp32 && l == n_layers - 1;
    if (l == 0) {                                        // fp32-stored operands (observations, W_0), bf16-stored output
      pf.Kc = L.in_dim[0], pf.lda = L.in_dim[0], pf.ldb = L.in_dim[0];
      const int n_conv = (int)((cdiv64(tot, 256) + 7) / 8 * 8), t0 = tiles_of<64, 64>(pf);
      constexpr size_t lds0 = gemm::smem_bytes<64, 64, true, true>();
      hipLaunchKernelGGL(fwd0_w16_kernel, dim3((unsigned)(n_conv + t0 * nets)), dim3(256), lds0, s, pf, n_conv, t0, params, w.w16,
                         w.w16t, ws16);
    } else {                                             // bf16-stored operands: contraction sizes in FLOAT units
      pf.Kc = L.in_dim[l] / 2, pf.lda = L.in_dim[l] / 2, pf.ldb = L.in_dim[l] / 2;
      const bool big = pf.J >= 128 && L.in_dim[l] >= 256;
      if (out32) {
        if (big) launch_gemm_prec<128, 128, true, true, gemm::EPI_BIAS_ELU, 6>(pf, s);
        else launch_gemm_prec<64, 64, true, true, gemm::EPI_BIAS_ELU, 6>(pf, s);
      } else {
        if (big) launch_gemm_prec<128, 128, true, true, gemm::EPI_BIAS_ELU, 3>(pf, s);
        else launch_gemm_prec<64, 64, true, true, gemm::EPI_BIAS_ELU, 3>(pf, s);      // (64x64 for a wide layer: measured 0.8 us slower)
      }
    }
  }
  return tot;
}
