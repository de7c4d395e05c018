// RunningMeanStd: column moments over a batch, Chan merge into the running state, normalise.
//
// Replaces cleanrl/ppo.py:12-62 (about a dozen eager launches per call; called T+1 times per
// iteration on the (N,D) observations and twice on the (T*N,) values/returns).
//   rms_moments_partial : per-block column sums of x and x^2 in fp64 (fixed order, no atomics)
//   rms_moments_final   : fold the partials -> sums[2*D]                    (all-reduce point)
//   rms_merge           : batch mean / biased variance, then the reference's fp32 Chan merge
//   rms_normalize       : (x - mean) / sqrt(var + eps), IEEE sqrt and division
// Built with -ffp-contract=off; the merge keeps the reference's op order:
//   new_mean = mean + fl(fl(delta*n)/tot);  M2 = fl(fl(var*count)+fl(bvar*n)) + fl(fl(fl(delta^2*count)*n)/tot)
#include "common.h"

namespace {

constexpr int kThreads = 256;

template <typename XT = float>
__global__ __launch_bounds__(kThreads) void rms_moments_partial(const XT* __restrict__ x, int64_t N, int D,
                                                                int64_t ldx, int rows_per_block,
                                                                double* __restrict__ partial) {
  __shared__ double s1[kThreads], s2[kThreads];
  const int Dc = D < kThreads ? D : kThreads;
  const int G = kThreads / Dc;
  const int c0 = threadIdx.x % Dc;
  const int g = threadIdx.x / Dc;
  for (int cb = 0; cb < D; cb += Dc) {
    const int c = cb + c0;
    const bool active = g < G && c < D;
    double a = 0.0, b = 0.0;
    if (active) {
      for (int64_t r0 = (int64_t)blockIdx.x * rows_per_block; r0 < N; r0 += (int64_t)gridDim.x * rows_per_block) {
        const int64_t r1 = r0 + rows_per_block < N ? r0 + rows_per_block : N;
        for (int64_t r = r0 + g; r < r1; r += G) {
          const double v = (double)x[r * ldx + c];
          a += v;
          b += v * v;
        }
      }
    }
    s1[threadIdx.x] = a;
    s2[threadIdx.x] = b;
    __syncthreads();
    if (g == 0 && c < D) {
      for (int gg = 1; gg < G; ++gg) {
        a += s1[gg * Dc + c0];
        b += s2[gg * Dc + c0];
      }
      partial[(int64_t)blockIdx.x * 2 * D + c] = a;
      partial[(int64_t)blockIdx.x * 2 * D + D + c] = b;
    }
    __syncthreads();
  }
}

// fold the per-block partials; thread (c,g): value c of the 2*D sums, partial-row group g, then a
// fixed-order combine of the G group sums in LDS (deterministic)
__global__ __launch_bounds__(kThreads) void rms_moments_final(const double* __restrict__ partial, int nblk, int D,
                                                              double* __restrict__ sums) {
  __shared__ double sm[kThreads];
  const int W = 2 * D;
  const int Wc = W < kThreads ? W : kThreads;
  const int G = kThreads / Wc;
  const int c0 = threadIdx.x % Wc, g = threadIdx.x / Wc;
  for (int cb = 0; cb < W; cb += Wc) {
    const int c = cb + c0;
    double a = 0.0;
    if (g < G && c < W) {
#pragma unroll 8
      for (int b = g; b < nblk; b += G) a += partial[(int64_t)b * W + c];
    }
    sm[threadIdx.x] = a;
    __syncthreads();
    if (g == 0 && c < W) {
      for (int gg = 1; gg < G; ++gg) a += sm[gg * Wc + c0];
      sums[c] = a;
    }
    __syncthreads();
  }
}

__global__ __launch_bounds__(kThreads) void rms_merge(const double* __restrict__ sums, double n, int D,
                                                      float* __restrict__ mean, float* __restrict__ var,
                                                      float* __restrict__ count) {
  const float cnt = count[0];
  const float nf = (float)n;                 // batch_count (Python int -> fp32 at the tensor op)
  const float tot = cnt + nf;                // tot_count = count + batch_count
  __syncthreads();                           // everyone has read count before it is rewritten
  for (int c = threadIdx.x; c < D; c += blockDim.x) {
    const double m = sums[c] / n;
    double v = sums[D + c] / n - m * m;      // biased variance (correction=0)
    if (v < 0.0) v = 0.0;
    const float bm = (float)m, bv = (float)v;
    const float delta = bm - mean[c];
    float t = delta * nf;                    // delta * batch_count
    t = t / tot;                             //   / tot_count
    const float new_mean = mean[c] + t;
    const float m_a = var[c] * cnt;
    const float m_b = bv * nf;
    float d2 = delta * delta;                // torch.square(delta)
    d2 = d2 * cnt;
    d2 = d2 * nf;
    d2 = d2 / tot;
    float M2 = m_a + m_b;
    M2 = M2 + d2;
    mean[c] = new_mean;
    var[c] = M2 / tot;
  }
  if (threadIdx.x == 0) count[0] = tot;
}

// single-GPU update: fold the partials and merge in ONE launch (2*D <= kFusedMax sums staged in LDS)
constexpr int kFusedMax = 1024;
__global__ __launch_bounds__(kThreads) void rms_final_merge(const double* __restrict__ partial, int nblk, double n,
                                                            int D, float* __restrict__ mean, float* __restrict__ var,
                                                            float* __restrict__ count) {
  __shared__ double sums[kFusedMax];
  __shared__ double sm[kThreads];
  const int W = 2 * D;
  const int Wc = W < kThreads ? W : kThreads;
  const int G = kThreads / Wc;
  const int c0 = threadIdx.x % Wc, g = threadIdx.x / Wc;
  for (int cb = 0; cb < W; cb += Wc) {
    const int c = cb + c0;
    double a = 0.0;
    if (g < G && c < W) {
#pragma unroll 8
      for (int b = g; b < nblk; b += G) a += partial[(int64_t)b * W + c];
    }
    sm[threadIdx.x] = a;
    __syncthreads();
    if (g == 0 && c < W) {
      for (int gg = 1; gg < G; ++gg) a += sm[gg * Wc + c0];
      sums[c] = a;
    }
    __syncthreads();
  }
  const float cnt = count[0];
  const float nf = (float)n;
  const float tot = cnt + nf;
  __syncthreads();
  for (int c = threadIdx.x; c < D; c += blockDim.x) {
    const double m = sums[c] / n;
    double v = sums[D + c] / n - m * m;
    if (v < 0.0) v = 0.0;
    const float bm = (float)m, bv = (float)v;
    const float delta = bm - mean[c];
    float t = delta * nf;
    t = t / tot;
    const float new_mean = mean[c] + t;
    const float m_a = var[c] * cnt;
    const float m_b = bv * nf;
    float d2 = delta * delta;
    d2 = d2 * cnt;
    d2 = d2 * nf;
    d2 = d2 / tot;
    float M2 = m_a + m_b;
    M2 = M2 + d2;
    mean[c] = new_mean;
    var[c] = M2 / tot;
  }
  if (threadIdx.x == 0) count[0] = tot;
}

template <typename XT = float>
__global__ __launch_bounds__(kThreads) void rms_normalize(const XT* __restrict__ x, int64_t N, int D,
                                                          int64_t ldx, const float* __restrict__ mean,
                                                          const float* __restrict__ var, float eps,
                                                          float* __restrict__ out, int64_t ldo) {
  extern __shared__ float lds[];  // [D] mean, [D] sqrt(var+eps)
  float* s_mean = lds;
  float* s_den = lds + D;
  for (int c = threadIdx.x; c < D; c += blockDim.x) {
    s_mean[c] = mean[c];
    s_den[c] = sqrtf(var[c] + eps);
  }
  __syncthreads();
  const int64_t total = N * D;
  for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (int64_t)gridDim.x * blockDim.x) {
    const int64_t r = e / D;
    const int c = (int)(e - r * D);
    const float v = (float)x[r * ldx + c] - s_mean[c];
    out[r * ldo + c] = v / s_den[c];
  }
}

// grid of rms_moments_partial: up to 128 workgroups (= partial rows folded afterwards); a thread walks at most 16 rows of
// a slab, fewer when that would leave the launch with under ~96 workgroups (98304 scalars: 24 workgroups x 16 dependent
// row loads took 19 us, 128 x 3 take a third of that)
inline void moments_grid(int64_t N, int G, int* rows_per_block, int* nblk) {
  int per_thread = 16;
  if (cdiv64(N, (int64_t)G * 16) < 96) {
    per_thread = (int)cdiv64(N, (int64_t)G * 128);
    if (per_thread < 1) per_thread = 1;
  }
  *rows_per_block = G * per_thread;
  int64_t nb = cdiv64(N, *rows_per_block);
  *nblk = (int)(nb > 128 ? 128 : nb);
}

int launch_moments(catppo_ctx* ctx, const float* x, int64_t N, int D, int64_t ldx, double* sums, hipStream_t s) {
  const int Dc = D < kThreads ? D : kThreads;
  const int G = kThreads / Dc;
  int rows_per_block, nblk;
  moments_grid(N, G, &rows_per_block, &nblk);
  WsCarver ws(ctx);
  double* partial = ws.take<double>((uint64_t)nblk * 2 * D);
  CATPPO_NEED_WS(ctx, partial);
  hipLaunchKernelGGL(rms_moments_partial<float>, dim3(nblk), dim3(kThreads), 0, s, x, N, D, ldx, rows_per_block, partial);
  CATPPO_CHECK_LAUNCH(ctx);
  hipLaunchKernelGGL(rms_moments_final, dim3(1), dim3(kThreads), 0, s, partial, nblk, D, sums);
  CATPPO_CHECK_LAUNCH(ctx);
  return CATPPO_OK;
}

// whole-batch advantage normalisation (skrl/ppo.py:436: (A - A.mean()) / (A.std() + 1e-8), unbiased std):
// every block folds the <=128 fp64 partial moment rows in the same fixed order, so all blocks hold the
// same mean / std, then normalises its slice.
__global__ __launch_bounds__(kThreads) void adv_normalize_kernel(const float* __restrict__ x, int64_t n,
                                                                 const double* __restrict__ partial, int nblk,
                                                                 float* __restrict__ out,
                                                                 float* __restrict__ stats) {
  double s1 = 0.0, s2 = 0.0;
  for (int b = 0; b < nblk; ++b) {
    s1 += partial[2 * b];
    s2 += partial[2 * b + 1];
  }
  const double cnt = (double)n;
  const double mean_d = s1 / cnt;
  double var = n > 1 ? (s2 - cnt * mean_d * mean_d) / (cnt - 1.0) : __builtin_nan("");
  if (var < 0.0) var = 0.0;
  const float mean = (float)mean_d;
  const float den = (float)sqrt(var) + 1e-8f;
  if (stats != nullptr && blockIdx.x == 0 && threadIdx.x == 0) stats[0] = mean, stats[1] = den;
  for (int64_t i = (int64_t)blockIdx.x * kThreads + threadIdx.x; i < n; i += (int64_t)gridDim.x * kThreads) {
    const float c = x[i] - mean;
    out[i] = c / den;
  }
}

__global__ void value_bootstrap_kernel(float* __restrict__ rew, const float* __restrict__ val,
                                       const uint8_t* __restrict__ to, float gamma, int64_t n) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  float b = gamma * val[i];                  // self.gamma * res_dict["values"]
  b = b * (to[i] ? 1.0f : 0.0f);             //   * time_outs.float()
  rew[i] = rew[i] + b;                       // shaped_rewards += ...
}

}  // namespace

static int rms_moments_f32(catppo_ctx* ctx, const float* x, int64_t N, int D, int64_t ldx, double* sums,
                                  void* stream) {
  CATPPO_CHECK_ARG(ctx, ctx != nullptr);
  CATPPO_CHECK_ARG(ctx, x && sums && N >= 1 && D >= 1 && D <= 65536 && ldx >= D);
  return launch_moments(ctx, x, N, D, ldx, sums, static_cast<hipStream_t>(stream));
}

extern "C" int catppo_rms_merge(catppo_ctx* ctx, const double* sums, double n, int D, float* mean, float* var,
                                float* count, void* stream) {
  CATPPO_CHECK_ARG(ctx, ctx != nullptr);
  CATPPO_CHECK_ARG(ctx, sums && mean && var && count && n >= 1.0 && D >= 1);
  hipLaunchKernelGGL(rms_merge, dim3(1), dim3(kThreads), 0, static_cast<hipStream_t>(stream), sums, n, D, mean, var,
                     count);
  CATPPO_CHECK_LAUNCH(ctx);
  return CATPPO_OK;
}

static int rms_update_f32(catppo_ctx* ctx, const float* x, int64_t N, int D, int64_t ldx, float* mean,
                                 float* var, float* count, void* stream) {
  CATPPO_CHECK_ARG(ctx, ctx != nullptr);
  CATPPO_CHECK_ARG(ctx, x && mean && var && count && N >= 1 && D >= 1 && D <= 65536 && ldx >= D);
  hipStream_t s = static_cast<hipStream_t>(stream);
  // the fp64 column sums live at the tail of the workspace, after the per-block partials
  const int Dc = D < kThreads ? D : kThreads;
  int rows_per_block, nblk;
  moments_grid(N, kThreads / Dc, &rows_per_block, &nblk);
  WsCarver ws(ctx);
  double* partial = ws.take<double>((uint64_t)nblk * 2 * D);
  double* sums = ws.take<double>((uint64_t)2 * D);
  CATPPO_NEED_WS(ctx, partial);
  CATPPO_NEED_WS(ctx, sums);
  if (2 * D <= kFusedMax) {
    hipLaunchKernelGGL(rms_moments_partial<float>, dim3(nblk), dim3(kThreads), 0, s, x, N, D, ldx, rows_per_block, partial);
    CATPPO_CHECK_LAUNCH(ctx);
    hipLaunchKernelGGL(rms_final_merge, dim3(1), dim3(kThreads), 0, s, (const double*)partial, nblk, (double)N, D, mean,
                       var, count);
    CATPPO_CHECK_LAUNCH(ctx);
    return CATPPO_OK;
  }
  if (int rc = launch_moments(ctx, x, N, D, ldx, sums, s)) return rc;
  hipLaunchKernelGGL(rms_merge, dim3(1), dim3(kThreads), 0, s, sums, (double)N, D, mean, var, count);
  CATPPO_CHECK_LAUNCH(ctx);
  return CATPPO_OK;
}

static int rms_normalize_f32(catppo_ctx* ctx, const float* x, int64_t N, int D, int64_t ldx,
                                    const float* mean, const float* var, float eps, float* out, int64_t ldo,
                                    void* stream) {
  CATPPO_CHECK_ARG(ctx, ctx != nullptr);
  CATPPO_CHECK_ARG(ctx, x && mean && var && out && N >= 1 && D >= 1 && D <= 16384 && ldx >= D && ldo >= D);
  int64_t nblk = cdiv64(N * D, kThreads * 4);
  if (nblk > 2048) nblk = 2048;
  hipLaunchKernelGGL(rms_normalize<float>, dim3((unsigned)nblk), dim3(kThreads), sizeof(float) * 2 * D,
                     static_cast<hipStream_t>(stream), x, N, D, ldx, mean, var, eps, out, ldo);
  CATPPO_CHECK_LAUNCH(ctx);
  return CATPPO_OK;
}

extern "C" int catppo_adv_normalize(catppo_ctx* ctx, const float* advantages, int64_t n, float* out, float* stats,
                                    void* stream) {
  CATPPO_CHECK_ARG(ctx, ctx != nullptr);
  CATPPO_CHECK_ARG(ctx, advantages && out && n >= 1);
  hipStream_t s = static_cast<hipStream_t>(stream);
  const int rows_per_block = kThreads * 16;
  int nblk = (int)cdiv64(n, rows_per_block);
  if (nblk > 128) nblk = 128;
  WsCarver ws(ctx);
  double* partial = ws.take<double>((uint64_t)nblk * 2);
  CATPPO_NEED_WS(ctx, partial);
  hipLaunchKernelGGL(rms_moments_partial<float>, dim3(nblk), dim3(kThreads), 0, s, advantages, n, 1, (int64_t)1, rows_per_block,
                     partial);
  CATPPO_CHECK_LAUNCH(ctx);
  int64_t nb = cdiv64(n, kThreads * 4);
  if (nb > 2048) nb = 2048;
  hipLaunchKernelGGL(adv_normalize_kernel, dim3((unsigned)nb), dim3(kThreads), 0, s, advantages, n,
                     (const double*)partial, nblk, out, stats);
  CATPPO_CHECK_LAUNCH(ctx);
  return CATPPO_OK;
}

extern "C" int catppo_value_bootstrap(catppo_ctx* ctx, float* rewards, const float* values, const uint8_t* time_outs,
                                      float gamma, int64_t N, void* stream) {
  CATPPO_CHECK_ARG(ctx, ctx != nullptr);
  CATPPO_CHECK_ARG(ctx, rewards && values && time_outs && N >= 1);
  hipLaunchKernelGGL(value_bootstrap_kernel, dim3((unsigned)cdiv64(N, 256)), dim3(256), 0,
                     static_cast<hipStream_t>(stream), rewards, values, time_outs, gamma, N);
  CATPPO_CHECK_LAUNCH(ctx);
  return CATPPO_OK;
}

// ---- fp16 inputs (BASELINE config 5: fp16 rollout planes) ------------------------------------------------------
extern "C" int catppo_rms_moments_ex(catppo_ctx* ctx, const void* x, int x_dtype, int64_t N, int D, int64_t ldx,
                                     double* sums, void* stream) {
  if (x_dtype == CATPPO_F32) return rms_moments_f32(ctx, static_cast<const float*>(x), N, D, ldx, sums, stream);
  CATPPO_CHECK_ARG(ctx, ctx != nullptr);
  CATPPO_CHECK_ARG(ctx, x_dtype == CATPPO_F16 && x && sums && N >= 1 && D >= 1 && D <= 65536 && ldx >= D);
  hipStream_t s = static_cast<hipStream_t>(stream);
  const int Dc = D < kThreads ? D : kThreads;
  int rows_per_block, nblk;
  moments_grid(N, kThreads / Dc, &rows_per_block, &nblk);
  WsCarver ws(ctx);
  double* partial = ws.take<double>((uint64_t)nblk * 2 * D);
  CATPPO_NEED_WS(ctx, partial);
  hipLaunchKernelGGL(rms_moments_partial<_Float16>, dim3(nblk), dim3(kThreads), 0, s, static_cast<const _Float16*>(x), N,
                     D, ldx, rows_per_block, partial);
  CATPPO_CHECK_LAUNCH(ctx);
  hipLaunchKernelGGL(rms_moments_final, dim3(1), dim3(kThreads), 0, s, partial, nblk, D, sums);
  CATPPO_CHECK_LAUNCH(ctx);
  return CATPPO_OK;
}

extern "C" int catppo_rms_update_ex(catppo_ctx* ctx, const void* x, int x_dtype, int64_t N, int D, int64_t ldx,
                                    float* mean, float* var, float* count, void* stream) {
  if (x_dtype == CATPPO_F32) return rms_update_f32(ctx, static_cast<const float*>(x), N, D, ldx, mean, var, count, stream);
  CATPPO_CHECK_ARG(ctx, ctx != nullptr);
  CATPPO_CHECK_ARG(ctx, x_dtype == CATPPO_F16);
  CATPPO_CHECK_ARG(ctx, x && mean && var && count && N >= 1 && D >= 1 && 2 * D <= kFusedMax && ldx >= D);
  hipStream_t s = static_cast<hipStream_t>(stream);
  const int Dc = D < kThreads ? D : kThreads;
  int rows_per_block, nblk;
  moments_grid(N, kThreads / Dc, &rows_per_block, &nblk);
  WsCarver ws(ctx);
  double* partial = ws.take<double>((uint64_t)nblk * 2 * D);
  CATPPO_NEED_WS(ctx, partial);
  hipLaunchKernelGGL(rms_moments_partial<_Float16>, dim3(nblk), dim3(kThreads), 0, s, static_cast<const _Float16*>(x), N,
                     D, ldx, rows_per_block, partial);
  CATPPO_CHECK_LAUNCH(ctx);
  hipLaunchKernelGGL(rms_final_merge, dim3(1), dim3(kThreads), 0, s, (const double*)partial, nblk, (double)N, D, mean,
                     var, count);
  CATPPO_CHECK_LAUNCH(ctx);
  return CATPPO_OK;
}

extern "C" int catppo_rms_normalize_ex(catppo_ctx* ctx, const void* x, int x_dtype, int64_t N, int D, int64_t ldx,
                                       const float* mean, const float* var, float eps, float* out, int64_t ldo,
                                       void* stream) {
  if (x_dtype == CATPPO_F32)
    return rms_normalize_f32(ctx, static_cast<const float*>(x), N, D, ldx, mean, var, eps, out, ldo, stream);
  CATPPO_CHECK_ARG(ctx, ctx != nullptr);
  CATPPO_CHECK_ARG(ctx, x_dtype == CATPPO_F16);
  CATPPO_CHECK_ARG(ctx, x && mean && var && out && N >= 1 && D >= 1 && D <= 16384 && ldx >= D && ldo >= D);
  int64_t nblk = cdiv64(N * D, kThreads * 4);
  if (nblk > 2048) nblk = 2048;
  hipLaunchKernelGGL(rms_normalize<_Float16>, dim3((unsigned)nblk), dim3(kThreads), sizeof(float) * 2 * D,
                     static_cast<hipStream_t>(stream), static_cast<const _Float16*>(x), N, D, ldx, mean, var, eps, out,
                     ldo);
  CATPPO_CHECK_LAUNCH(ctx);
  return CATPPO_OK;
}
