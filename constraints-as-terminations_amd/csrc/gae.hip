// GAE backward scan over a time-major (T,N) rollout with float (soft) dones.
//
// Replaces the reference's Python loop of ~11 eager ops per step (cleanrl/ppo.py:251-277,
// 264 launches at T=24) with one launch.  One lane owns VEC consecutive envs and walks t from
// T-1 to 0; at fixed t consecutive lanes read consecutive addresses of each (T,N) plane, so
// every load/store is a fully coalesced 256 B (VEC=1) or 1 KiB (VEC=4) wave transaction.
// The loads of step t do not depend on the recurrence, so the unrolled loop keeps several
// steps of loads in flight while the 5-flop dependent chain runs.
//
// Arithmetic is the reference's op order in unfused fp32 (this file is built with
// -ffp-contract=off) => advantages / returns are bit-identical to the CPU torch loop.
// Algorithmic HBM traffic: 4 reads + 2 writes = 24 B per env-step (+12 B per env bootstrap).
#include "common.h"

namespace {

// every element of the (T,N) planes is touched exactly once: stream them past the caches (nt).
// ST = storage type of the planes: float, or _Float16 (BASELINE config 5 "fp16 rollout buffer": the recurrence
// still runs in fp32 on the widened values, results are rounded to fp16 (RNE) on the way out).
template <int VEC, typename ST>
__device__ __forceinline__ void load(const ST* p, float (&v)[VEC]) {
  using vec_t = __attribute__((ext_vector_type(VEC))) ST;
  if constexpr (VEC == 1) {
    v[0] = (float)__builtin_nontemporal_load(p);
  } else {
    const vec_t q = __builtin_nontemporal_load(reinterpret_cast<const vec_t*>(p));
#pragma unroll
    for (int k = 0; k < VEC; ++k) v[k] = (float)q[k];
  }
}
template <int VEC, typename ST>
__device__ __forceinline__ void store(ST* p, const float (&v)[VEC]) {
  using vec_t = __attribute__((ext_vector_type(VEC))) ST;
  if constexpr (VEC == 1) {
    __builtin_nontemporal_store((ST)v[0], p);
  } else {
    vec_t q;
#pragma unroll
    for (int k = 0; k < VEC; ++k) q[k] = (ST)v[k];
    __builtin_nontemporal_store(q, reinterpret_cast<vec_t*>(p));
  }
}

// KIND selects the recurrence (all three are the reference's float-done variants):
//   0 CleanRL   (cleanrl/ppo.py:255-276)     d = done_{t+1}, time-out channel td_{t+1}
//   1 rl_games  (rl_games/cat_common.py:96-103 -> A2CBase.discount_values with float fdones): the CleanRL
//               recurrence without the time-out channel (tn == 1 exactly, one plane less to read)
//   2 skrl      (skrl/ppo.py:397-442)        d = done_t:  A_t = (r_t - v_t) + (g*nd_t) * (v_{t+1} + l*A_{t+1})
template <int VEC, int KIND, typename ST = float, int UNR = 4>
__global__ __launch_bounds__(256) void gae_scan(const ST* __restrict__ rew, const ST* __restrict__ val,
                                                const ST* __restrict__ done, const ST* __restrict__ tdone,
                                                const ST* __restrict__ next_val,
                                                const ST* __restrict__ next_done,
                                                const ST* __restrict__ next_tdone, float gamma, float gl,
                                                ST* __restrict__ adv, ST* __restrict__ ret, int T,
                                                int64_t N) {
  // XCD-aware block order (speed only): workgroup b runs on XCD b % 8 and every XCD has its own L2 / TLB; give each
  // XCD one contiguous eighth of the env axis so that it touches one page (not eight) per row of every plane
  int64_t blk = blockIdx.x;
  {
    const int64_t nb = gridDim.x, q = nb >> 3, r = nb & 7, xcd = blk & 7;
    blk = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (blk >> 3);
  }
  const int64_t env = (blk * blockDim.x + threadIdx.x) * VEC;
  if (env >= N) return;
  float vnext[VEC], dn[VEC], tdn[VEC], last[VEC];
  load<VEC, ST>(next_val + env, vnext);
  if constexpr (KIND != 2) load<VEC, ST>(next_done + env, dn);
  if constexpr (KIND == 0) load<VEC, ST>(next_tdone + env, tdn);
#pragma unroll
  for (int k = 0; k < VEC; ++k) last[k] = 0.0f;

#pragma unroll UNR
  for (int t = T - 1; t >= 0; --t) {
    const int64_t off = (int64_t)t * N + env;
    float r[VEC], v[VEC], d[VEC], td[VEC], a[VEC], q[VEC];
    load<VEC, ST>(rew + off, r);
    load<VEC, ST>(val + off, v);
    load<VEC, ST>(done + off, d);     // KIND 0/1: consumed by step t-1;  KIND 2: by this step
    if constexpr (KIND == 0) load<VEC, ST>(tdone + off, td);
#pragma unroll
    for (int k = 0; k < VEC; ++k) {
      if constexpr (KIND == 2) {
        const float nd = 1.0f - d[k];
        const float g = gamma * nd;         // discount_factor * not_dones[i]
        float c = gl * last[k];             // lambda_coefficient * advantage
        c = vnext[k] + c;                   // next_values + ...
        c = g * c;
        const float e = r[k] - v[k];        // rewards[i] - values[i]
        last[k] = e + c;
      } else {
        const float nn = 1.0f - dn[k];
        float x = gamma * vnext[k];   // GAMMA * nextvalues
        x = x * nn;                   //   * nextnonterminal
        float c = gl * nn;            // (GAMMA*GAE_LAMBDA) * nextnonterminal
        if constexpr (KIND == 0) {
          const float tn = 1.0f - tdn[k];
          x = x * tn;                 //   * true_nextnonterminal
          c = c * tn;
          tdn[k] = td[k];
        }
        float delta = r[k] + x;
        delta = delta - v[k];
        c = c * last[k];
        last[k] = delta + c;
        dn[k] = d[k];
      }
      a[k] = last[k];
      q[k] = last[k] + v[k];        // returns = advantages + values
      vnext[k] = v[k];
    }
    store<VEC, ST>(adv + off, a);
    store<VEC, ST>(ret + off, q);
  }
}

template <int KIND, typename ST>
void launch_gae(bool wide, const ST* rewards, const ST* values, const ST* dones, const ST* true_dones,
                const ST* next_value, const ST* next_done, const ST* next_true_done, float gamma, float gl,
                ST* advantages, ST* returns, int T, int64_t N, hipStream_t s) {
  constexpr int WV = 16 / (int)sizeof(ST);     // 16-B lanes: 4 fp32 / 8 fp16 envs
  if (wide) {
    const int block = 256;
    const dim3 grid((unsigned)cdiv64(N / WV, block));
    // with >= 16 waves per CU in the grid, eight time steps of loads in flight per lane pay (4.8 GB fp32 sweep:
    // 5.4 -> 6.1 TB/s); with fewer waves the longer dependent prologue costs more than it hides
    if (N / WV >= (int64_t)256 * 64 * 16)
      gae_scan<WV, KIND, ST, 8><<<grid, dim3(block), 0, s>>>(rewards, values, dones, true_dones, next_value, next_done,
                                                             next_true_done, gamma, gl, advantages, returns, T, N);
    else
      gae_scan<WV, KIND, ST, 4><<<grid, dim3(block), 0, s>>>(rewards, values, dones, true_dones, next_value, next_done,
                                                             next_true_done, gamma, gl, advantages, returns, T, N);
  } else {
    // small N: one wave per block so that 4096 envs already spread over 64 CUs
    const int block = N >= 65536 ? 256 : 64;
    gae_scan<1, KIND, ST><<<dim3((unsigned)cdiv64(N, block)), dim3(block), 0, s>>>(
        rewards, values, dones, true_dones, next_value, next_done, next_true_done, gamma, gl, advantages, returns, T, N);
  }
}

template <typename ST>
int gae_any(catppo_ctx* ctx, int kind, const ST* rewards, const ST* values, const ST* dones, const ST* true_dones,
            const ST* next_value, const ST* next_done, const ST* next_true_done, float gamma, float gamma_lambda,
            ST* advantages, ST* returns, int T, int64_t N, void* stream) {
  CATPPO_CHECK_ARG(ctx, ctx != nullptr);
  CATPPO_CHECK_ARG(ctx, kind >= CATPPO_GAE_CLEANRL && kind <= CATPPO_GAE_SKRL);
  CATPPO_CHECK_ARG(ctx, rewards && values && dones && next_value && advantages && returns);
  CATPPO_CHECK_ARG(ctx, kind == CATPPO_GAE_SKRL || next_done != nullptr);
  CATPPO_CHECK_ARG(ctx, kind != CATPPO_GAE_CLEANRL || (true_dones && next_true_done));
  CATPPO_CHECK_ARG(ctx, T >= 1 && N >= 1);
  hipStream_t s = static_cast<hipStream_t>(stream);
  auto aligned16 = [](const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; };
  // wide path only when every (T,N) row starts 16-B aligned and there are enough envs to
  // fill the chip with 16-B lanes (256 CUs x >=8 waves)
  constexpr int WV = 16 / (int)sizeof(ST);
  const bool wide = (N % WV == 0) && N >= (int64_t)256 * 64 * 8 && aligned16(rewards) &&
                    aligned16(values) && aligned16(dones) && aligned16(true_dones) && aligned16(next_value) &&
                    aligned16(next_done) && aligned16(next_true_done) && aligned16(advantages) &&
                    aligned16(returns) && (N * sizeof(ST)) % 16 == 0;
  switch (kind) {
    case CATPPO_GAE_CLEANRL:
      launch_gae<0, ST>(wide, rewards, values, dones, true_dones, next_value, next_done, next_true_done, gamma,
                        gamma_lambda, advantages, returns, T, N, s);
      break;
    case CATPPO_GAE_RL_GAMES:
      launch_gae<1, ST>(wide, rewards, values, dones, nullptr, next_value, next_done, nullptr, gamma, gamma_lambda,
                        advantages, returns, T, N, s);
      break;
    default:
      launch_gae<2, ST>(wide, rewards, values, dones, nullptr, next_value, nullptr, nullptr, gamma, gamma_lambda,
                        advantages, returns, T, N, s);
  }
  CATPPO_CHECK_LAUNCH(ctx);
  return CATPPO_OK;
}

// ---------------------------------------------------------------------------------------------------------------
// Scan mode (small N): the time axis of one env is split over G = 2^k lanes of a wavefront.
//   lane (g, e): env e of the wave's E = 64/G envs, time chunk g = steps [g*CH, (g+1)*CH)
//   1. the lane loads its chunk (independent of every other lane) and composes, from the chunk's last step
//      backwards, the affine map  A_in -> A_out = Dm + Cm * A_in   of the recurrence A_t = delta_t + c_t * A_{t+1}
//   2. inclusive suffix scan of the maps over g with wavefront shuffles (log2 G steps, lanes g*E+e and (g+s)*E+e):
//      afterwards Dm is A at the first step of the lane's chunk given A_T = 0
//   3. the value entering the chunk is the scanned Dm of lane g+1 (one more shuffle); replay the chunk serially
//      from it and store advantages / returns.
// For a fixed g the E lanes read E consecutive envs of one (T,N) row, so a wave touches G row segments of 4E bytes
// per load - the planes were written by the rollout a moment ago and sit in L2 / Infinity Cache at these sizes.
// 4096 envs x 24 steps: 512 wavefronts instead of 64, per-lane dependent chain 3 steps + 3 shuffles instead of 24.
template <int G, int CHMAX>
__global__ __launch_bounds__(256) void gae_scan_lanes(const float* __restrict__ rew, const float* __restrict__ val,
                                                      const float* __restrict__ done, const float* __restrict__ tdone,
                                                      const float* __restrict__ next_val,
                                                      const float* __restrict__ next_done,
                                                      const float* __restrict__ next_tdone, float gamma, float gl,
                                                      float* __restrict__ adv, float* __restrict__ ret, int T,
                                                      int64_t N, int CH) {
  constexpr int E = 64 / G;
  const int lane = threadIdx.x & 63;
  const int g = lane / E, e = lane - g * E;
  const int64_t wave = (int64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  const int64_t env = wave * E + e;
  const bool live = env < N;
  const int t0 = g * CH, t1 = (t0 + CH) < T ? (t0 + CH) : T;     // [t0, t1); empty when t0 >= T
  float dl[CHMAX], cc[CHMAX], vv[CHMAX];
  float Dm = 0.0f, Cm = 1.0f;                                      // identity map for empty chunks / dead lanes
  if (live) {
#pragma unroll
    for (int k = 0; k < CHMAX; ++k) {
      const int t = t0 + k;
      if (k < CH && t < T) {
        const int64_t off = (int64_t)t * N + env;
        const bool lastrow = t == T - 1;
        const float r = rew[off], v = val[off];
        const float nv = lastrow ? next_val[env] : val[off + N];
        const float nd = lastrow ? next_done[env] : done[off + N];
        const float ntd = lastrow ? next_tdone[env] : tdone[off + N];
        const float nn = 1.0f - nd, tn = 1.0f - ntd;
        float x = gamma * nv;
        x = x * nn;
        x = x * tn;
        float c = gl * nn;
        c = c * tn;
        float d = r + x;
        d = d - v;
        dl[k] = d, cc[k] = c, vv[k] = v;
      } else {
        dl[k] = 0.0f, cc[k] = 1.0f, vv[k] = 0.0f;
      }
    }
    // compose from the chunk's last step backwards: A_{t} = dl + cc * A_{t+1}
#pragma unroll
    for (int k = CHMAX - 1; k >= 0; --k) {
      Dm = dl[k] + cc[k] * Dm;
      Cm = cc[k] * Cm;
    }
  }
  // inclusive suffix scan over g: (D,C)_g <- (D,C)_g o (D,C)_{g+s}
#pragma unroll
  for (int s = 1; s < G; s <<= 1) {
    const float D2 = __shfl_down(Dm, s * E, 64), C2 = __shfl_down(Cm, s * E, 64);
    if (g + s < G) {
      Dm = Dm + Cm * D2;
      Cm = Cm * C2;
    }
  }
  float a_in = __shfl_down(Dm, E, 64);       // A at the first step of chunk g+1
  if (g == G - 1) a_in = 0.0f;
  if (!live) return;
#pragma unroll
  for (int k = CHMAX - 1; k >= 0; --k) {
    const int t = t0 + k;
    if (k < CH && t < t1) {
      a_in = dl[k] + cc[k] * a_in;
      const int64_t off = (int64_t)t * N + env;
      adv[off] = a_in;
      ret[off] = a_in + vv[k];
    }
  }
}

}  // namespace

static int gae_scan_impl(catppo_ctx* ctx, int mode, const float* rewards, const float* values,
                               const float* dones, const float* true_dones, const float* next_value,
                               const float* next_done, const float* next_true_done, float gamma, float gamma_lambda,
                               float* advantages, float* returns, int T, int64_t N, void* stream) {
  if (mode == CATPPO_GAE_SERIAL)
    return gae_any<float>(ctx, CATPPO_GAE_CLEANRL, rewards, values, dones, true_dones, next_value, next_done,
                          next_true_done, gamma, gamma_lambda, advantages, returns, T, N, stream);
  CATPPO_CHECK_ARG(ctx, ctx != nullptr);
  CATPPO_CHECK_ARG(ctx, mode == CATPPO_GAE_SCAN);
  CATPPO_CHECK_ARG(ctx, rewards && values && dones && true_dones && next_value && next_done && next_true_done &&
                            advantages && returns && T >= 1 && N >= 1);
  hipStream_t s = static_cast<hipStream_t>(stream);
  // lanes per env: as many as keep >= 2 steps per lane, at most 16; chunk length CH = ceil(T / G) <= 8
  int G = 8;                                      // 8 lanes x 8 envs per wave: 32-B row segments
  if ((T + G - 1) / G > 4) G = 16;                // long horizons: more lanes per env, chunks of <= 8 steps
  while (G > 1 && (T + G - 1) / G < 2) G >>= 1;
  int CH = (T + G - 1) / G;
  if (CH > 8) {    // long horizons: the serial kernel already has T-deep independent loads per lane
    return gae_any<float>(ctx, CATPPO_GAE_CLEANRL, rewards, values, dones, true_dones, next_value, next_done,
                         next_true_done, gamma, gamma_lambda, advantages, returns, T, N, stream);
  }
  const int E = 64 / G;
  const int64_t waves = cdiv64(N, E);
  const int wpb = 4;
  const dim3 grid((unsigned)cdiv64(waves, wpb)), block(64 * wpb);
#define CATPPO_GAE_SCAN_LAUNCH(GG, CM)                                                                              \
  hipLaunchKernelGGL((gae_scan_lanes<GG, CM>), grid, block, 0, s, rewards, values, dones, true_dones, next_value,   \
                     next_done, next_true_done, gamma, gamma_lambda, advantages, returns, T, N, CH)
  const int CM = CH <= 2 ? 2 : (CH <= 4 ? 4 : 8);
  if (G == 16 && CM == 2) CATPPO_GAE_SCAN_LAUNCH(16, 2);
  else if (G == 16 && CM == 4) CATPPO_GAE_SCAN_LAUNCH(16, 4);
  else if (G == 16) CATPPO_GAE_SCAN_LAUNCH(16, 8);
  else if (G == 8 && CM == 2) CATPPO_GAE_SCAN_LAUNCH(8, 2);
  else if (G == 8) CATPPO_GAE_SCAN_LAUNCH(8, 4);
  else if (G == 4) CATPPO_GAE_SCAN_LAUNCH(4, 2);
  else if (G == 2) CATPPO_GAE_SCAN_LAUNCH(2, 2);
  else CATPPO_GAE_SCAN_LAUNCH(1, 2);
#undef CATPPO_GAE_SCAN_LAUNCH
  CATPPO_CHECK_LAUNCH(ctx);
  return CATPPO_OK;
}

// ABI 0.6: the one GAE entry (catppo_gae / _ex / _f16 / _mode of ABI <= 0.5 are inline wrappers in include/catppo_compat.h)
extern "C" int catppo_gae_planes(catppo_ctx* ctx, int kind, int mode, int dtype, const void* rewards, const void* values,
                                 const void* dones, const void* true_dones, const void* next_value, const void* next_done,
                                 const void* next_true_done, float gamma, float gamma_lambda, void* advantages,
                                 void* returns, int T, int64_t N, void* stream) {
  CATPPO_CHECK_ARG(ctx, ctx != nullptr);
  CATPPO_CHECK_ARG(ctx, dtype == CATPPO_F32 || dtype == CATPPO_F16);
  CATPPO_CHECK_ARG(ctx, mode == CATPPO_GAE_SERIAL || (mode == CATPPO_GAE_SCAN && kind == CATPPO_GAE_CLEANRL && dtype == CATPPO_F32));
  if (mode == CATPPO_GAE_SCAN)
    return gae_scan_impl(ctx, mode, (const float*)rewards, (const float*)values, (const float*)dones, (const float*)true_dones,
                         (const float*)next_value, (const float*)next_done, (const float*)next_true_done, gamma, gamma_lambda,
                         (float*)advantages, (float*)returns, T, N, stream);
  if (dtype == CATPPO_F16) {
    using h = _Float16;
    return gae_any<h>(ctx, kind, (const h*)rewards, (const h*)values, (const h*)dones, (const h*)true_dones, (const h*)next_value,
                      (const h*)next_done, (const h*)next_true_done, gamma, gamma_lambda, (h*)advantages, (h*)returns, T, N, stream);
  }
  return gae_any<float>(ctx, kind, (const float*)rewards, (const float*)values, (const float*)dones, (const float*)true_dones,
                        (const float*)next_value, (const float*)next_done, (const float*)next_true_done, gamma, gamma_lambda,
                        (float*)advantages, (float*)returns, T, N, stream);
}
