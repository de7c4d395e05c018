// Minibatch gather and the head + PPO-loss kernels (cleanrl/ppo.py:298-345): ppo_gather_kernel, head_loss_kernel (heads +
// losses + head backward on stored activations) and fwd_head_kernel (last hidden layer + heads + loss + head backward in one
// launch).  Part of mlp.hip's translation unit (see mlp_common.h).
#pragma once

// ------------------------------------------------------------------------------- minibatch gather
// grid = (row chunks of one minibatch, minibatches).  Minibatch m = samples inds[m*M .. m*M + M_m) lands in the
// contiguous slices xmb[m*M ..], act[m*M ..], scal[4*m*M + {0,1,2,3}*M_m ..], adv_part[m][chunk][2].
__global__ __launch_bounds__(256) void ppo_gather_kernel(const float* __restrict__ b_obs, const float* __restrict__ b_act,
                                                         const float* __restrict__ b_logp,
                                                         const float* __restrict__ b_adv,
                                                         const float* __restrict__ b_ret,
                                                         const float* __restrict__ b_val,
                                                         const int64_t* __restrict__ inds, int64_t total, int64_t M,
                                                         int Dp, int A, float* __restrict__ xmb,
                                                         float* __restrict__ act, float* __restrict__ scal,
                                                         double* __restrict__ adv_part,
                                                         const catppo_iter_state* __restrict__ rng_state, int rng_epoch,
                                                         int adv_f16, int64_t* __restrict__ inds_out) {
  __shared__ int64_t s_idx[kGatherRows];
  const int64_t m0 = (int64_t)blockIdx.y * M;                     // first sample of this minibatch
  const int64_t Mm = (total - m0) < M ? (total - m0) : M;         // its size (the last one may be short)
  const int64_t r0 = (int64_t)blockIdx.x * kGatherRows;           // row chunk inside the minibatch
  if (r0 >= Mm) {
    if (threadIdx.x == 0) {
      adv_part[2 * ((int64_t)blockIdx.y * gridDim.x + blockIdx.x)] = 0.0;
      adv_part[2 * ((int64_t)blockIdx.y * gridDim.x + blockIdx.x) + 1] = 0.0;
    }
    return;
  }
  const int rows = (int)((Mm - r0) < kGatherRows ? (Mm - r0) : kGatherRows);
  xmb += m0 * Dp, act += m0 * A, scal += 4 * m0;
  adv_part += 2 * (int64_t)blockIdx.y * gridDim.x;
  if (threadIdx.x < rows) {
    int64_t src;
    if (rng_state != nullptr) {     // keyed bijection of [0,total): no index array, no sort (rng.h)
      rng::FeistelPerm perm;
      perm.init(rng_state->seed, rng_state->iteration, rng_epoch, total);
      src = perm(m0 + r0 + threadIdx.x);
      if (inds_out != nullptr) inds_out[m0 + r0 + threadIdx.x] = src;
    } else {
      src = inds[m0 + r0 + threadIdx.x];
    }
    s_idx[threadIdx.x] = src;
  }
  __syncthreads();
  const int q4 = Dp / 4;
  for (int f = threadIdx.x; f < rows * q4; f += 256) {
    const int r = f / q4, q = f - r * q4;
    reinterpret_cast<float4*>(xmb + (r0 + r) * Dp)[q] = reinterpret_cast<const float4*>(b_obs + s_idx[r] * Dp)[q];
  }
  for (int f = threadIdx.x; f < rows * A; f += 256) {
    const int r = f / A, k = f - r * A;
    act[(r0 + r) * A + k] = b_act[s_idx[r] * A + k];
  }
  if (threadIdx.x < 64) {   // wave 0: the four per-sample scalars + advantage moments
    double a1 = 0.0, a2 = 0.0;
    if (threadIdx.x < rows) {
      const int64_t src = s_idx[threadIdx.x], dst = r0 + threadIdx.x;
      const float adv = adv_f16 ? (float)reinterpret_cast<const _Float16*>(b_adv)[src] : b_adv[src];
      scal[0 * Mm + dst] = b_logp[src];
      scal[1 * Mm + dst] = adv;
      scal[2 * Mm + dst] = b_ret[src];
      scal[3 * Mm + dst] = b_val[src];
      a1 = (double)adv;
      a2 = a1 * a1;
    }
    a1 = wave_sum_d(a1);
    a2 = wave_sum_d(a2);
    if (threadIdx.x == 0) {
      adv_part[2 * blockIdx.x] = a1;
      adv_part[2 * blockIdx.x + 1] = a2;
    }
  }
}

// ------------------------------------------------------------------------------- heads + PPO loss + head backward
struct HeadArgs {
  const float *Hc, *Ha;        // [M, HL] last hidden activations (critic, actor)
  float *dZc, *dZa;            // [M, HL] out: gradient w.r.t. last hidden PRE-activations
  const float *W4c, *b4c, *W4a, *b4a, *logstd;
  const float *act, *oldlogp, *adv, *ret_n, *val_n;   // gathered minibatch
  const double* adv_part;      // [n_adv_part][2]
  int n_adv_part;
  const float* adv_stats;      // external {mean, std+1e-8} or null
  const float *vrms_mean, *vrms_var;
  float *part_w, *part_s;      // per-block partials
  int32_t* branch_out;         // debug (catppo_debug_clip_branches): [2][M] clip-branch codes, or null
  int64_t M;
  int A;
  catppo_ppo_hparams hp;
};

// clip-branch code of one sample and one clipped quantity: 0 inside [centre - clip, centre + clip], 1 below, 2 above - where the
// gradient of max(unclipped, clipped) switches (cleanrl/ppo.py:320-341); exported for tests/test_gpu_parity_sizes.py.  The
// value-loss code carries a second field (<< 2): which of (unclipped, clipped) is the max - 1 unclipped, 2 clipped, 0 tie - the
// other surface the gradient jumps at (2 (v - R) against 0 while the value difference is outside the clip range)
__device__ __forceinline__ int clip_code(float v, float centre, float clip) { return v < centre - clip ? 1 : (v > centre + clip ? 2 : 0); }

// waves per block: a 32-row tile of a wide last layer (HL >= 256) fills the CU's LDS alone, so the block brings
// its own parallelism (16 waves x 2 rows at HL = 256); narrower layers co-reside 2-3 blocks per CU and do better
// with 8 x 4, and HL = 512 needs more than the 128 VGPRs a 1024-thread block may use
template <int CPL>
constexpr int head_waves() { return CPL == 4 ? 16 : 8; }   // CPL 8 needs > 128 VGPRs: 8 waves
constexpr int kHeadMaxBlocks = 512;  // = number of weight-gradient partials folded afterwards (2 blocks per CU)

// Heads + PPO loss + backward through the heads, one tile of 32 minibatch rows at a time:
//   phase 1 (wave per row)  last-hidden rows -> registers AND an LDS tile; A+1 dot products per row
//           (batched 16-value butterfly), log-prob, clipped losses, analytic d loss/d mu, d loss/d v;
//           dZ of the last hidden layer is stored; the per-row head gradients go to an LDS [32][16] tile
//   phase 2 (thread per weight column)  dW4 += G^T . H over the 32 rows of the tile from LDS; accumulators
//           stay in registers across the tiles of the block => ONE partial per block, no per-wave
//           reduction rounds
template <int CPL, int TRS = 0>   // TRS: row-tile override (16 for small minibatches: twice the workgroups)
__global__ __launch_bounds__(head_waves<CPL>() * 64, (CPL <= 4 ? 4 : 2)) void head_loss_kernel(const HeadArgs g) {
  constexpr int kHeadWaves = head_waves<CPL>();
  constexpr int TR = TRS ? TRS : (CPL == 8 ? 16 : kHeadRowsPerBlock);
  constexpr int kHeadRowsPerWave = TR / kHeadWaves;
  constexpr int HL = CPL * 64;
  constexpr int NT = kHeadWaves * 64;
  constexpr int NG = NT / HL >= 1 ? NT / HL : 1;       // phase-2 thread groups (HL <= 512)
  constexpr int KPG = 16 / NG;                         // head outputs per group (16 slots)
  constexpr int VS = 15;                               // slot of the critic output; actions use slots 0..A-1
  // ALL shared memory lives in the dynamic region: a static __shared__ object in front of it would
  // shift its base off 16-B alignment and every ds_read_b128 below would be replayed (64 cycles each)
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int A = g.A;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int NS = 2 * A + 1 + kHeadDiag;
  // every loop below runs over the 16 compile-time slots; unused slots carry zeros (weights, gradients)
  // so there is no data-dependent control flow inside the row loop
  float* s_wa = lds;                       // [16*HL]  actor head weights, rows >= A zero
  float* sHa = s_wa + 16 * HL;             // [TR*HL]  actor last-hidden tile
  float* sHc = sHa + TR * HL;              // [TR*HL]  critic last-hidden tile
  float* sG = sHc + TR * HL;               // [TR*16]  per-row head gradients: d mu_k (k<A), 0, ..., d v at VS
  float* ls = sG + TR * 16;                // [NS]     scalars: db4a[A], db4c, dlogstd[A], diag[8]
  float* s_adv = ls + 48;                  // [2]      advantage mean, std + 1e-8   (NS <= 2*15+1+8 = 39)

  for (int o = tid; o < 16 * HL; o += NT) s_wa[o] = o < A * HL ? g.W4a[o] : 0.0f;
  for (int o = tid; o < TR * 16; o += NT) sG[o] = 0.0f;
  // advantage statistics over the minibatch (ppo.py:314-318): mean, unbiased std
  if (wave == 0) {
    if (g.hp.norm_adv && g.adv_stats == nullptr) {
      double a1 = 0.0, a2 = 0.0;
      for (int b = lane; b < g.n_adv_part; b += 64) {
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
  __syncthreads();
  const float adv_mean = s_adv[0], adv_den = s_adv[1];
  const float clipc = g.hp.clip_coef, invM = g.hp.inv_global_batch;
  const float vden = sqrtf(g.vrms_var[0] + 1e-8f), vmean = g.vrms_mean[0];
  const bool norm_adv = g.hp.norm_adv != 0, clip_vloss = g.hp.clip_vloss != 0;
  const float ent_coef_m = g.hp.ent_coef * invM, vf_half = g.hp.vf_coef * 0.5f;

  // after reduce16 the four lanes with slot(lane) == k hold the total of value k
  const int slot = ((lane >> 5) & 1) * 8 + ((lane >> 4) & 1) * 4 + ((lane >> 3) & 1) * 2 + ((lane >> 2) & 1);
  const bool mine = slot < A;
  const bool leader = mine && (lane & 3) == 0;    // one lane per action dim accumulates / publishes
  const float sd = mine ? expf(g.logstd[slot]) : 1.0f;
  const float var = sd * sd, lsd = logf(sd);
  const float ba = mine ? g.b4a[slot] : 0.0f;
  float ent_row = 0.0f;                           // entropy is state independent
  {
    const float e = mine ? kEntConst + lsd : 0.0f;
#pragma unroll
    for (int k = 0; k < VS; ++k) ent_row += lane_bcast(e, slot_lane(k));
  }
  float gls = 0.0f;                               // d loss / d logstd_k (leader lanes)
  float d_pg = 0.0f, d_v = 0.0f, d_ent = 0.0f, d_kl = 0.0f, d_okl = 0.0f, d_cf = 0.0f;
  float wc[CPL];
#pragma unroll
  for (int c = 0; c < CPL; ++c) wc[c] = g.W4c[lane * CPL + c];
  const float bc = g.b4c[0];

  // phase-2 ownership: weight column c2, slots [k0, k0+KPG)
  const int c2 = tid % HL, grp = tid / HL, k0 = grp * KPG;
  float acc[KPG];
#pragma unroll
  for (int kk = 0; kk < KPG; ++kk) acc[kk] = 0.0f;
  float accb = 0.0f;                              // bias gradients: threads 0..15 (one per slot)

  const int64_t n_tiles = (g.M + TR - 1) / TR;
  for (int64_t tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
    const int64_t row0 = tile * TR;
    const int rows = (int)((g.M - row0) < TR ? (g.M - row0) : TR);
    // ------------------------------------------------------------------ phase 1
    for (int rr = 0; rr < kHeadRowsPerWave; ++rr) {
      const int r = wave * kHeadRowsPerWave + rr;
      if (r >= rows) break;                        // wave-uniform
      const int64_t i = row0 + r;
      float hc[CPL], ha[CPL], part[16];
      load_vec<CPL>(g.Hc + i * HL + lane * CPL, hc);
      load_vec<CPL>(g.Ha + i * HL + lane * CPL, ha);
      const float a_taken = mine ? g.act[i * A + slot] : 0.0f;
      const float oldlogp = g.oldlogp[i], adv_raw = g.adv[i], R = g.ret_n[i], Vo = g.val_n[i];
      store_vec<CPL>(sHc + r * HL + lane * CPL, hc);
      store_vec<CPL>(sHa + r * HL + lane * CPL, ha);
#pragma unroll
      for (int k = 0; k < VS; ++k) {
        float d = 0.0f, wk[CPL];
        load_vec<CPL>(s_wa + k * HL + lane * CPL, wk);
#pragma unroll
        for (int c = 0; c < CPL; ++c) d = fmaf(ha[c], wk[c], d);
        part[k] = d;
      }
      {
        float d = 0.0f;
#pragma unroll
        for (int c = 0; c < CPL; ++c) d = fmaf(hc[c], wc[c], d);
        part[VS] = d;
      }
      const float tot = reduce16(part, lane);     // lanes of slot k: mu_k (k<A) / critic output (slot VS)
      const float mu = tot + ba;
      const float v = lane_bcast(tot, slot_lane(VS)) + bc;

      // ---- log-prob of the taken action
      const float diff = mine ? a_taken - mu : 0.0f;
      const float term = mine ? -(diff * diff) / (2.0f * var) - lsd - kHalfLog2Pi : 0.0f;
      float newlogp = 0.0f;
#pragma unroll
      for (int k = 0; k < VS; ++k) newlogp += lane_bcast(term, slot_lane(k));
      const float logratio = newlogp - oldlogp;
      const float ratio = expf(logratio);
      d_okl += -logratio;
      d_kl += (ratio - 1.0f) - logratio;
      d_cf += fabsf(ratio - 1.0f) > clipc ? 1.0f : 0.0f;

      const float adv = norm_adv ? (adv_raw - adv_mean) / adv_den : adv_raw;
      const float rc = ratio < 1.0f - clipc ? 1.0f - clipc : (ratio > 1.0f + clipc ? 1.0f + clipc : ratio);
      const float pg1 = -adv * ratio, pg2 = -adv * rc;
      const bool inside = ratio >= 1.0f - clipc && ratio <= 1.0f + clipc;
      // d max(pg1,pg2) / d ratio   (torch.max splits ties 1/2 : 1/2; clamp passes gradient inside only)
      const float dr_tie = 0.5f * -adv + (inside ? 0.5f * -adv : 0.0f);
      const float dr = pg1 > pg2 ? -adv : (pg1 < pg2 ? (inside ? -adv : 0.0f) : dr_tie);
      d_pg += pg1 > pg2 ? pg1 : pg2;
      const float g_logp = dr * ratio * invM;      // d loss / d newlogprob_i

      // ---- value head loss
      const float nv = (v - vmean) / vden;         // value_rms(newvalue, update=False)
      const float e1 = nv - R;
      const float vl1 = e1 * e1;
      const float dl = nv - Vo;
      const float cl = dl < -clipc ? -clipc : (dl > clipc ? clipc : dl);
      const float e2 = (Vo + cl) - R;
      const float vl2 = e2 * e2;
      const bool in2 = dl >= -clipc && dl <= clipc;
      const float dnv_c = vl1 > vl2 ? 2.0f * e1 : (vl1 < vl2 ? (in2 ? 2.0f * e2 : 0.0f) : e1 + (in2 ? e2 : 0.0f));
      const float vl = clip_vloss ? (vl1 > vl2 ? vl1 : vl2) : vl1;
      const float dnv = clip_vloss ? dnv_c : 2.0f * e1;
      d_v += 0.5f * vl;
      d_ent += ent_row;
      if (g.branch_out != nullptr && lane == 0) {
        g.branch_out[i] = clip_code(ratio, 1.0f, clipc);
        g.branch_out[g.M + i] = clip_code(dl, 0.0f, clipc) | ((vl1 > vl2 ? 1 : (vl1 < vl2 ? 2 : 0)) << 2);
      }
      const float g_v = vf_half * dnv * invM / vden;   // d loss / d v_i

      // ---- backward through the heads
      const float gm = mine ? g_logp * diff / var : 0.0f;           // d loss / d mu_ik   (lanes of slot k)
      if (leader) {
        gls += g_logp * (diff * diff / var - 1.0f) - ent_coef_m;
        sG[r * 16 + slot] = gm;
      }
      if (lane == 63) sG[r * 16 + VS] = g_v;         // lane 63 has slot 15 = VS
      float dha[CPL];
#pragma unroll
      for (int c = 0; c < CPL; ++c) dha[c] = 0.0f;
#pragma unroll
      for (int k = 0; k < VS; ++k) {
        const float gmk = lane_bcast(gm, slot_lane(k));
        float wk[CPL];
        load_vec<CPL>(s_wa + k * HL + lane * CPL, wk);
#pragma unroll
        for (int c = 0; c < CPL; ++c) dha[c] = fmaf(gmk, wk[c], dha[c]);
      }
      float oa[CPL], oc[CPL];
#pragma unroll
      for (int c = 0; c < CPL; ++c) {
        oa[c] = dha[c] * (ha[c] > 0.0f ? 1.0f : ha[c] + 1.0f);
        oc[c] = (g_v * wc[c]) * (hc[c] > 0.0f ? 1.0f : hc[c] + 1.0f);
      }
      store_vec_wt<CPL>(g.dZa + i * HL + lane * CPL, oa);
      store_vec_wt<CPL>(g.dZc + i * HL + lane * CPL, oc);
    }
    __syncthreads();
    // ------------------------------------------------------------------ phase 2: dW4 += G^T . H
    for (int r = 0; r < rows; ++r) {
      const float ha2 = sHa[r * HL + c2], hc2 = sHc[r * HL + c2];
      float gk[KPG];
      load_vec<KPG>(sG + r * 16 + k0, gk);        // k0 is a multiple of KPG: one or two b128 broadcasts
#pragma unroll
      for (int kk = 0; kk < KPG; ++kk) acc[kk] = fmaf(gk[kk], (k0 + kk) == VS ? hc2 : ha2, acc[kk]);
    }
    if (tid < 16) {
      for (int r = 0; r < rows; ++r) accb += sG[r * 16 + tid];
    }
    __syncthreads();
  }

  // ---- per-block partials: weight gradients straight from the phase-2 registers, scalars through LDS
  float* pw = g.part_w + (int64_t)blockIdx.x * (A + 1) * HL;   // rows 0..A-1 = dW4a, row A = dW4c
#pragma unroll
  for (int kk = 0; kk < KPG; ++kk) {
    const int k = k0 + kk;
    if (k < A) pw[k * HL + c2] = acc[kk];
    else if (k == VS) pw[A * HL + c2] = acc[kk];
  }
  for (int w = 0; w < kHeadWaves; ++w) {           // fixed wave order => deterministic
    if (wave == w) {
      if (leader) ls[A + 1 + slot] = w == 0 ? gls : ls[A + 1 + slot] + gls;
      if (lane == 63) {
        float* dg = ls + 2 * A + 1;
        const float vals[kHeadDiag] = {d_pg, d_v, d_ent, 0.0f, d_kl, d_okl, d_cf, 0.0f};
#pragma unroll
        for (int q = 0; q < kHeadDiag; ++q) dg[q] = w == 0 ? vals[q] : dg[q] + vals[q];
      }
    }
    __syncthreads();
  }
  if (tid < A) ls[tid] = accb;                     // db4a[0..A-1]
  if (tid == VS) ls[A] = accb;                     // db4c
  __syncthreads();
  float* ps = g.part_s + (int64_t)blockIdx.x * NS;
  for (int o = tid; o < NS; o += NT) ps[o] = ls[o];
}

// ------------------------------------------------------------------------------- last hidden layer + heads + loss
// One launch instead of the last forward GEMM followed by head_loss_kernel (28 us at M = 16384 with no matrix work,
// 33 MB of last-layer activations written and read back): a workgroup owns 64 rows of ONE network over the full
// last-layer width, leaves H = elu(X W^T + b) in LDS (gemm::EPI_BIAS_ELU_LDS) and runs that network's head, its part
// of the PPO loss and the backward through the head on the tile.  The three head products are small GEMMs on the
// same fp32 MFMA (16 head outputs, rows past the real count zero):
//   A  Y[64,16]   = H[64,HL] . Wh^T          each wave a quarter of the contraction, quarters added in fixed order
//   -  row math   one thread per row: log-prob / ratio / clipped surrogate / d loss/d mu, or value loss / d loss/d v
//                 (the arithmetic of head_loss_kernel, ppo.py:299-345) -> G[64,16]
//   C  dWh[16,HL] = G^T . H                  contraction over the 64 rows; per-workgroup partial
//   B  dZ[64,HL]  = (G . Wh) * elu'(H)       written over H in LDS, every wave then streams out its own 32 x HL/2 region
// Partial rows [0, RB) belong to the actor workgroups, [RB, 2 RB) to the critic's (row layout of head_loss_kernel,
// each kind writes only its own entries; the fold reads them with separate base pointers).
// -DFWD_HEAD_TL (tools/fwd_head_timeline.py): thread 0 of every workgroup stamps the shader clock at the step boundaries
#ifdef FWD_HEAD_TL
__device__ unsigned long long* g_fhtl;    // [2 nets][1024 workgroups][8 stamps]
#define FH_TL(i) do { if (threadIdx.x == 0 && g_fhtl) g_fhtl[(net * 1024 + blockIdx.x) * 8 + (i)] = wall_clock64(); } while (0)
#else
#define FH_TL(i) do { } while (0)
#endif

#ifndef FWD_HEAD_LEAN_CRITIC
#define FWD_HEAD_LEAN_CRITIC 1     // 0: the critic's head steps on the matrix pipe like the actor's, critic workgroups dispatched first (rounds 2-5)
#endif
#ifndef FWD_HEAD_MFMA16
#define FWD_HEAD_MFMA16 1          // 0: steps A and C on v_mfma_f32_32x32x2_f32 with the 16 head outputs padded to 32 (rounds 2-4)
#endif
template <int HL>
constexpr size_t fwd_head_lds_floats() { return (size_t)64 * (HL + gemm::kLdsTilePad) + 64 * 16 + 64 * 16 + 64 * 8 + 4 + HL; }

template <int HL, int PREC = 0>      // PREC: operand precision of the hidden-layer GEMM (gemm_body); the head products stay fp32
__global__ __launch_bounds__(256, 2) void fwd_head_kernel(const Params p, const HeadArgs g) {   // two workgroups per CU
  using gemm::f32x16;
  constexpr int BM = 64, LD = HL + gemm::kLdsTilePad;
  constexpr int KQ = HL / 4;            // contraction share of a wave in step A
  constexpr int TNB = HL / 64;          // 32-column tiles per wave in step B (waves 2 x 2)
  [[maybe_unused]] constexpr int TNC = HL / 128;         // 32-column tiles per wave in step C (waves 1 x 4)
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* Hs = smem;                     // [64][LD]   activated tile, later dZ
  float* sG = Hs + BM * LD;             // [64][16]   d loss / d head output k of row r (zero beyond the real outputs)
  float* sMu = sG + BM * 16;            // [64][16]   head outputs, later the per-row d loss / d logstd_k terms
  float* sD = sMu + BM * 16;            // [64][8]    per-row diagnostics {pg, v, ent, -, kl, old_kl, clipfrac, -}
  float* s_adv = sD + BM * 8;           // [2]        advantage mean, std + 1e-8
  [[maybe_unused]] float* sW = s_adv + 4;   // [HL]   lean critic epilogue: the value head's weight row
  // Round 6: the ACTOR workgroups are dispatched first (blockIdx.z = 0).  The two workgroups of a CU - the actor's and the
  // critic's tile of the same 64 rows - drift apart by themselves: the one dispatched first wins the matrix pipe, ends its main
  // loop ~8 us ahead and runs its epilogue under the rest of the partner's loop; the epilogue of the SECOND one is exposed
  // (13.8 us of the launch's 57 in round 5, profiles/r5_fwd_head_timeline.txt).  So the second one is now the critic, whose
  // single head output needs no matrix instruction (the lean epilogue below).
  const int net = FWD_HEAD_LEAN_CRITIC ? 1 - (int)blockIdx.z : (int)blockIdx.z;           // 0 critic, 1 actor (Params::op order)
  const int tid = threadIdx.x, lane = tid & 63, l31 = lane & 31, h = lane >> 5;
  const int q = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int A = g.A;
  const int RB = gridDim.x;
  const int tile = gemm::xcd_tile_index(blockIdx.x, gridDim.x);
  const int64_t i0 = (int64_t)tile * BM;
  const int rows = (int)((g.M - i0) < BM ? (g.M - i0) : BM);
  const int NS = 2 * A + 1 + kHeadDiag;

  if (net == 1 && tid < 64) {           // advantage statistics over the minibatch (ppo.py:314-318): mean, unbiased std
    if (g.hp.norm_adv && g.adv_stats == nullptr) {
      double a1 = 0.0, a2 = 0.0;
      for (int b = lane; b < g.n_adv_part; b += 64) {
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

  FH_TL(0);
  const float* Wh = net == 1 ? g.W4a : g.W4c;          // [KH][HL] head weights of this network
  const int KH = net == 1 ? A : 1;
  const float4 zero4 = make_float4(0.f, 0.f, 0.f, 0.f);
  const int wm = q >> 1, wn = q & 1;
  const bool mat = !(FWD_HEAD_LEAN_CRITIC && net == 0);      // this workgroup's head steps run on the matrix pipe
  // Everything the epilogue reads from global memory is requested here, ahead of the main loop: the head weights in
  // the operand layouts of steps A and B (rows past KH zero) and the gathered scalars of the row this thread will
  // work on (row math: four threads per row, thread part pp owns the action dims pp, pp+4, pp+8, pp+12).
#if FWD_HEAD_MFMA16
  // steps A and C on v_mfma_f32_16x16x4_f32 (round 5): 16 head outputs = ONE 16-wide tile, no zero-padded half as in the
  // 32x32x2 form (64 MFMAs x 32 cycles per wave and step instead of 64 x 64; profiles/r5_fwd_head_timeline.txt).
  // lane = (c16 = lane % 16: head output / column inside a tile, g4 = lane / 16: one of the 4 k of an instruction)
  const int c16 = lane & 15, g4 = lane >> 4;
  float4 bw[KQ / 16];          // step A: Wh[c16][q KQ + 16 blk + 4 g4 + s], MFMA (blk, s) contracts k = 16 blk + 4 g + s over g
#pragma unroll
  for (int kb = 0; kb < KQ / 16; ++kb)
    bw[kb] = mat && c16 < KH ? *reinterpret_cast<const float4*>(Wh + c16 * HL + q * KQ + 16 * kb + 4 * g4) : zero4;
#else
  float4 bw[KQ / 8];
#pragma unroll
  for (int kb = 0; kb < KQ / 8; ++kb)
    bw[kb] = l31 < KH ? *reinterpret_cast<const float4*>(Wh + l31 * HL + q * KQ + 8 * kb + 4 * h) : zero4;
#endif
  float bwB[TNB][2][4];
#pragma unroll
  for (int tn = 0; tn < TNB; ++tn)
#pragma unroll
    for (int blk = 0; blk < 2; ++blk)
#pragma unroll
      for (int sx = 0; sx < 4; ++sx) {
        const int kk = 8 * blk + 4 * h + sx;
        bwB[tn][blk][sx] = mat && kk < KH ? Wh[kk * HL + wn * (HL / 2) + 32 * tn + l31] : 0.0f;
      }
  const int rr = tid >> 2, pp = tid & 3;               // row math: row, part
  const bool rvalid = rr < rows;
  const int64_t ri = i0 + (rvalid ? rr : 0);
  const float rs0 = net == 1 ? g.oldlogp[ri] : g.ret_n[ri];
  const float rs1 = net == 1 ? g.adv[ri] : g.val_n[ri];
  float ract[4], rls[4], rb[4];
#pragma unroll
  for (int kk = 0; kk < 4; ++kk) {
    const int k = pp + 4 * kk;
    const bool on = net == 1 && k < A;
    ract[kk] = on ? g.act[ri * A + k] : 0.0f;
    rls[kk] = on ? g.logstd[k] : 0.0f;
    rb[kk] = on ? g.b4a[k] : 0.0f;
  }
  const float rbc = g.b4c[0], rvv = g.vrms_var[0], rvm = g.vrms_mean[0];

  constexpr bool kLean = FWD_HEAD_LEAN_CRITIC != 0;
  if (kLean && net == 0 && tid < HL / 4)               // (sW lies beyond the slab buffers; the main loop's barriers publish it)
    *reinterpret_cast<float4*>(sW + 4 * tid) = *reinterpret_cast<const float4*>(g.W4c + 4 * tid);

  gemm::gemm_body<BM, HL, true, true, gemm::EPI_BIAS_ELU_LDS, gemm::BK, PREC, 1>(p, tile, net, smem);
  FH_TL(1);

  const float clipc = g.hp.clip_coef, invM = g.hp.inv_global_batch;
  const int prow = net == 1 ? tile : RB + tile;
  if (kLean && net == 0) {
    // ---- lean critic epilogue (round 6): ONE head output - steps A, C, B as plain fp32 FMA chains in exactly the order the
    //      matrix instructions walk them (v_mfma_f32_16x16x4_f32 / 32x32x2 = one FMA chain per element in k order,
    //      tools/mfma16_probe.hip), so every value is bit-identical to the matrix form; no operand padding (15 of 16 head
    //      columns were zeros), no partial-sum round trips through LDS, two workgroup barriers instead of six.
    __syncthreads();                                     // H tile complete
    // A: v = H[r,:] . w.  Four threads per row, thread part pp = contraction quarter pp (what wave pp did): k = pp KQ + 16 kb
    //    + 4 g + s in the order kb, s, g; quarters combined (q0 + q1) + (q2 + q3)
    float vsum;
    {
      float acc = 0.0f;
      const float* hrow = Hs + rr * LD + pp * KQ;
      const float* wq = sW + pp * KQ;
#pragma unroll
      for (int kb = 0; kb < KQ / 16; ++kb) {
        float hv[4][4], wv[4][4];
#pragma unroll
        for (int gq = 0; gq < 4; ++gq) {
          const float4 h4 = *reinterpret_cast<const float4*>(hrow + 16 * kb + 4 * gq);
          const float4 w4 = *reinterpret_cast<const float4*>(wq + 16 * kb + 4 * gq);
          hv[gq][0] = h4.x, hv[gq][1] = h4.y, hv[gq][2] = h4.z, hv[gq][3] = h4.w;
          wv[gq][0] = w4.x, wv[gq][1] = w4.y, wv[gq][2] = w4.z, wv[gq][3] = w4.w;
        }
#pragma unroll
        for (int sx = 0; sx < 4; ++sx)
#pragma unroll
          for (int gq = 0; gq < 4; ++gq) acc = __builtin_fmaf(hv[gq][sx], wv[gq][sx], acc);
      }
      const float t2 = acc + __shfl_xor(acc, 1, 64);     // q0 + q1 | q2 + q3
      vsum = t2 + __shfl_xor(t2, 2, 64);
    }
    FH_TL(2);
    // row math (value loss, ppo.py:325-338): thread part 0 of the row
    {
      float dg1 = 0.0f, gm0 = 0.0f;
      if (rvalid && pp == 0) {
        const bool clip_vloss = g.hp.clip_vloss != 0;
        const float vden = sqrtf(rvv + 1e-8f), vmean = rvm;
        const float vf_half = g.hp.vf_coef * 0.5f;
        const float R = rs0, Vo = rs1;
        const float v = vsum + rbc;
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
        dg1 = 0.5f * vl;
        gm0 = vf_half * dnv * invM / vden;         // d loss / d v_i  (slot 0)
      }
#pragma unroll
      for (int kk = 0; kk < 4; ++kk) sG[rr * 16 + pp + 4 * kk] = (pp == 0 && kk == 0) ? gm0 : 0.0f, sMu[rr * 16 + pp + 4 * kk] = 0.0f;
      if (pp == 0) {
#pragma unroll
        for (int e = 0; e < 8; ++e) sD[rr * 8 + e] = e == 1 ? dg1 : 0.0f;
      }
    }
    __syncthreads();
    FH_TL(3);
    // C: dW4c[c] = sum_r G[r] H[r][c], one thread per column, rows ascending (the matrix form's chain: row 4 s + g)
    if (tid < HL) {
      float acc = 0.0f;
#pragma unroll 8
      for (int r = 0; r < BM; ++r) acc = __builtin_fmaf(sG[r * 16], Hs[r * LD + tid], acc);
      g.part_w[(int64_t)prow * (A + 1) * HL + (int64_t)A * HL + tid] = acc;      // row A = dW4c
    }
    FH_TL(4);
    // B + stream-out: dZ[r][c] = fl(G[r] w[c]) * elu'(H[r][c]); a wave writes whole rows (1 KB / 512 B contiguous)
    if constexpr (PREC == 3) {                   // bf16-stored dZ: 8 columns per lane
      constexpr int C8 = HL / 8;
      constexpr int RPP = 256 / C8;
      const int c8 = tid % C8, r_in = tid / C8;
      float wv[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) wv[e] = sW[8 * c8 + e];
#pragma unroll 4
      for (int rb = 0; rb < BM; rb += RPP) {
        const int r = rb + r_in;
        if (r < rows) {
          const float gr = sG[r * 16];
          float o[8];
#pragma unroll
          for (int e = 0; e < 8; ++e) {
            const float hv = Hs[r * LD + 8 * c8 + e];
            o[e] = __builtin_fmaf(gr, wv[e], 0.0f) * (hv > 0.0f ? 1.0f : hv + 1.0f);
          }
          store8_bf16_wt(reinterpret_cast<uint16_t*>(g.dZc) + (i0 + r) * HL + 8 * c8, o);
        }
      }
    } else {
      constexpr int C4 = HL / 4;                 // float4 chunks per row
      constexpr int RPP = 256 / C4;              // rows per pass of the workgroup
      const int c4 = tid % C4, r_in = tid / C4;
      const float4 w4 = *reinterpret_cast<const float4*>(sW + 4 * c4);
      const float wv[4] = {w4.x, w4.y, w4.z, w4.w};
#pragma unroll 4
      for (int rb = 0; rb < BM; rb += RPP) {
        const int r = rb + r_in;
        if (r < rows) {
          const float gr = sG[r * 16];
          const float4 h4 = *reinterpret_cast<const float4*>(Hs + r * LD + 4 * c4);
          const float hv[4] = {h4.x, h4.y, h4.z, h4.w};
          float o[4];
#pragma unroll
          for (int e = 0; e < 4; ++e) o[e] = __builtin_fmaf(gr, wv[e], 0.0f) * (hv[e] > 0.0f ? 1.0f : hv[e] + 1.0f);
          store_vec_wt<4>(g.dZc + (i0 + r) * HL + 4 * c4, o);
        }
      }
    }
    FH_TL(5);
  } else {

  // ---- A: head outputs.  MFMA step (blk, s) of lane-half h contracts k = 8 blk + 4 h + s - the same permutation on
  //         both operands (gemm_body's K-contiguous fragments)
  {
    __syncthreads();                                     // H tile complete
#if FWD_HEAD_MFMA16
    using f32x4 = __attribute__((ext_vector_type(4))) float;
    f32x4 c[4];                                          // rows 16 t + 4 g4 + reg, head output c16
#pragma unroll
    for (int t = 0; t < 4; ++t) c[t] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int kb = 0; kb < KQ / 16; ++kb) {
      float4 a[4];
#pragma unroll
      for (int t = 0; t < 4; ++t) a[t] = *reinterpret_cast<const float4*>(Hs + (16 * t + c16) * LD + q * KQ + 16 * kb + 4 * g4);
#pragma unroll
      for (int t = 0; t < 4; ++t) c[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[t].x, bw[kb].x, c[t], 0, 0, 0);
#pragma unroll
      for (int t = 0; t < 4; ++t) c[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[t].y, bw[kb].y, c[t], 0, 0, 0);
#pragma unroll
      for (int t = 0; t < 4; ++t) c[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[t].z, bw[kb].z, c[t], 0, 0, 0);
#pragma unroll
      for (int t = 0; t < 4; ++t) c[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[t].w, bw[kb].w, c[t], 0, 0, 0);
    }
    // the four contraction quarters in fixed order, two rounds: sMu = q0 + q1, sG = q2 + q3 (sG is free until the row
    // math writes it); the row math adds the two halves
    for (int w = 0; w < 2; ++w) {
      if ((q & 1) == w) {
        float* half = (q >> 1) ? sG : sMu;
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            float* d = half + (16 * t + 4 * g4 + r) * 16 + c16;
            *d = (w == 0 ? 0.0f : *d) + c[t][r];
          }
      }
      __syncthreads();
    }
  }
#else
    f32x16 c0, c1;
#pragma unroll
    for (int r = 0; r < 16; ++r) c0[r] = 0.0f, c1[r] = 0.0f;
#pragma unroll
    for (int kb = 0; kb < KQ / 8; ++kb) {
      const float4 a0 = *reinterpret_cast<const float4*>(Hs + l31 * LD + q * KQ + 8 * kb + 4 * h);
      const float4 a1 = *reinterpret_cast<const float4*>(Hs + (32 + l31) * LD + q * KQ + 8 * kb + 4 * h);
      c0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a0.x, bw[kb].x, c0, 0, 0, 0);
      c1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a1.x, bw[kb].x, c1, 0, 0, 0);
      c0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a0.y, bw[kb].y, c0, 0, 0, 0);
      c1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a1.y, bw[kb].y, c1, 0, 0, 0);
      c0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a0.z, bw[kb].z, c0, 0, 0, 0);
      c1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a1.z, bw[kb].z, c1, 0, 0, 0);
      c0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a0.w, bw[kb].w, c0, 0, 0, 0);
      c1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a1.w, bw[kb].w, c1, 0, 0, 0);
    }
    // the four contraction quarters in fixed order, two rounds: sMu = q0 + q1, sG = q2 + q3 (sG is free until the row
    // math writes it); the row math adds the two halves
    for (int w = 0; w < 2; ++w) {
      if ((q & 1) == w && l31 < 16) {
        float* half = (q >> 1) ? sG : sMu;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int row = (r & 3) + 8 * (r >> 2) + 4 * h;
          float* d0 = half + row * 16 + l31;
          float* d1 = half + (32 + row) * 16 + l31;
          *d0 = (w == 0 ? 0.0f : *d0) + c0[r];
          *d1 = (w == 0 ? 0.0f : *d1) + c1[r];
        }
      }
      __syncthreads();
    }
  }

#endif

  FH_TL(2);
  // ---- row math: four threads per row
  {
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
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) sG[r * 16 + pp + 4 * kk] = gm[kk], sMu[r * 16 + pp + 4 * kk] = gl[kk];
    if (pp == 0) {
#pragma unroll
      for (int e = 0; e < 8; ++e) sD[r * 8 + e] = dg[e];
    }
  }
  __syncthreads();

  FH_TL(3);
  // ---- C: head weight gradient of the tile, dWh[k][c] = sum_r G[r][k] H[r][c]; wave q owns TNC column tiles
  {
#if FWD_HEAD_MFMA16
    // D tile = 16 head outputs x 16 columns; contraction over the tile's 64 rows, 4 per instruction (row 4 step + g4)
    using f32x4 = __attribute__((ext_vector_type(4))) float;
    constexpr int TC = HL / 64;                          // 16-column tiles per wave (a wave owns HL / 4 columns)
    f32x4 cc[TC];
#pragma unroll
    for (int t = 0; t < TC; ++t) cc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
    const int col0 = q * (HL / 4);
#pragma unroll 4
    for (int s = 0; s < BM / 4; ++s) {
      const int r = 4 * s + g4;
      const float a = sG[r * 16 + c16];
#pragma unroll
      for (int t = 0; t < TC; ++t)
        cc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, Hs[r * LD + col0 + 16 * t + c16], cc[t], 0, 0, 0);
    }
    float* pw = g.part_w + (int64_t)prow * (A + 1) * HL + (net == 1 ? 0 : (int64_t)A * HL);   // rows 0..A-1 = dW4a, row A = dW4c
#pragma unroll
    for (int t = 0; t < TC; ++t)
#pragma unroll
      for (int r = 0; r < 4; ++r) {                      // accumulator row 4 g4 + r = head output
        const int k = 4 * g4 + r;
        if (k < KH) pw[k * HL + col0 + 16 * t + c16] = cc[t][r];
      }
  }
#else
    f32x16 cc[TNC];
#pragma unroll
    for (int t = 0; t < TNC; ++t)
#pragma unroll
      for (int r = 0; r < 16; ++r) cc[t][r] = 0.0f;
    const int col0 = q * (HL / 4);
#pragma unroll 8
    for (int s = 0; s < BM / 2; ++s) {
      const int r = 2 * s + h;
      const float a = l31 < 16 ? sG[r * 16 + l31] : 0.0f;
#pragma unroll
      for (int t = 0; t < TNC; ++t)
        cc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, Hs[r * LD + col0 + 32 * t + l31], cc[t], 0, 0, 0);
    }
    float* pw = g.part_w + (int64_t)prow * (A + 1) * HL + (net == 1 ? 0 : (int64_t)A * HL);   // rows 0..A-1 = dW4a, row A = dW4c
#pragma unroll
    for (int t = 0; t < TNC; ++t)
#pragma unroll
      for (int r = 0; r < 8; ++r) {                      // accumulator rows 0..15 = head outputs
        const int k = (r & 3) + 8 * (r >> 2) + 4 * h;
        if (k < KH) pw[k * HL + col0 + 32 * t + l31] = cc[t][r];
      }
  }
#endif
  __syncthreads();                                       // every read of H is done: step B overwrites it
  FH_TL(4);

  // ---- B: dZ = (G . Wh) * elu'(H), in place
  {
    f32x16 cb[TNB];
#pragma unroll
    for (int tn = 0; tn < TNB; ++tn)
#pragma unroll
      for (int r = 0; r < 16; ++r) cb[tn][r] = 0.0f;
#pragma unroll
    for (int blk = 0; blk < 2; ++blk) {
      const float4 a4 = *reinterpret_cast<const float4*>(sG + (32 * wm + l31) * 16 + 8 * blk + 4 * h);
      const float av[4] = {a4.x, a4.y, a4.z, a4.w};
#pragma unroll
      for (int s = 0; s < 4; ++s)
#pragma unroll
        for (int tn = 0; tn < TNB; ++tn)
          cb[tn] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[s], bwB[tn][blk][s], cb[tn], 0, 0, 0);
    }
#pragma unroll
    for (int tn = 0; tn < TNB; ++tn)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = 32 * wm + (r & 3) + 8 * (r >> 2) + 4 * h;
        float* hp = Hs + row * LD + wn * (HL / 2) + 32 * tn + l31;
        const float hv = *hp;
        *hp = cb[tn][r] * (hv > 0.0f ? 1.0f : hv + 1.0f);      // elu'(z) = 1 (z>0) | elu(z) + 1
      }
  }
  // every wave streams out its own 32 x HL/2 region of dZ (LDS operations of a wave execute in order: no workgroup
  // barrier between its in-place writes and these reads)
  __builtin_amdgcn_wave_barrier();
  FH_TL(5);
  if constexpr (PREC == 3) {                   // bf16-stored dZ: 8 columns per lane
    constexpr int C8 = HL / 16;                // 8-column chunks per region row
    constexpr int RPI = 64 / C8;
    const int c8 = lane % C8, r_in = lane / C8;
    uint16_t* dZ = reinterpret_cast<uint16_t*>(net == 1 ? g.dZa : g.dZc);
#pragma unroll 4
    for (int rb = 0; rb < 32; rb += RPI) {
      const int r = 32 * wm + rb + r_in;
      if (r < rows) {
        const float4 v0 = *reinterpret_cast<const float4*>(Hs + r * LD + wn * (HL / 2) + 8 * c8);
        const float4 v1 = *reinterpret_cast<const float4*>(Hs + r * LD + wn * (HL / 2) + 8 * c8 + 4);
        const float o[8] = {v0.x, v0.y, v0.z, v0.w, v1.x, v1.y, v1.z, v1.w};
        store8_bf16_wt(dZ + (i0 + r) * HL + wn * (HL / 2) + 8 * c8, o);
      }
    }
  } else {
    constexpr int C4 = HL / 8;                 // float4 chunks per region row
    constexpr int RPI = 64 / C4;               // region rows per store instruction
    const int c4 = lane % C4, r_in = lane / C4;
    float* dZ = net == 1 ? g.dZa : g.dZc;
#pragma unroll 4
    for (int rb = 0; rb < 32; rb += RPI) {
      const int r = 32 * wm + rb + r_in;
      if (r < rows) {
        const float4 v = *reinterpret_cast<const float4*>(Hs + r * LD + wn * (HL / 2) + 4 * c4);
        const float o[4] = {v.x, v.y, v.z, v.w};
        store_vec_wt<4>(dZ + (i0 + r) * HL + wn * (HL / 2) + 4 * c4, o);
      }
    }
  }
  }   // matrix-form epilogue
  // ---- scalars of the tile: bias / logstd gradients, diagnostics.  16 row groups of 4 rows, combined in fixed order
  //      through LDS (the H / dZ tile is free again once every row has been streamed out)
  __syncthreads();
  FH_TL(6);
  {
    float* red = Hs;                                    // [16 groups][40]: 16 db, 16 dlogstd, 8 diag
    const int k = tid & 15, grp = tid >> 4;
    float db = 0.0f, dl = 0.0f, dd = 0.0f;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int r = 4 * grp + j;
      db += sG[r * 16 + k], dl += sMu[r * 16 + k];
      if (k < 8) dd += sD[r * 8 + k];
    }
    red[grp * 40 + k] = db, red[grp * 40 + 16 + k] = dl;
    if (k < 8) red[grp * 40 + 32 + k] = dd;
  }
  __syncthreads();
  float* ps = g.part_s + (int64_t)prow * NS;
  if (tid < 40) {
    float v = 0.0f;
#pragma unroll
    for (int grp = 0; grp < 16; ++grp) v += Hs[grp * 40 + tid];
    if (tid < 16) {
      if (net == 1) { if (tid < A) ps[tid] = v; }          // db4a[k]
      else if (tid == 0) ps[A] = v;                        // db4c
    } else if (tid < 32) {
      if (net == 1 && tid - 16 < A) ps[A + 1 + tid - 16] = v;   // dlogstd[k]
    } else {
      ps[2 * A + 1 + tid - 32] = v;                        // diagnostics
    }
  }
  FH_TL(7);
}

#ifdef FWD_HEAD_TL
extern "C" int catppo_debug_fwd_head_tl(void* buf) {     // timeline builds only: not part of include/catppo.h
  unsigned long long* pbuf = static_cast<unsigned long long*>(buf);
  return hipMemcpyToSymbol(HIP_SYMBOL(g_fhtl), &pbuf, sizeof(pbuf)) == hipSuccess ? 0 : -1;
}
#endif
