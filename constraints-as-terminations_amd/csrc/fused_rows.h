// Row-tile fused actor-critic pass (included inside mlp.hip's anonymous namespace).
//
// One 512-thread workgroup owns R rows of ONE network (critic or actor) and walks the whole chain
// with the activations of the tile resident in LDS:
//     X tile -> [hidden layer: fp32-MFMA GEMM + bias + ELU] x L -> head (+ PPO loss and its analytic
//     gradient) -> [data-gradient GEMM x ELU'] x (L-1)
// Only what the later weight-gradient GEMMs need is written to HBM (H_l and dZ_l tiles); no layer
// ever re-reads its input from HBM, and the per-layer launch / prologue / store-drain costs of the
// layer-wise path (six launches per minibatch) collapse into one launch.  Weights stream from L2
// through double-buffered LDS slabs of 16 contraction steps, one barrier per slab; the A operand is
// the LDS-resident tile (row stride MAXW+4 floats: 16-B-slot index = row mod 16 for every ds_read_b128
// lane group => conflict-free).  Eight waves: WR = R/32 row groups x WC = 8/WR column groups, every
// wave owns up to two 32x32 MFMA accumulators per layer.
//
//   TRAIN = true   replaces forward_hidden + head_loss_kernel + the two data-gradient GEMMs
//   TRAIN = false  rollout: forward + Gaussian sample / log-prob (actor task) or value (critic task)
#pragma once

constexpr int kFusedVS = 15;            // LDS column of the critic's d loss / d v in the per-row gradient tile

struct FusedArgs {
  const float* x;        // [M, Dp] (gathered) observations
  const float* params;   // flat parameter buffer
  int n_hidden, A, Dp;
  int hidden[CATPPO_MAX_HIDDEN];
  int64_t off_w[2][CATPPO_MAX_HIDDEN + 1], off_b[2][CATPPO_MAX_HIDDEN + 1], off_logstd;
  float* H[2][CATPPO_MAX_HIDDEN];    // TRAIN: activations written for the weight-gradient GEMMs
  float* dZ[2][CATPPO_MAX_HIDDEN];   // TRAIN: pre-activation gradients written for the weight-gradient GEMMs
  int64_t M;
  // TRAIN head inputs (gathered minibatch)
  const float *act, *oldlogp, *adv, *ret_n, *val_n;
  const double* adv_part;
  int n_adv_part;
  const float* adv_stats;
  const float *vrms_mean, *vrms_var;
  float *part_w, *part_s;            // [n_tiles][(A+1)*HL], [n_tiles][2A+1+8]
  catppo_ppo_hparams hp;
  // rollout head
  const float *eps, *given;
  float *action, *logprob, *value;
};

template <int R, int MAXW>
constexpr size_t fused_lds_bytes() {   // independent of the wave count
  return sizeof(float) * ((size_t)R * (MAXW + 4) + 2 * (size_t)MAXW * 20 + (size_t)R * 16 + 16 * 8 + 64);
}

template <int R, int MAXW, int NW, bool TRAIN>
__global__ __launch_bounds__(NW * 64) void fused_rows_kernel(const FusedArgs g) {
  using gemm::f32x16;
  constexpr int kFusedThreads = NW * 64;
  constexpr int TS = MAXW + 4;                 // tile row stride (floats)
  constexpr int SLAB = MAXW * 20;              // floats per weight-slab buffer
  constexpr int WR = R / 32, WC = NW / WR;     // wave grid
  constexpr int MAXT = MAXW / 32 / WC;         // 32x32 accumulators per wave (2)
  constexpr int NST = MAXW * 4 / kFusedThreads;  // float4 staging registers per thread per slab
  constexpr int NCMAX = MAXW / 64;             // head: columns per lane (lane, lane+64, ...)
  constexpr int RPW = R / NW;                  // head: rows per wave
  static_assert(MAXT >= 1 && NST >= 1, "bad tile geometry");

  extern __shared__ __attribute__((aligned(16))) float lds[];   // everything dynamic: keeps 16-B alignment
  float* tile = lds;                   // [R][TS]
  float* slab = tile + R * TS;         // [2][SLAB]
  float* sG = slab + 2 * SLAB;         // [R][16]   per-row head gradients (d mu_k | d v at kFusedVS)
  float* sW = sG + R * 16;             // [NW][16]  per-wave scalar partials (NW <= 8)
  float* sS = sW + 8 * 16;             // [64]      block scalars (adv mean / den, ...)

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int l31 = lane & 31, h = lane >> 5;
  const int wr = wave % WR, wc = wave / WR;
  const int net = blockIdx.y;                  // 0 = critic, 1 = actor
  const int64_t row0 = (int64_t)blockIdx.x * R;
  const int nl = g.n_hidden, A = g.A;

  // ------------------------------------------------------------------ X tile -> LDS
  {
    const int q4 = g.Dp / 4;
    for (int f = tid; f < R * q4; f += kFusedThreads) {
      const int r = f / q4, q = f - r * q4;
      float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
      if (row0 + r < g.M) v = reinterpret_cast<const float4*>(g.x + (row0 + r) * g.Dp)[q];
      *reinterpret_cast<float4*>(tile + r * TS + 4 * q) = v;
    }
  }
  __syncthreads();

  f32x16 acc[MAXT];
  float4 st0[NST], st1[NST];   // weight slabs in flight: global loads run TWO slabs ahead of the MFMAs

  // One GEMM of the chain: acc = tile[R x Kd] . B, with B streamed from `W` (row-major [*, ldw]).
  //   B_KC = true  (forward):        B(k, n) = W[n*ldw + k],  n < Nd   slab image [Nd][20]
  //   B_KC = false (data gradient):  B(k, n) = W[k*ldw + n],  n < Nd   slab image [16][Nd]
  // The L2 round trip of a slab (~1-2 us with every CU walking the same weights) is longer than the
  // 16 MFMAs per wave that consume one, so slab s+2 is requested before slab s is multiplied.
  auto run_gemm = [&](const float* __restrict__ W, int ldw, int Kd, int Nd, auto b_kc) {
    constexpr bool B_KC = decltype(b_kc)::value;
    const int n_units = Nd * 4;                       // float4 per slab in both layouts
    auto gload = [&](int s, float4 (&st)[NST]) {
#pragma unroll
      for (int q = 0; q < NST; ++q) {
        const int f = tid + q * kFusedThreads;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (f < n_units) {
          if (B_KC) {
            const int n = f >> 2, kq = f & 3;
            v = *reinterpret_cast<const float4*>(W + (int64_t)n * ldw + 16 * s + 4 * kq);
          } else {
            const int per = Nd >> 2, kr = f / per, cq = f - kr * per;
            v = *reinterpret_cast<const float4*>(W + (int64_t)(16 * s + kr) * ldw + 4 * cq);
          }
        }
        st[q] = v;
      }
    };
    auto lstore = [&](int buf, const float4 (&st)[NST]) {
      float* b = slab + buf * SLAB;
#pragma unroll
      for (int q = 0; q < NST; ++q) {
        const int f = tid + q * kFusedThreads;
        if (f < n_units) {
          if (B_KC) {
            const int n = f >> 2, kq = f & 3;
            *reinterpret_cast<float4*>(b + n * 20 + 4 * kq) = st[q];
          } else {
            const int per = Nd >> 2, kr = f / per, cq = f - kr * per;
            *reinterpret_cast<float4*>(b + kr * Nd + 4 * cq) = st[q];
          }
        }
      }
    };
    const float* arow = tile + (32 * wr + l31) * TS + 4 * h;
    auto compute = [&](int s) {
      const float* b = slab + (s & 1) * SLAB;
#pragma unroll
      for (int blk = 0; blk < 2; ++blk) {
        const float4 av = *reinterpret_cast<const float4*>(arow + 16 * s + 8 * blk);
        const float af[4] = {av.x, av.y, av.z, av.w};
#pragma unroll
        for (int t = 0; t < MAXT; ++t) {
          const int ct = wc + WC * t;
          if (ct * 32 < Nd) {                           // wave-uniform
            float bf[4];
            if (B_KC) {
              const float4 bv = *reinterpret_cast<const float4*>(b + (32 * ct + l31) * 20 + 8 * blk + 4 * h);
              bf[0] = bv.x, bf[1] = bv.y, bf[2] = bv.z, bf[3] = bv.w;
            } else {
#pragma unroll
              for (int q = 0; q < 4; ++q) bf[q] = b[(8 * blk + 4 * h + q) * Nd + 32 * ct + l31];
            }
#pragma unroll
            for (int q = 0; q < 4; ++q) acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[q], bf[q], acc[t], 0, 0, 0);
          }
        }
      }
    };
#pragma unroll
    for (int t = 0; t < MAXT; ++t)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[t][r] = 0.0f;
    const int n_slabs = Kd >> 4;
    gload(0, st0);
    if (n_slabs > 1) gload(1, st1);
    lstore(0, st0);
    __syncthreads();
    // slabs are handled in (even, odd) pairs so that the two register sets are addressed statically
    for (int s = 0; s < n_slabs; s += 2) {
      if (s + 2 < n_slabs) gload(s + 2, st0);
      compute(s);
      if (s + 1 < n_slabs) lstore(1, st1);
      __syncthreads();
      if (s + 1 < n_slabs) {
        if (s + 3 < n_slabs) gload(s + 3, st1);
        compute(s + 1);
        if (s + 2 < n_slabs) lstore(0, st0);
        __syncthreads();
      }
    }
  };

  // ------------------------------------------------------------------ forward chain
  for (int l = 0; l < nl; ++l) {
    const int in = l == 0 ? g.Dp : g.hidden[l - 1], out = g.hidden[l];
    run_gemm(g.params + g.off_w[net][l], in, in, out, std::true_type{});
    const float* bias = g.params + g.off_b[net][l];
    float* Hout = TRAIN ? g.H[net][l] : nullptr;
    // all waves are past the last slab barrier: the tile can be overwritten in place
#pragma unroll
    for (int t = 0; t < MAXT; ++t) {
      const int ct = wc + WC * t;
      if (ct * 32 < out) {
        const int col = 32 * ct + l31;
        const float bv = bias[col];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int row = 32 * wr + (r & 3) + 8 * (r >> 2) + 4 * h;
          const float v = gemm::elu_f(acc[t][r] + bv);
          tile[row * TS + col] = v;
          if (TRAIN) {
            if (row0 + row < g.M) Hout[(row0 + row) * out + col] = v;
          }
        }
      }
    }
    __syncthreads();
  }

  // ------------------------------------------------------------------ heads
  const int HL = g.hidden[nl - 1];
  const int NC = HL >> 6;
  const int slot = ((lane >> 5) & 1) * 8 + ((lane >> 4) & 1) * 4 + ((lane >> 3) & 1) * 2 + ((lane >> 2) & 1);
  float* sW4 = slab;                                   // head weights staged in the (idle) slab region
  if (net == 1) {
    for (int o = tid; o < 15 * HL; o += kFusedThreads) sW4[o] = o < A * HL ? g.params[g.off_w[1][nl] + o] : 0.0f;
  } else {
    for (int o = tid; o < HL; o += kFusedThreads) sW4[o] = g.params[g.off_w[0][nl] + o];
  }
  if (TRAIN) {
    for (int o = tid; o < R * 16; o += kFusedThreads) sG[o] = 0.0f;
    if (net == 1 && wave == 0) {
      // advantage statistics over the minibatch (ppo.py:314-318): mean, unbiased std
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
          double var = (a2 - n * mean * mean) / (n - 1.0);
          if (var < 0.0) var = 0.0;
          sS[0] = (float)mean;
          sS[1] = (float)sqrt(var) + 1e-8f;
        }
      } else if (lane == 0) {
        sS[0] = g.adv_stats ? g.adv_stats[0] : 0.0f;
        sS[1] = g.adv_stats ? g.adv_stats[1] : 1.0f;
      }
    }
  }
  __syncthreads();

  if (!TRAIN) {
    // ---------------------------------------------------------------- rollout heads
    if (net == 1) {
      const bool mine = slot < A;
      const float sd = mine ? expf(g.params[g.off_logstd + slot]) : 1.0f;
      const float var = sd * sd, lsd = logf(sd);
      const float ba = mine ? g.params[g.off_b[1][nl] + slot] : 0.0f;
      for (int rr = 0; rr < RPW; ++rr) {
        const int r = wave * RPW + rr;
        const int64_t i = row0 + r;
        if (i >= g.M) break;
        float hrow[NCMAX], part[16];
#pragma unroll
        for (int c = 0; c < NCMAX; ++c) hrow[c] = c < NC ? tile[r * TS + lane + 64 * c] : 0.0f;
#pragma unroll
        for (int k = 0; k < 15; ++k) {
          float d = 0.0f;
#pragma unroll
          for (int c = 0; c < NCMAX; ++c)
            if (c < NC) d = fmaf(hrow[c], sW4[k * HL + lane + 64 * c], d);
          part[k] = d;
        }
        part[15] = 0.0f;
        const float mu = reduce16(part, lane) + ba;
        float a = mu;
        if (mine && g.given != nullptr) a = g.given[i * A + slot];
        else if (mine && g.eps != nullptr) a = mu + sd * g.eps[i * A + slot];   // Normal.sample()
        const float diff = a - mu;
        const float term = mine ? (-(diff * diff) / (2.0f * var) - lsd - kHalfLog2Pi) : 0.0f;
        float lp = 0.0f;
#pragma unroll
        for (int k = 0; k < 15; ++k) lp += lane_bcast(term, slot_lane(k));
        if (mine && (lane & 3) == 0) g.action[i * A + slot] = a;
        if (lane == 0) g.logprob[i] = lp;
      }
    } else {
      const float bc = g.params[g.off_b[0][nl]];
      for (int rr = 0; rr < RPW; ++rr) {
        const int r = wave * RPW + rr;
        const int64_t i = row0 + r;
        if (i >= g.M) break;
        float d = 0.0f;
#pragma unroll
        for (int c = 0; c < NCMAX; ++c)
          if (c < NC) d = fmaf(tile[r * TS + lane + 64 * c], sW4[lane + 64 * c], d);
        d = wave_sum(d) + bc;
        if (lane == 0) g.value[i] = d;
      }
    }
    return;
  }

  // ------------------------------------------------------------------ training heads: loss + gradient
  const int NS = 2 * A + 1 + kHeadDiag;
  const int rows_valid = (int)((g.M - row0) < R ? (g.M - row0) : R);
  const float invM = g.hp.inv_global_batch, clipc = g.hp.clip_coef;
  float* dZout = g.dZ[net][nl - 1];
  float w_s[8];                                        // per-wave scalars (meaning depends on the net)
#pragma unroll
  for (int q = 0; q < 8; ++q) w_s[q] = 0.0f;
  float gls = 0.0f;

  if (net == 1) {
    const float adv_mean = sS[0], adv_den = sS[1];
    const bool norm_adv = g.hp.norm_adv != 0;
    const float ent_coef_m = g.hp.ent_coef * invM;
    const bool mine = slot < A;
    const bool leader = mine && (lane & 3) == 0;
    const float sd = mine ? expf(g.params[g.off_logstd + slot]) : 1.0f;
    const float var = sd * sd, lsd = logf(sd);
    const float ba = mine ? g.params[g.off_b[1][nl] + slot] : 0.0f;
    float ent_row = 0.0f;
    {
      const float e = mine ? kEntConst + lsd : 0.0f;
#pragma unroll
      for (int k = 0; k < 15; ++k) ent_row += lane_bcast(e, slot_lane(k));
    }
    for (int rr = 0; rr < RPW; ++rr) {
      const int r = wave * RPW + rr;
      if (r >= rows_valid) break;
      const int64_t i = row0 + r;
      float hrow[NCMAX], part[16];
#pragma unroll
      for (int c = 0; c < NCMAX; ++c) hrow[c] = c < NC ? tile[r * TS + lane + 64 * c] : 0.0f;
      const float a_taken = mine ? g.act[i * A + slot] : 0.0f;
      const float oldlogp = g.oldlogp[i], adv_raw = g.adv[i];
#pragma unroll
      for (int k = 0; k < 15; ++k) {
        float d = 0.0f;
#pragma unroll
        for (int c = 0; c < NCMAX; ++c)
          if (c < NC) d = fmaf(hrow[c], sW4[k * HL + lane + 64 * c], d);
        part[k] = d;
      }
      part[15] = 0.0f;
      const float mu = reduce16(part, lane) + ba;
      const float diff = mine ? a_taken - mu : 0.0f;
      const float term = mine ? -(diff * diff) / (2.0f * var) - lsd - kHalfLog2Pi : 0.0f;
      float newlogp = 0.0f;
#pragma unroll
      for (int k = 0; k < 15; ++k) newlogp += lane_bcast(term, slot_lane(k));
      const float logratio = newlogp - oldlogp;
      const float ratio = expf(logratio);
      w_s[5] += -logratio;                                       // old_approx_kl
      w_s[4] += (ratio - 1.0f) - logratio;                       // approx_kl
      w_s[6] += fabsf(ratio - 1.0f) > clipc ? 1.0f : 0.0f;       // clipfrac
      const float adv = norm_adv ? (adv_raw - adv_mean) / adv_den : adv_raw;
      const float rc = ratio < 1.0f - clipc ? 1.0f - clipc : (ratio > 1.0f + clipc ? 1.0f + clipc : ratio);
      const float pg1 = -adv * ratio, pg2 = -adv * rc;
      const bool inside = ratio >= 1.0f - clipc && ratio <= 1.0f + clipc;
      const float dr_tie = 0.5f * -adv + (inside ? 0.5f * -adv : 0.0f);
      const float dr = pg1 > pg2 ? -adv : (pg1 < pg2 ? (inside ? -adv : 0.0f) : dr_tie);
      w_s[0] += pg1 > pg2 ? pg1 : pg2;                           // pg loss
      w_s[2] += ent_row;                                         // entropy
      const float g_logp = dr * ratio * invM;
      const float gm = mine ? g_logp * diff / var : 0.0f;        // d loss / d mu_ik (lanes of slot k)
      if (leader) {
        gls += g_logp * (diff * diff / var - 1.0f) - ent_coef_m;
        sG[r * 16 + slot] = gm;
      }
    }
  } else {
    const float vden = sqrtf(g.vrms_var[0] + 1e-8f), vmean = g.vrms_mean[0];
    const bool clip_vloss = g.hp.clip_vloss != 0;
    const float vf_half = g.hp.vf_coef * 0.5f;
    const float bc = g.params[g.off_b[0][nl]];
    for (int rr = 0; rr < RPW; ++rr) {
      const int r = wave * RPW + rr;
      if (r >= rows_valid) break;
      const int64_t i = row0 + r;
      float d = 0.0f;
#pragma unroll
      for (int c = 0; c < NCMAX; ++c)
        if (c < NC) d = fmaf(tile[r * TS + lane + 64 * c], sW4[lane + 64 * c], d);
      const float v = wave_sum(d) + bc;
      const float R_ = g.ret_n[i], Vo = g.val_n[i];
      const float nv = (v - vmean) / vden;                       // value_rms(newvalue, update=False)
      const float e1 = nv - R_;
      const float vl1 = e1 * e1;
      const float dl = nv - Vo;
      const float cl = dl < -clipc ? -clipc : (dl > clipc ? clipc : dl);
      const float e2 = (Vo + cl) - R_;
      const float vl2 = e2 * e2;
      const bool in2 = dl >= -clipc && dl <= clipc;
      const float dnv_c = vl1 > vl2 ? 2.0f * e1 : (vl1 < vl2 ? (in2 ? 2.0f * e2 : 0.0f) : e1 + (in2 ? e2 : 0.0f));
      const float vl = clip_vloss ? (vl1 > vl2 ? vl1 : vl2) : vl1;
      const float dnv = clip_vloss ? dnv_c : 2.0f * e1;
      w_s[1] += 0.5f * vl;                                       // v loss
      const float g_v = vf_half * dnv * invM / vden;             // d loss / d v_i
      if (lane == 0) sG[r * 16 + kFusedVS] = g_v;
    }
  }
  __syncthreads();

  // ---- head weight-gradient partial of this tile: dW4[k][c] = sum_r G[r][k] * H[r][c]
  //      thread = (column c2, row group rg); the NG row groups are combined through LDS in fixed order
  {
    int NG = kFusedThreads / HL;                       // HL in {64,...,512}
    if (NG > 1 + SLAB / (16 * HL)) NG = 1 + SLAB / (16 * HL);   // the combine buffer is one slab buffer
    const int c2 = tid % HL, rg = tid / HL;
    float a16[16];
#pragma unroll
    for (int k = 0; k < 16; ++k) a16[k] = 0.0f;
    if (rg < NG) {
      for (int r = rg; r < rows_valid; r += NG) {
        const float hv = tile[r * TS + c2];
        const float4* gp = reinterpret_cast<const float4*>(sG + r * 16);
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const float4 gv = gp[q];
          a16[4 * q] = fmaf(gv.x, hv, a16[4 * q]);
          a16[4 * q + 1] = fmaf(gv.y, hv, a16[4 * q + 1]);
          a16[4 * q + 2] = fmaf(gv.z, hv, a16[4 * q + 2]);
          a16[4 * q + 3] = fmaf(gv.w, hv, a16[4 * q + 3]);
        }
      }
    }
    // combine the row groups: group rg > 0 publishes, group 0 adds in order (slab region is free except
    // for sW4 which lives in its first 15*HL floats: use the second slab buffer)
    float* comb = slab + SLAB;                         // [NG-1][16][HL]  <= 16*512 floats per group
    if (rg > 0 && rg < NG) {
#pragma unroll
      for (int k = 0; k < 16; ++k) comb[((rg - 1) * 16 + k) * HL + c2] = a16[k];
    }
    __syncthreads();
    if (rg == 0) {
      for (int gidx = 1; gidx < NG; ++gidx)
#pragma unroll
        for (int k = 0; k < 16; ++k) a16[k] += comb[((gidx - 1) * 16 + k) * HL + c2];
      float* pw = g.part_w + (int64_t)blockIdx.x * (A + 1) * HL;   // rows 0..A-1 = dW4a, row A = dW4c
      if (net == 1) {
#pragma unroll
        for (int k = 0; k < 15; ++k)
          if (k < A) pw[k * HL + c2] = a16[k];
      } else {
        pw[A * HL + c2] = a16[kFusedVS];
      }
    }
  }
  // per-block scalars: bias gradients = column sums of G; dlogstd / diagnostics through the wave partials
  if (lane == 63) {
#pragma unroll
    for (int q = 0; q < 8; ++q) sW[wave * 16 + q] = w_s[q];
  }
  __syncthreads();
  {
    float* ps = g.part_s + (int64_t)blockIdx.x * NS;
    if (net == 1) {
      if (tid < A) {
        float s = 0.0f;
        for (int r = 0; r < rows_valid; ++r) s += sG[r * 16 + tid];
        ps[tid] = s;                                           // db4a
      }
      if (tid >= 64 && tid < 64 + kHeadDiag) {
        const int q = tid - 64;
        if (q != 1) {                                          // slot 1 (v loss) belongs to the critic task
          float s = 0.0f;
          for (int w = 0; w < NW; ++w) s += sW[w * 16 + q];
          ps[2 * A + 1 + q] = s;
        }
      }
    } else {
      if (tid == 0) {
        float s = 0.0f;
        for (int r = 0; r < rows_valid; ++r) s += sG[r * 16 + kFusedVS];
        ps[A] = s;                                             // db4c
      }
      if (tid == 64) {
        float s = 0.0f;
        for (int w = 0; w < NW; ++w) s += sW[w * 16 + 1];
        ps[2 * A + 1 + 1] = s;                                 // v loss
      }
    }
  }
  // dlogstd: leader lanes hold per-wave sums; fold the 8 waves in order through sW (second half)
  __syncthreads();
  if (net == 1) {
    for (int w = 0; w < NW; ++w) {
      if (wave == w && (lane & 3) == 0 && slot < A) sW[slot] = w == 0 ? gls : sW[slot] + gls;
      __syncthreads();
    }
    if (tid < A) g.part_s[(int64_t)blockIdx.x * NS + A + 1 + tid] = sW[tid];
  }
  __syncthreads();

  // ---- dZ of the last hidden layer, written in place over the H tile (each wave owns its rows)
  if (net == 1) {
    for (int rr = 0; rr < RPW; ++rr) {
      const int r = wave * RPW + rr;
      if (r >= rows_valid) break;
      float dh[NCMAX], hrow[NCMAX];
#pragma unroll
      for (int c = 0; c < NCMAX; ++c) dh[c] = 0.0f, hrow[c] = c < NC ? tile[r * TS + lane + 64 * c] : 0.0f;
#pragma unroll
      for (int k = 0; k < 15; ++k) {
        const float gk = sG[r * 16 + k];
#pragma unroll
        for (int c = 0; c < NCMAX; ++c)
          if (c < NC) dh[c] = fmaf(gk, sW4[k * HL + lane + 64 * c], dh[c]);
      }
#pragma unroll
      for (int c = 0; c < NCMAX; ++c)
        if (c < NC) {
          const float v = dh[c] * (hrow[c] > 0.0f ? 1.0f : hrow[c] + 1.0f);
          tile[r * TS + lane + 64 * c] = v;
          dZout[(row0 + r) * HL + lane + 64 * c] = v;
        }
    }
  } else {
    for (int rr = 0; rr < RPW; ++rr) {
      const int r = wave * RPW + rr;
      if (r >= rows_valid) break;
      const float gv = sG[r * 16 + kFusedVS];
#pragma unroll
      for (int c = 0; c < NCMAX; ++c)
        if (c < NC) {
          const float hv = tile[r * TS + lane + 64 * c];
          const float v = (gv * sW4[lane + 64 * c]) * (hv > 0.0f ? 1.0f : hv + 1.0f);
          tile[r * TS + lane + 64 * c] = v;
          dZout[(row0 + r) * HL + lane + 64 * c] = v;
        }
    }
  }
  // rows beyond M keep garbage in the tile; they only ever produce garbage rows that are never stored
  __syncthreads();

  // ------------------------------------------------------------------ data-gradient chain
  for (int l = nl - 1; l >= 1; --l) {
    const int out = g.hidden[l], in = g.hidden[l - 1];    // dZ_{l-1}[R x in] = dZ_l[R x out] . W_l[out x in]
    run_gemm(g.params + g.off_w[net][l], in, out, in, std::false_type{});
    const float* Hprev = g.H[net][l - 1];
    float* dZp = g.dZ[net][l - 1];
#pragma unroll
    for (int t = 0; t < MAXT; ++t) {
      const int ct = wc + WC * t;
      if (ct * 32 < in) {
        const int col = 32 * ct + l31;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int row = 32 * wr + (r & 3) + 8 * (r >> 2) + 4 * h;
          float v = 0.0f;
          if (row0 + row < g.M) {
            const float hact = Hprev[(row0 + row) * in + col];
            v = acc[t][r] * (hact > 0.0f ? 1.0f : hact + 1.0f);
            dZp[(row0 + row) * in + col] = v;
          }
          tile[row * TS + col] = v;
        }
      }
    }
    __syncthreads();
  }
}
