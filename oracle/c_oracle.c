/* ORACLE (test infrastructure, NOT product code): plain-C restatement of the two bit-exact
 * pieces of the hot path, built by __graft_entry__.build() with gcc -ffp-contract=off.
 *   cat_step_oracle : reference cat/constraint_manager.py:39-82,213-229 + cat/cat_env.py:102-107,118-121
 *   gae_oracle      : reference cleanrl/ppo.py:251-277
 * Checked against the reference-generated vectors in tests/test_oracle_golden.py
 * (test_c_oracle_*); only tests/, smoke() and bench.py's cpu_baseline leg may load it. */
#include <math.h>
#include <stdint.h>

static float nanmaxf(float m, float x) { return (x > m || x != x) ? x : m; }

/* cstr [N,K] row-major; term_off [n_terms+1]; term_dp [n_terms] = fl32(max_p-min_p); rm [K] in/out;
 * reward [N] in/out or NULL; reset [N] or NULL; outputs cstr_prob [N], dones [N] or NULL,
 * ep_viol/ep_prob [n_terms,N] in/out, probs [N,K] or NULL. */
void cat_step_oracle(const float* cstr, int64_t N, int K, const int32_t* term_off, int n_terms,
                     const float* term_dp, float min_p, float tau, float one_minus_tau, int first_call,
                     float* rm, float* reward, const uint8_t* reset, float* cstr_prob, float* dones,
                     float* ep_viol, float* ep_prob, float* probs) {
  for (int c = 0; c < K; ++c) {
    float m = cstr[c];
    for (int64_t i = 1; i < N; ++i) m = nanmaxf(m, cstr[i * K + c]);
    m = (m < 1e-6f) ? 1e-6f : m;
    if (first_call) {
      rm[c] = m;
    } else {
      float a = rm[c] * tau;
      float b = one_minus_tau * m;
      rm[c] = a + b;
    }
  }
  for (int64_t i = 0; i < N; ++i) {
    float pmax = 0.0f;
    for (int t = 0; t < n_terms; ++t) {
      float tm = 0.0f;
      for (int c = term_off[t]; c < term_off[t + 1]; ++c) {
        const float x = cstr[i * K + c];
        float p = 0.0f;
        if (x > 0.0f) {
          float q = x / rm[c];
          q = q < 0.0f ? 0.0f : (q > 1.0f ? 1.0f : q);
          const float s = q * term_dp[t];
          p = min_p + s;
        }
        if (probs) probs[i * K + c] = p;
        tm = (c == term_off[t]) ? p : nanmaxf(tm, p);
      }
      ep_viol[(int64_t)t * N + i] = ep_viol[(int64_t)t * N + i] + (tm > 0.0f ? 1.0f : 0.0f);
      ep_prob[(int64_t)t * N + i] = ep_prob[(int64_t)t * N + i] + tm;
      pmax = (t == 0) ? tm : nanmaxf(pmax, tm);
    }
    cstr_prob[i] = pmax;
    if (reward) {
      const float omp = 1.0f - pmax;
      const float r = reward[i] * omp;
      reward[i] = (r < 0.0f) ? 0.0f : r;
    }
    if (dones) dones[i] = (reset && reset[i]) ? 1.0f : pmax;
  }
}

void gae_oracle(const float* rew, const float* val, const float* done, const float* tdone,
                const float* next_val, const float* next_done, const float* next_tdone, float gamma,
                float gl, float* adv, float* ret, int T, int64_t N) {
  for (int64_t e = 0; e < N; ++e) {
    float vnext = next_val[e], dn = next_done[e], tdn = next_tdone[e], last = 0.0f;
    for (int t = T - 1; t >= 0; --t) {
      const int64_t o = (int64_t)t * N + e;
      const float nn = 1.0f - dn, tn = 1.0f - tdn;
      float x = gamma * vnext;
      x = x * nn;
      x = x * tn;
      float delta = rew[o] + x;
      delta = delta - val[o];
      float c = gl * nn;
      c = c * tn;
      c = c * last;
      last = delta + c;
      adv[o] = last;
      ret[o] = last + val[o];
      vnext = val[o];
      dn = done[o];
      tdn = tdone[o];
    }
  }
}
