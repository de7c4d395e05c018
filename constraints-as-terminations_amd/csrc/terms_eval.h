// Constraint-term evaluation shared by the stand-alone term kernel (cat_terms.hip) and the fused rollout step
// (rollout.hip): the descriptor table type and eval_term(), one (env, column) value of one term.
// Unfused fp32 (-ffp-contract=off): the abs/limit families are bit-identical to the reference's torch ops
// (cat/constraints.py:23-235).  Round 5: so are the norm based ones (C8 base_orientation, C13 foot_contact_force, the
// contact terms and the command-norm gates): torch.norm(dim=-1) on the pinning platform (torch 2.10 CPU, the build the
// goldens come from) reduces a short last dimension as ONE FMA CHAIN - acc = x0*x0 (rounded); acc = fma(x_i, x_i, acc) -
// measured against five candidate orders on 200 000 random vectors (0 mismatches for the chain, 10-15 % for every
// unfused order, tests/golden/gen_golden.py `gen_terms_scale` re-checks it).  norm3 / the 2-norm below spell that chain
// out with explicit fmaf (rounds 1-4 summed unfused products: <= 4 ulp of the norm away from the reference in 7 % of the
// force elements - enough to flip `c > 0` for a force within an ulp of its limit).
#pragma once

#include "common.h"

namespace terms {

constexpr int kRows = 16;         // envs per tile
constexpr int kMaxBlocks = 256;   // = partial column-maximum rows handed to the CaT step
constexpr int kMaxTerms = 16;

struct TermTable {
  int n;
  int off[kMaxTerms + 1];
  uint32_t wmagic[kMaxTerms];     // catppo_div_magic(width): (env, column) of a flat index without an integer division
  catppo_term_desc d[kMaxTerms];
};

__device__ __forceinline__ float norm3(const float* p) {
  float s = p[0] * p[0];
  s = __fmaf_rn(p[1], p[1], s);
  s = __fmaf_rn(p[2], p[2], s);
  return sqrtf(s);
}

// max over history of |F[e,h,b,:]|
__device__ __forceinline__ float force_peak(const float* forces, int64_t fstride, int64_t env, int H, int B, int b) {
  const float* base = forces + env * fstride + (int64_t)b * 3;
  float m = norm3(base);
  for (int h = 1; h < H; ++h) m = nanmax(m, norm3(base + (int64_t)h * B * 3));
  return m;
}

// `ids`: the term's id list (d.ids, or a copy of it in LDS)
__device__ __forceinline__ float eval_term(const catppo_term_desc& d, const int32_t* ids, int64_t env, int j,
                                           const float* forces, int64_t fstride, int H, int B, const float* command,
                                           int cld) {
  float out = 0.0f;
  switch (d.kind) {
    case CATPPO_TERM_ABS_LIMIT: {
      out = fabsf(d.x[env * d.x_ld + ids[j]]) - d.limit;
    } break;
    case CATPPO_TERM_ABS_DIFF_LIMIT: {
      const float df = d.x[env * d.x_ld + ids[j]] - d.y[env * d.y_ld + ids[j]];
      out = fabsf(df) - d.limit;
    } break;
    case CATPPO_TERM_ABS_DIFF_LIMIT_GATE_CMDY: {
      const float df = d.x[env * d.x_ld + ids[j]] - d.y[env * d.y_ld + ids[j]];
      const float c = fabsf(df) - d.limit;
      const float gate = fabsf(command[env * cld + 1]) < d.aux ? 1.0f : 0.0f;
      out = c * gate;
    } break;
    case CATPPO_TERM_GREATER: {
      out = d.x[env * d.x_ld + ids[0]] > d.limit ? 1.0f : 0.0f;
    } break;
    case CATPPO_TERM_CONTACT_ANY: {
      bool any = false;
      for (int b = 0; b < d.n_ids; ++b) any = any || (force_peak(forces, fstride, env, H, B, ids[b]) > d.limit);
      out = any ? 1.0f : 0.0f;
    } break;
    case CATPPO_TERM_NORM2_LIMIT: {
      const float a = d.x[env * d.x_ld + 0], b = d.x[env * d.x_ld + 1];
      float s = a * a;
      s = __fmaf_rn(b, b, s);           // torch.norm's chain, see the header
      out = sqrtf(s) - d.limit;
    } break;
    case CATPPO_TERM_AIR_TIME: {
      const float gate = norm3(command + env * cld) > d.aux ? 1.0f : 0.0f;
      float c = d.limit - d.x[env * d.x_ld + ids[j]];
      c = c * d.y[env * d.y_ld + ids[j]];
      out = c * gate;
    } break;
    case CATPPO_TERM_N_FOOT_CONTACT: {
      int n = 0;
      for (int b = 0; b < d.n_ids; ++b) n += force_peak(forces, fstride, env, H, B, ids[b]) > 1.0f ? 1 : 0;
      int diff = n - (int)d.limit;
      diff = diff < 0 ? -diff : diff;
      const float gate = norm3(command + env * cld) > d.aux ? 1.0f : 0.0f;
      out = (float)diff * gate;
    } break;
    case CATPPO_TERM_ACTION_RATE: {
      const float df = fabsf(d.x[env * d.x_ld + ids[j]] - d.y[env * d.y_ld + ids[j]]);
      out = df / d.aux - d.limit;
    } break;
    case CATPPO_TERM_FORCE_LIMIT: {
      out = force_peak(forces, fstride, env, H, B, ids[j]) - d.limit;
    } break;
    case CATPPO_TERM_LIMIT_MINUS: {
      out = d.limit - d.x[env * d.x_ld + ids[0]];
    } break;
    case CATPPO_TERM_ABS_LIMIT_GATE_CMDNORM_LT: {
      const float c = fabsf(d.x[env * d.x_ld + ids[j]]) - d.limit;
      const float gate = norm3(command + env * cld) < d.aux ? 1.0f : 0.0f;
      out = c * gate;
    } break;
    default:
      break;
  }
  return out;
}

// validate a host descriptor array and copy it into a by-value kernel table; returns nullptr or a message
inline const char* build_table(const catppo_term_desc* desc, int n_terms, const float* forces,
                               int64_t forces_env_stride, int H, int B, const float* command, int command_ld, int K,
                               TermTable* tab) {
  if (!desc || n_terms < 1 || n_terms > kMaxTerms) return "1 <= n_terms <= 16 descriptors";
  tab->n = n_terms;
  int off = 0;
  for (int t = 0; t < n_terms; ++t) {
    const catppo_term_desc& d = desc[t];
    if (!(d.width >= 1 && d.n_ids >= 0 && d.n_ids <= CATPPO_TERM_MAX_IDS)) return "term width / n_ids out of range";
    const bool needs_forces = d.kind == CATPPO_TERM_CONTACT_ANY || d.kind == CATPPO_TERM_N_FOOT_CONTACT ||
                              d.kind == CATPPO_TERM_FORCE_LIMIT;
    const bool needs_cmd = d.kind == CATPPO_TERM_ABS_DIFF_LIMIT_GATE_CMDY || d.kind == CATPPO_TERM_AIR_TIME ||
                           d.kind == CATPPO_TERM_N_FOOT_CONTACT || d.kind == CATPPO_TERM_ABS_LIMIT_GATE_CMDNORM_LT;
    const bool needs_y = d.kind == CATPPO_TERM_ABS_DIFF_LIMIT || d.kind == CATPPO_TERM_ABS_DIFF_LIMIT_GATE_CMDY ||
                         d.kind == CATPPO_TERM_AIR_TIME || d.kind == CATPPO_TERM_ACTION_RATE;
    if (needs_forces && !(forces != nullptr && H >= 1 && B >= 1 && forces_env_stride >= (int64_t)H * B * 3))
      return "term needs the contact-force history";
    if (needs_cmd && !(command != nullptr && command_ld >= 3)) return "term needs the velocity command";
    if (!needs_forces && d.x == nullptr) return "term needs its primary state tensor";
    if (needs_y && d.y == nullptr) return "term needs its secondary state tensor";
    const bool per_id = d.kind != CATPPO_TERM_GREATER && d.kind != CATPPO_TERM_CONTACT_ANY &&
                        d.kind != CATPPO_TERM_NORM2_LIMIT && d.kind != CATPPO_TERM_N_FOOT_CONTACT &&
                        d.kind != CATPPO_TERM_LIMIT_MINUS;
    if (!(per_id ? d.width == d.n_ids : d.width == 1)) return "term width does not match its id list";
    tab->off[t] = off;
    tab->wmagic[t] = catppo_div_magic((uint32_t)d.width);
    tab->d[t] = d;
    off += d.width;
  }
  tab->off[n_terms] = off;
  if (off != K) return "sum of term widths != K";
  return nullptr;
}

}  // namespace terms
