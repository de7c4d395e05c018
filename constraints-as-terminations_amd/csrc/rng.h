// Counter-based randomness for the rollout noise and the minibatch permutation (gfx950).
//
//   philox4x32_10  Salmon et al., SC'11.  Stateless: (counter, key) -> 4 x 32 random bits, so a kernel needs no
//                  generator state and a replayed hipGraph draws fresh numbers as soon as one counter word (the
//                  iteration number in the device-resident catppo_iter_state) moves.  Restated in numpy in
//                  oracle/rng_oracle.py and pinned there by the Random123 known-answer vectors.
//   box_muller4    four N(0,1) values from one Philox block (replaces Normal.sample(), cleanrl/ppo.py:111)
//   FeistelPerm    keyed bijection of [0,total): 6-round balanced Feistel network on the next power of four,
//                  cycle walking for the overshoot (replaces torch.randperm + the index array, cleanrl/ppo.py:295)
#pragma once

#include <hip/hip_runtime.h>

#include <cstdint>

namespace rng {

struct u32x4 {
  uint32_t x, y, z, w;
};

__host__ __device__ __forceinline__ uint32_t mulhi32(uint32_t a, uint32_t b) {
#if defined(__HIP_DEVICE_COMPILE__)
  return __umulhi(a, b);
#else
  return (uint32_t)(((uint64_t)a * b) >> 32);
#endif
}

__host__ __device__ __forceinline__ u32x4 philox4x32_10(u32x4 c, uint32_t k0, uint32_t k1) {
  constexpr uint32_t M0 = 0xD2511F53u, M1 = 0xCD9E8D57u, W0 = 0x9E3779B9u, W1 = 0xBB67AE85u;
#pragma unroll
  for (int r = 0; r < 10; ++r) {
    const uint32_t hi0 = mulhi32(M0, c.x), lo0 = M0 * c.x;
    const uint32_t hi1 = mulhi32(M1, c.z), lo1 = M1 * c.z;
    c = u32x4{hi1 ^ c.y ^ k0, lo1, hi0 ^ c.w ^ k1, lo0};
    k0 += W0;
    k1 += W1;
  }
  return c;
}

// uint32 -> fp32 in the OPEN interval (0,1): ((x >> 8) + 0.5) * 2^-24, exact
__device__ __forceinline__ float uniform_open(uint32_t x) { return ((float)(x >> 8) + 0.5f) * 5.9604644775390625e-8f; }

// element `which` (0..3) of the four normals of one Philox block: pairs (0,1), (2,3)
__device__ __forceinline__ float box_muller_pick(const u32x4& b, int which) {
  const uint32_t a = which < 2 ? b.x : b.z, c = which < 2 ? b.y : b.w;
  const float r = sqrtf(-2.0f * logf(uniform_open(a)));
  const float th = 6.283185307179586f * uniform_open(c);
  return r * ((which & 1) ? sinf(th) : cosf(th));
}

__host__ __device__ __forceinline__ uint32_t mix32(uint32_t x) {
  x = (x ^ (x >> 16)) * 0x7FEB352Du;
  x = (x ^ (x >> 15)) * 0x846CA68Bu;
  return x ^ (x >> 16);
}

struct FeistelPerm {
  uint32_t key[6];
  uint32_t total;
  int hb;   // half width in bits: 4^hb >= total

  __host__ __device__ void init(uint64_t seed, int64_t iteration, int32_t epoch, int64_t n) {
    const uint32_t k0 = (uint32_t)seed, k1 = (uint32_t)(seed >> 32);
    const u32x4 a = philox4x32_10(u32x4{(uint32_t)epoch, (uint32_t)iteration, 0x50455245u, 0x4D555445u}, k0, k1);
    const u32x4 b = philox4x32_10(u32x4{(uint32_t)epoch, (uint32_t)iteration, 0x50455246u, 0x4D555445u}, k0, k1);
    key[0] = a.x, key[1] = a.y, key[2] = a.z, key[3] = a.w, key[4] = b.x, key[5] = b.y;
    total = (uint32_t)n;
    hb = 1;
    while ((int64_t(1) << (2 * hb)) < n) ++hb;
  }
  __host__ __device__ __forceinline__ uint32_t enc(uint32_t x) const {
    const uint32_t mask = (1u << hb) - 1u;
    uint32_t L = (x >> hb) & mask, R = x & mask;
#pragma unroll
    for (int r = 0; r < 6; ++r) {
      const uint32_t f = mix32(R + key[r]) & mask;
      const uint32_t t = L ^ f;
      L = R;
      R = t;
    }
    return (L << hb) | R;
  }
  __host__ __device__ __forceinline__ int64_t operator()(int64_t j) const {
    uint32_t y = enc((uint32_t)j);
    while (y >= total) y = enc(y);   // cycle walking: the domain is < 4x total, so < 4 rounds on average
    return (int64_t)y;
  }
};

}  // namespace rng
