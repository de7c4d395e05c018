"""ORACLE (test infrastructure, NOT product code) - numpy restatement of the on-device randomness.

The reference draws its action noise with ``Normal.sample()`` (cleanrl/ppo.py:111) and its minibatch order with
``torch.randperm`` (:295) from torch's generator; neither stream is reproducible across devices, so parity tests
inject noise / permutations.  The build's own on-device generators (csrc/rng.h) are a counter-based
Philox4x32-10 (Salmon et al., "Parallel random numbers: as easy as 1, 2, 3", SC'11 - the published algorithm,
pinned below by its known-answer vectors) + Box-Muller, and a keyed Feistel bijection with cycle walking for
the permutation.  This file restates both so that the kernels can be checked element for element.
Only tests/ may import it.
"""
from __future__ import annotations

import numpy as np

M0, M1 = np.uint64(0xD2511F53), np.uint64(0xCD9E8D57)
W0, W1 = np.uint32(0x9E3779B9), np.uint32(0xBB67AE85)
U32 = np.uint64(0xFFFFFFFF)


def philox4x32_10(ctr, key):
    """ctr: (...,4) uint32, key: (...,2) uint32 -> (...,4) uint32"""
    c = [np.asarray(ctr[..., i], np.uint64) for i in range(4)]
    k0 = np.asarray(key[..., 0], np.uint32).copy()
    k1 = np.asarray(key[..., 1], np.uint32).copy()
    for _ in range(10):
        p0 = M0 * c[0]
        p1 = M1 * c[2]
        hi0, lo0 = p0 >> np.uint64(32), p0 & U32
        hi1, lo1 = p1 >> np.uint64(32), p1 & U32
        c = [(hi1 ^ c[1] ^ k0.astype(np.uint64)) & U32, lo1, (hi0 ^ c[3] ^ k1.astype(np.uint64)) & U32, lo0]
        with np.errstate(over="ignore"):
            k0 = (k0 + W0).astype(np.uint32)
            k1 = (k1 + W1).astype(np.uint32)
    return np.stack([x.astype(np.uint32) for x in c], axis=-1)


# Random123 known-answer vectors (kat_vectors, "philox4x32 10")
KAT = [
    ((0, 0, 0, 0), (0, 0), (0x6627e8d5, 0xe169c58d, 0xbc57ac4c, 0x9b00dbd8)),
    ((0xffffffff,) * 4, (0xffffffff,) * 2, (0x408f276d, 0x41c83b0e, 0xa20bc7c6, 0x6d5451fd)),
    ((0x243f6a88, 0x85a308d3, 0x13198a2e, 0x03707344), (0xa4093822, 0x299f31d0),
     (0xd16cfe09, 0x94fdcceb, 0x5001e420, 0x24126ea1)),
]


def uniform_open(x):
    """uint32 -> fp32 in (0,1): ((x >> 8) + 0.5) * 2^-24  (every value exactly representable)"""
    return ((np.asarray(x, np.uint32) >> np.uint32(8)).astype(np.float32) + np.float32(0.5)) * np.float32(2.0 ** -24)


def box_muller(x4):
    """(...,4) uint32 -> (...,4) fp32 N(0,1): pairs (0,1) and (2,3): r = sqrt(-2 ln u_a), theta = 2 pi u_b,
    outputs r cos(theta), r sin(theta)."""
    u = uniform_open(x4)
    out = np.empty(u.shape, np.float32)
    for a in (0, 2):
        r = np.sqrt(np.float32(-2.0) * np.log(u[..., a]).astype(np.float32)).astype(np.float32)
        th = (np.float32(6.283185307179586) * u[..., a + 1]).astype(np.float32)
        out[..., a] = r * np.cos(th).astype(np.float32)
        out[..., a + 1] = r * np.sin(th).astype(np.float32)
    return out


def action_noise(seed: int, iteration: int, step: int, n_envs: int, act_dim: int) -> np.ndarray:
    """eps[i,k] of catppo_policy_act_rng: counter {i, k//4, step, iteration}, key = (seed lo, seed hi), lane k%4"""
    nq = (act_dim + 3) // 4
    i = np.arange(n_envs, dtype=np.uint32)[:, None].repeat(nq, 1)
    q = np.arange(nq, dtype=np.uint32)[None, :].repeat(n_envs, 0)
    ctr = np.stack([i, q, np.full_like(i, step), np.full_like(i, iteration & 0xFFFFFFFF)], -1)
    key = np.broadcast_to(np.array([seed & 0xFFFFFFFF, (seed >> 32) & 0xFFFFFFFF], np.uint32), ctr.shape[:-1] + (2,))
    z = box_muller(philox4x32_10(ctr, key))
    return z.reshape(n_envs, nq * 4)[:, :act_dim].copy()


# ------------------------------------------------------------------ keyed bijection of [0, total)
def _mix(x):
    x = np.asarray(x, np.uint32)
    with np.errstate(over="ignore"):
        x = (x ^ (x >> np.uint32(16))) * np.uint32(0x7FEB352D)
        x = (x ^ (x >> np.uint32(15))) * np.uint32(0x846CA68B)
    return x ^ (x >> np.uint32(16))


def feistel_round_keys(seed: int, iteration: int, epoch: int) -> np.ndarray:
    ctr = np.array([[epoch & 0xFFFFFFFF, iteration & 0xFFFFFFFF, 0x50455245, 0x4D555445],     # "PERE","MUTE"
                    [epoch & 0xFFFFFFFF, iteration & 0xFFFFFFFF, 0x50455246, 0x4D555445]], np.uint32)
    key = np.array([[seed & 0xFFFFFFFF, (seed >> 32) & 0xFFFFFFFF]] * 2, np.uint32)
    return philox4x32_10(ctr, key).reshape(-1)          # 8 round keys


def permutation(seed: int, iteration: int, epoch: int, total: int) -> np.ndarray:
    """P(j), j = 0..total-1: 6-round balanced Feistel network on 2*hb bits (4^hb >= total), cycle walking"""
    hb = 1
    while (1 << (2 * hb)) < total:
        hb += 1
    mask = np.uint32((1 << hb) - 1)
    keys = feistel_round_keys(seed, iteration, epoch)

    def enc(x):
        L, R = (x >> np.uint32(hb)) & mask, x & mask
        for r in range(6):
            with np.errstate(over="ignore"):
                f = _mix(R + keys[r]) & mask
            L, R = R, L ^ f
        return (L << np.uint32(hb)) | R

    y = enc(np.arange(total, dtype=np.uint32))
    bad = y >= total
    while bad.any():
        y[bad] = enc(y[bad])
        bad = y >= total
    return y.astype(np.int64)
