"""ORACLE (test infrastructure, NOT product code) - numpy restatement of ``CaTA2CAgent.play_steps`` with float dones.

Follows the reference line by line (rl_games/cat_common.py:35-112): the buffer slot ``n`` receives the observation and
the dones of BEFORE the env step (:46-47), rewards are shaped and - with ``value_bootstrap`` - increased by
``gamma * values * time_outs`` (:57-64), an episode ends where ``dones >= 1.0`` (:71-73), the meters are updated with
the running returns / lengths of the finished episodes (:75-79), the running returns are multiplied by the FLOAT
``1 - dones`` and the finished lengths zeroed (:82-88); after the horizon ``discount_values`` (rl_games, published
recurrence = ``ppo_oracle.gae_rl_games``) gives the advantages, ``returns = advs + values`` (:90-101), and the listed
planes are handed over as ``swap_and_flatten01`` views (:103-110).  Pinned by tests/golden/rlg_play_steps.npz, which
gen_golden.py produces by executing the reference's own method.  ``AverageMeter`` restates rl_games' torch_ext class
(third party, not under /root/reference: parity unpinned against rl_games itself).
"""
from __future__ import annotations

import numpy as np

F32 = np.float32


class AverageMeter:
    def __init__(self, shape, max_size):
        self.max_size, self.current_size, self.mean = int(max_size), 0, np.zeros(shape, F32)

    def update(self, values: np.ndarray):
        size = values.shape[0]
        if size == 0:
            return
        new_mean = values.astype(F32).mean(axis=0, dtype=F32)
        size = int(np.clip(size, 0, self.max_size))
        old_size = min(self.max_size - size, self.current_size)
        size_sum = old_size + size
        self.current_size = size_sum
        self.mean = ((self.mean * F32(old_size) + new_mean * F32(size)) / F32(size_sum)).astype(F32)


def swap_and_flatten01(a: np.ndarray) -> np.ndarray:
    s = a.shape
    return np.ascontiguousarray(a.swapaxes(0, 1)).reshape(s[0] * s[1], *s[2:])


def discount_values(fdones, last_values, mb_fdones, mb_values, mb_rewards, gamma, tau):
    """rl_games A2CBase.discount_values with float dones; every fp32 rounding explicit (torch op order)"""
    T = mb_rewards.shape[0]
    g, gl = F32(gamma), F32(gamma * tau)
    adv = np.zeros_like(mb_rewards, dtype=F32)
    last = np.zeros_like(mb_rewards[0], dtype=F32)
    for t in reversed(range(T)):
        if t == T - 1:
            nn_, nv = (F32(1) - fdones).astype(F32), last_values
        else:
            nn_, nv = (F32(1) - mb_fdones[t + 1]).astype(F32), mb_values[t + 1]
        nn_ = nn_[:, None]
        delta = ((mb_rewards[t] + ((g * nv).astype(F32) * nn_).astype(F32)).astype(F32) - mb_values[t]).astype(F32)
        last = (delta + ((gl * nn_).astype(F32) * last).astype(F32)).astype(F32)
        adv[t] = last
    return adv


class PlayStepsOracle:
    def __init__(self, N, T, D, A, obs0, gamma=0.99, tau=0.95, reward_scale=1.0, value_bootstrap=True,
                 games_to_track=100):
        self.N, self.T, self.gamma, self.tau = N, T, gamma, tau
        self.reward_scale, self.value_bootstrap = F32(reward_scale), value_bootstrap
        self.obs = np.asarray(obs0, F32)
        self.dones = np.ones(N, F32)
        self.current_rewards, self.current_shaped_rewards = np.zeros((N, 1), F32), np.zeros((N, 1), F32)
        self.current_lengths = np.zeros(N, F32)
        self.game_rewards, self.game_shaped_rewards = AverageMeter((1,), games_to_track), AverageMeter((1,), games_to_track)
        self.game_lengths = AverageMeter((), games_to_track)
        z = lambda *s: np.zeros((T, N, *s), F32)
        self.buf = {"obses": z(D), "rewards": z(1), "values": z(1), "neglogpacs": z(), "dones": z(), "actions": z(A),
                    "mus": z(A), "sigmas": z(A)}
        self.done_masks = []

    def play_steps(self, res_of_step, env_of_step, last_values, tensor_list):
        """res_of_step(n) -> dict of policy outputs; env_of_step(n, actions) -> (next_obs, rewards (N,1), dones, time_outs)"""
        for n in range(self.T):
            res = res_of_step(n)
            self.buf["obses"][n], self.buf["dones"][n] = self.obs, self.dones
            for k in ("actions", "neglogpacs", "values", "mus", "sigmas"):
                self.buf[k][n] = res[k]
            self.obs, rewards, dones, time_outs = env_of_step(n, res["actions"])
            shaped = (rewards * self.reward_scale).astype(F32)
            if self.value_bootstrap and time_outs is not None:
                shaped = (shaped + ((F32(self.gamma) * res["values"]).astype(F32)
                                    * time_outs.astype(F32)[:, None]).astype(F32)).astype(F32)
            self.buf["rewards"][n] = shaped
            self.dones = dones.astype(F32)
            self.current_rewards = (self.current_rewards + rewards).astype(F32)
            self.current_shaped_rewards = (self.current_shaped_rewards + shaped).astype(F32)
            self.current_lengths = (self.current_lengths + F32(1)).astype(F32)
            done = self.dones >= F32(1.0)
            self.done_masks.append(done.copy())
            self.game_rewards.update(self.current_rewards[done])
            self.game_shaped_rewards.update(self.current_shaped_rewards[done])
            self.game_lengths.update(self.current_lengths[done])
            nd = (F32(1.0) - self.dones).astype(F32)[:, None]
            self.current_rewards = (self.current_rewards * nd).astype(F32)
            self.current_shaped_rewards = (self.current_shaped_rewards * nd).astype(F32)
            self.current_lengths[done] = 0
        advs = discount_values(self.dones, last_values, self.buf["dones"], self.buf["values"], self.buf["rewards"],
                               self.gamma, self.tau)
        returns = (advs + self.buf["values"]).astype(F32)
        batch = {k: swap_and_flatten01(self.buf[k]) for k in tensor_list if k in self.buf}
        batch["returns"] = swap_and_flatten01(returns)
        batch["played_frames"] = self.N * self.T
        return batch
