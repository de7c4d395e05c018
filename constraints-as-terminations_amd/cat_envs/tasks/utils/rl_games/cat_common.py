"""Device versions of the two return-path steps of ``CaTA2CAgent.play_steps``
(reference: rl_games/cat_common.py:59-64 and :96-103)."""
from __future__ import annotations

import torch

from cat_envs import native


def bootstrap_time_outs(shaped_rewards: torch.Tensor, values: torch.Tensor, time_outs: torch.Tensor, gamma: float):
    """``shaped_rewards += gamma * values * time_outs.float()`` in place (``value_bootstrap``), one launch"""
    nat = native.get(shaped_rewards.device)
    nat.value_bootstrap(shaped_rewards, values.reshape(shaped_rewards.shape).contiguous(),
                        time_outs.reshape(-1).to(torch.uint8), gamma)
    return shaped_rewards


def discount_values(fdones, last_extrinsic_values, mb_fdones, mb_extrinsic_values, mb_rewards, gamma: float,
                    tau: float):
    """GAE over a (horizon, num_actors[, 1]) buffer with float dones; returns ``mb_advs`` shaped like
    ``mb_rewards``.  One launch instead of 7 eager ops per step of the horizon."""
    shape = mb_rewards.shape
    T, N = shape[0], mb_rewards[0].numel()
    plane = lambda t: t.reshape(T, N).float().contiguous()  # noqa: E731
    row = lambda t: t.reshape(N).float().contiguous()  # noqa: E731
    nat = native.get(mb_rewards.device)
    adv = torch.empty(T, N, device=mb_rewards.device)
    ret = torch.empty_like(adv)
    nat.gae_rl_games(row(fdones), row(last_extrinsic_values), plane(mb_fdones), plane(mb_extrinsic_values),
                     plane(mb_rewards), gamma, tau, adv, ret)
    return adv.reshape(shape)
