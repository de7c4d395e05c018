from __future__ import annotations

import torch

from cat_envs import native


def compute_gae(rewards: torch.Tensor, dones: torch.Tensor, values: torch.Tensor, next_values: torch.Tensor,
                discount_factor: float = 0.99, lambda_coefficient: float = 0.95):
    """(returns, normalised advantages) like the nested function at skrl/ppo.py:397-442.

    ``rewards/dones/values``: (memory_size, num_envs, 1) float; ``next_values``: (num_envs, 1) = the
    value of the state after the last stored step (``last_values`` in the reference).  Two launches for
    the recurrence + normalisation instead of ~8 eager ops per memory row.
    """
    shape = rewards.shape
    T, N = shape[0], rewards[0].numel()
    plane = lambda t: t.reshape(T, N).float().contiguous()  # noqa: E731
    nat = native.get(rewards.device)
    adv = torch.empty(T, N, device=rewards.device)
    ret = torch.empty_like(adv)
    nat.gae_skrl(plane(rewards), plane(dones), plane(values), next_values.reshape(N).float().contiguous(),
                 discount_factor, lambda_coefficient, adv, ret)
    nat.adv_normalize(adv, adv)
    return ret.reshape(shape), adv.reshape(shape)
