"""Float-dones experience buffer of the reference's rl_games front end.

Reference: rl_games/cat_experience.py:20-33 subclasses rl_games' ``ExperienceBuffer`` and re-creates its ``dones``
plane as fp32 (rl_games stores uint8; a CaT termination is a probability in [0,1]).  rl_games (1.6.1 per the
reference's setup.py) is not vendored, so the base class' contract is restated here from its published API - the part
``CaTA2CAgent.play_steps`` uses (rl_games/cat_common.py:17-112): construction from ``(env_info, algo_info, device)``,
``tensor_dict``, ``update_data(name, index, val)``, ``get_transformed(op)``, ``get_transformed_list(op, names)``.
PARITY UNPINNED against rl_games itself (no source, no test in the reference tree).

Planes are time-major ``(horizon_length, num_actors * num_agents, ...)`` - the layout ``discount_values`` /
``catppo_gae_planes(CATPPO_GAE_RL_GAMES)`` scans with coalesced rows - and live on the HIP device.
"""
from __future__ import annotations

from typing import Dict, Iterable, Sequence

import numpy as np
import torch


def _shape_of(space) -> tuple:
    if space is None:
        return ()
    if isinstance(space, (tuple, list)):
        return tuple(int(s) for s in space)
    return tuple(int(s) for s in getattr(space, "shape", ()))


class CaTExperienceBuffer:
    """``CaTExperienceBuffer(env_info, algo_info, device)`` with

    env_info   {"observation_space": space | shape | {name: space}, "action_space": space | shape,
                "agents": int (default 1), "value_size": int (default 1), "state_space": optional}
    algo_info  {"num_actors": int, "horizon_length": int, "has_central_value": bool, "use_action_masks": bool}
    """

    def __init__(self, env_info: dict, algo_info: dict, device="cuda"):
        self.env_info, self.algo_info = env_info, algo_info
        self.device = torch.device(device)
        self.num_agents = int(env_info.get("agents", 1))
        self.value_size = int(env_info.get("value_size", 1))
        self.num_actors = int(algo_info["num_actors"])
        self.horizon_length = int(algo_info["horizon_length"])
        self.has_central_value = bool(algo_info.get("has_central_value", False))
        self.use_action_masks = bool(algo_info.get("use_action_masks", False))
        batch = self.num_actors * self.num_agents
        self.obs_base_shape = (self.horizon_length, batch)
        self.state_base_shape = (self.horizon_length, self.num_actors)
        self.tensor_dict: Dict[str, object] = {}
        self._init_from_env_info(env_info)

    @classmethod
    def from_shapes(cls, num_actors: int, horizon_length: int, obs_shape: Sequence[int], actions_num: int,
                    device="cuda"):
        return cls({"observation_space": tuple(obs_shape), "action_space": (int(actions_num),)},
                   {"num_actors": num_actors, "horizon_length": horizon_length}, device)

    # -- construction ----------------------------------------------------------------------------
    def _plane(self, shape=(), base=None, dtype=torch.float32) -> torch.Tensor:
        return torch.zeros(*(base or self.obs_base_shape), *shape, dtype=dtype, device=self.device)

    def _init_from_env_info(self, env_info: dict):
        obs = env_info["observation_space"]
        if isinstance(obs, dict):
            self.tensor_dict["obses"] = {k: self._plane(_shape_of(v)) for k, v in obs.items()}
        else:
            self.tensor_dict["obses"] = self._plane(_shape_of(obs))
        if self.has_central_value:
            self.tensor_dict["states"] = self._plane(_shape_of(env_info.get("state_space")), self.state_base_shape)
        val = (self.value_size,)
        self.tensor_dict["rewards"] = self._plane(val)
        self.tensor_dict["values"] = self._plane(val)
        self.tensor_dict["neglogpacs"] = self._plane()
        # the CaT override (cat_experience.py:27-33): float32 termination probability, not uint8
        self.tensor_dict["dones"] = self._plane()
        act = _shape_of(env_info["action_space"])          # continuous (Box) actions, like the Solo12 task
        self.tensor_dict["actions"] = self._plane(act)
        self.tensor_dict["mus"] = self._plane(act)
        self.tensor_dict["sigmas"] = self._plane(act)
        if self.use_action_masks:
            self.tensor_dict["action_masks"] = self._plane(act, dtype=torch.bool)

    # -- the contract play_steps relies on ---------------------------------------------------------
    def update_data(self, name: str, index: int, val):
        dst = self.tensor_dict[name]
        if isinstance(val, dict):
            for k, v in val.items():
                dst[k][index].copy_(v.reshape(dst[k][index].shape))
        else:
            dst[index].copy_(val.reshape(dst[index].shape))      # bool / uint8 dones widen to fp32 in the copy

    def update_data_rnn(self, name: str, indices, play_mask, val):
        dst = self.tensor_dict[name]
        if isinstance(val, dict):
            for k, v in val.items():
                dst[k][indices, play_mask] = v
        else:
            dst[indices, play_mask] = val

    def get_transformed(self, transform_op) -> dict:
        out = {}
        for k, v in self.tensor_dict.items():
            out[k] = {kd: transform_op(vd) for kd, vd in v.items()} if isinstance(v, dict) else transform_op(v)
        return out

    def get_transformed_list(self, transform_op, tensor_list: Iterable[str]) -> dict:
        out = {}
        for k in tensor_list:
            v = self.tensor_dict.get(k)
            if v is None:
                continue
            out[k] = {kd: transform_op(vd) for kd, vd in v.items()} if isinstance(v, dict) else transform_op(v)
        return out


class CaTVectorizedReplayBuffer:
    """Ring replay buffer with a float32 ``dones`` column (reference rl_games/cat_experience.py:7-17 re-creates the
    uint8 ``dones`` of rl_games' ``VectorizedReplayBuffer`` as fp32).  Contract restated from rl_games' published class
    (not vendored: parity unpinned): ``add(obs, action, reward, next_obs, done)`` appends a batch of transitions at the
    ring position (wrapping), ``sample(batch_size)`` draws uniformly from the filled part and returns
    ``(obses, actions, rewards, next_obses, dones)``."""

    def __init__(self, obs_shape, action_shape, capacity: int, device="cuda"):
        self.device = torch.device(device)
        e = lambda *s: torch.empty((capacity, *s), dtype=torch.float32, device=self.device)   # noqa: E731
        self.obses, self.next_obses = e(*obs_shape), e(*obs_shape)
        self.actions, self.rewards = e(*action_shape), e(1)
        self.dones = e(1)                      # the CaT override: termination PROBABILITY, float32
        self.capacity, self.idx, self.full = int(capacity), 0, False

    def add(self, obs, action, reward, next_obs, done):
        n = obs.shape[0]
        remaining = min(self.capacity - self.idx, n)
        overflow = n - remaining
        if remaining < n:                      # wrap: the tail of the batch goes to the front
            for dst, src in ((self.obses, obs), (self.actions, action), (self.rewards, reward),
                             (self.next_obses, next_obs), (self.dones, done)):
                dst[0:overflow] = src[-overflow:].reshape(overflow, *dst.shape[1:])
            self.full = True
        for dst, src in ((self.obses, obs), (self.actions, action), (self.rewards, reward),
                         (self.next_obses, next_obs), (self.dones, done)):
            dst[self.idx:self.idx + remaining] = src[:remaining].reshape(remaining, *dst.shape[1:])
        self.idx = (self.idx + n) % self.capacity
        self.full = self.full or self.idx == 0

    def sample(self, batch_size: int, generator=None):
        hi = self.capacity if self.full else self.idx
        idxs = torch.randint(0, hi, (batch_size,), device=self.device, generator=generator)
        return self.obses[idxs], self.actions[idxs], self.rewards[idxs], self.next_obses[idxs], self.dones[idxs]


def swap_and_flatten01(arr: torch.Tensor) -> torch.Tensor:
    """(horizon, actors, ...) -> (actors * horizon, ...) like rl_games.common.a2c_common.swap_and_flatten01"""
    if arr is None:
        return arr
    s = arr.size()
    return arr.transpose(0, 1).reshape(s[0] * s[1], *s[2:])
