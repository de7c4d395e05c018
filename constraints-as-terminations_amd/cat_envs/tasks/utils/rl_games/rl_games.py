"""``RlGamesVecEnvWrapperCaT``: the env wrapper of the reference's rl_games front end (rl_games/rl_games.py:9-43).

The reference overrides ONE method of isaaclab_rl's ``RlGamesVecEnvWrapper`` - ``step`` - so that the float
termination probability (``terminated``) is what rl_games sees as ``dones`` (the stock wrapper ORs terminated and
truncated into a bool) and the time-outs travel in ``extras["time_outs"]`` for ``value_bootstrap``.  isaaclab_rl is
not available (and not under /root/reference): the surrounding wrapper contract is restated from the reference's
usage only - constructor ``(env, rl_device, clip_obs, clip_actions)``, ``reset()``, ``step()``, ``_process_obs`` -
parity unpinned against isaaclab_rl itself."""
from __future__ import annotations

import torch


class RlGamesVecEnvWrapperCaT:
    def __init__(self, env, rl_device: str = "cuda:0", clip_obs: float = float("inf"), clip_actions: float = float("inf")):
        self.env = env
        self._rl_device = torch.device(rl_device)
        self._clip_obs, self._clip_actions = float(clip_obs), float(clip_actions)
        self._sim_device = getattr(env.unwrapped, "device", self._rl_device)
        self.rlg_num_states = 0

    @property
    def unwrapped(self):
        return self.env.unwrapped

    @property
    def num_envs(self) -> int:
        return self.unwrapped.num_envs

    def get_number_of_agents(self) -> int:
        return 1

    def get_env_info(self) -> dict:
        u = self.unwrapped
        return {"observation_space": u.single_observation_space["policy"], "action_space": u.single_action_space}

    def _process_obs(self, obs_dict):
        obs = obs_dict["policy"]
        if self._clip_obs != float("inf"):
            obs = torch.clamp(obs, -self._clip_obs, self._clip_obs)
        return obs.to(self._rl_device)

    def reset(self):
        obs_dict, _ = self.env.reset()
        return self._process_obs(obs_dict)

    def step(self, actions):  # noqa: D102   reference rl_games/rl_games.py:10-43
        actions = actions.detach().clone().to(device=self._sim_device)
        if self._clip_actions != float("inf"):
            actions = torch.clamp(actions, -self._clip_actions, self._clip_actions)
        obs_dict, rew, terminated, truncated, extras = self.env.step(actions)
        # time-out information for value_bootstrap (infinite-horizon tasks only)
        if not getattr(self.unwrapped.cfg, "is_finite_horizon", False):
            extras["time_outs"] = truncated.to(device=self._rl_device)
        obs_and_states = self._process_obs(obs_dict)
        rew = rew.to(device=self._rl_device)
        dones = terminated                    # CaT: the FLOAT termination probability, not terminated | truncated
        extras = {k: (v.to(device=self._rl_device, non_blocking=True) if hasattr(v, "to") else v)
                  for k, v in extras.items()}
        if "log" in extras:                   # remap extras from "log" to "episode"
            extras["episode"] = extras.pop("log")
        return obs_and_states, rew, dones, extras
