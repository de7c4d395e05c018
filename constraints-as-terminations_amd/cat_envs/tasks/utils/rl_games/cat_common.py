"""``CaTA2CAgent.play_steps`` on the HIP kernels (reference: rl_games/cat_common.py).

The reference subclasses rl_games' ``A2CAgent`` to make three things float: the ``dones`` plane of the experience
buffer, the agent's ``dones`` vector, and the episode bookkeeping inside ``play_steps`` (``dones.ge(1.0)`` ends an
episode, ``current_rewards *= 1 - dones``).  rl_games itself (models, schedulers, the PPO update of ``A2CAgent``) is a
third-party package and out of scope; what is built here is the rollout half the reference rewrites:

* ``CaTA2CAgent`` - a self-contained driver with the members ``play_steps`` touches (``obs``, ``dones``,
  ``experience_buffer``, ``update_list`` / ``tensor_list``, ``current_rewards`` / ``_shaped_rewards`` / ``_lengths``,
  the three game meters, ``value_bootstrap``, ``gamma`` / ``tau``, ``rewards_shaper``) whose ``play_steps()`` returns
  the reference's ``batch_dict``.  Policy outputs come from the build's ``Agent`` (HIP MLP) by default;
  ``get_action_values`` / ``get_values`` / ``env_step`` are the reference's hook names and can be overridden.
* per env step the bookkeeping is ONE launch (``catppo_rlg_episode_step``) instead of ~12 eager ops and a
  ``nonzero()`` host sync; the meters (rl_games ``AverageMeter``) live in device memory.
* ``bootstrap_time_outs`` (:59-64) and ``discount_values`` (:96-103) are the device functions of round 2.
"""
from __future__ import annotations

import time

import torch

from cat_envs import native

from .cat_experience import CaTExperienceBuffer, swap_and_flatten01


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


class DeviceAverageMeter:
    """read-only view of one of the three device-resident game meters (rl_games ``AverageMeter`` surface:
    ``current_size`` and ``get_mean()``); reading synchronises, ``play_steps`` itself never reads"""

    def __init__(self, owner: "CaTA2CAgent", which: str):
        self._o, self._w = owner, which

    def _read(self):
        return self._o.nat.rlg_meters_read(self._o._meters)

    @property
    def current_size(self) -> int:
        return int(getattr(self._read(), {"rewards": "size_rewards", "shaped": "size_shaped", "lengths": "size_lengths"}[self._w]))

    def get_mean(self):
        m = self._read()
        if self._w == "lengths":
            return float(m.mean_lengths)
        arr = m.mean_rewards if self._w == "rewards" else m.mean_shaped_rewards
        return [float(arr[v]) for v in range(self._o.value_size)]


class CaTA2CAgent:
    """Rollout half of the reference's ``CaTA2CAgent`` (rl_games/cat_common.py:8-112).

    ``vec_env``  object with ``step(actions) -> (obs, rewards, dones, infos)`` (``RlGamesVecEnvWrapperCaT``) and
                 ``reset() -> obs``; ``obs`` is a tensor or ``{"obs": tensor}``
    ``agent``    the build's ``cleanrl.ppo.Agent`` (or anything with ``get_action_and_value`` / ``get_value``);
                 only needed when the ``get_action_values`` / ``get_values`` hooks are not overridden
    ``config``   dict: horizon_length, gamma, tau, value_bootstrap, reward_scale (DefaultRewardsShaper.scale_value),
                 games_to_track, value_size
    """

    def __init__(self, vec_env, num_actors: int, obs_shape, actions_num: int, config: dict, agent=None, device="cuda",
                 algo_observer=None):
        c = dict(horizon_length=24, gamma=0.99, tau=0.95, value_bootstrap=True, reward_scale=1.0, games_to_track=100,
                 value_size=1)
        c.update(config or {})
        self.config, self.vec_env, self.agent = c, vec_env, agent
        self.ppo_device = torch.device(device)
        if self.ppo_device.type != "cuda" or not torch.cuda.is_available():
            raise RuntimeError("CaTA2CAgent needs a HIP device (MI355X); there is no CPU fallback")
        self.nat = native.get(self.ppo_device)
        self.num_actors, self.num_agents = int(num_actors), 1         # [::num_agents] striding of done indices: 1 agent
        self.horizon_length, self.gamma, self.tau = int(c["horizon_length"]), float(c["gamma"]), float(c["tau"])
        self.value_bootstrap, self.value_size = bool(c["value_bootstrap"]), int(c["value_size"])
        self.reward_scale = float(c["reward_scale"])
        self.has_central_value = self.use_action_masks = False
        self.algo_observer = algo_observer
        self.batch_size = self.horizon_length * self.num_actors
        self.env_info = {"observation_space": tuple(obs_shape), "action_space": (int(actions_num),), "agents": 1,
                         "value_size": self.value_size}
        self.update_list = ["actions", "neglogpacs", "values", "mus", "sigmas"]
        self.tensor_list = self.update_list + ["obses", "states", "dones"]
        self.obs = None
        self.init_tensors()

    # ------------------------------------------------------------------ reference: init_tensors (:14-33)
    def init_tensors(self):
        batch = self.num_agents * self.num_actors
        algo_info = {"num_actors": self.num_actors, "horizon_length": self.horizon_length,
                     "has_central_value": self.has_central_value, "use_action_masks": self.use_action_masks}
        # experience buffer with float32 dones, and the agent's own dones vector as float32 ones (:24-33)
        self.experience_buffer = CaTExperienceBuffer(self.env_info, algo_info, self.ppo_device)
        dev = self.ppo_device
        self.dones = torch.ones((batch,), dtype=torch.float32, device=dev)
        self.current_rewards = torch.zeros(batch, self.value_size, device=dev)
        self.current_shaped_rewards = torch.zeros(batch, self.value_size, device=dev)
        self.current_lengths = torch.zeros(batch, device=dev)                         # rl_games keeps them in fp32
        self._meters = self.nat.rlg_meters_new(int(self.config["games_to_track"]))
        self.game_rewards = DeviceAverageMeter(self, "rewards")
        self.game_shaped_rewards = DeviceAverageMeter(self, "shaped")
        self.game_lengths = DeviceAverageMeter(self, "lengths")
        self._done_mask = torch.zeros(batch, dtype=torch.uint8, device=dev)
        self._shaped = torch.zeros(batch, self.value_size, device=dev)

    # ------------------------------------------------------------------ hooks (rl_games A2CBase names)
    def rewards_shaper(self, rewards: torch.Tensor) -> torch.Tensor:
        """rl_games DefaultRewardsShaper with ``scale_value`` (shift 0, no clipping): a new tensor, like the original"""
        torch.mul(rewards, self.reward_scale, out=self._shaped)
        return self._shaped

    def cast_obs(self, x):
        return x

    def env_reset(self):
        obs = self.vec_env.reset()
        self.obs = obs if isinstance(obs, dict) else {"obs": obs}
        return self.obs

    def env_step(self, actions):
        obs, rewards, dones, infos = self.vec_env.step(actions)
        if rewards.ndim == 1:                       # A2CBase.env_step: value_size == 1 -> rewards.unsqueeze(1)
            rewards = rewards.unsqueeze(1)
        return (obs if isinstance(obs, dict) else {"obs": obs}), rewards, dones, infos

    def get_action_values(self, obs):
        """{"actions", "values" (N,1), "neglogpacs" (N,), "mus", "sigmas"} from the HIP policy"""
        a = self.agent
        x = obs["obs"]
        # ONE forward per env step: the noise is drawn here, so the mean follows from the sample the head kernel made,
        # mu = a - sigma * eps (within one rounding of |a| of the kernel's own mu; rl_games only feeds "mus" to its KL
        # estimate).  A second, deterministic forward just to read mu doubled the rollout's forward cost.
        n = x.reshape(-1, a.obs_dim).shape[0]
        eps = torch.randn(n, a.act_dim, device=x.device)
        action, logp, _, value = a.get_action_and_value(x, eps=eps)
        sig = torch.exp(a.actor_logstd.detach()).expand_as(action)
        mu = action - sig * eps
        return {"actions": action, "values": value, "neglogpacs": -logp, "mus": mu, "sigmas": sig}

    def get_values(self, obs):
        return self.agent.get_value(obs["obs"])

    def discount_values(self, fdones, last_extrinsic_values, mb_fdones, mb_extrinsic_values, mb_rewards):
        return discount_values(fdones, last_extrinsic_values, mb_fdones, mb_extrinsic_values, mb_rewards, self.gamma,
                               self.tau)

    # ------------------------------------------------------------------ reference: play_steps (:35-112)
    def play_steps(self):
        update_list, buf, nat = self.update_list, self.experience_buffer, self.nat
        step_time = 0.0
        if self.obs is None:
            self.env_reset()
        for n in range(self.horizon_length):
            res_dict = self.get_action_values(self.obs)
            buf.update_data("obses", n, self.obs["obs"])
            buf.update_data("dones", n, self.dones)                 # the dones of BEFORE this env step (:47)
            for k in update_list:
                buf.update_data(k, n, res_dict[k])
            t0 = time.time()
            self.obs, rewards, dones, infos = self.env_step(res_dict["actions"])
            step_time += time.time() - t0
            shaped_rewards = self.rewards_shaper(rewards)
            if self.value_bootstrap and "time_outs" in infos:      # :59-64, one launch
                bootstrap_time_outs(shaped_rewards, res_dict["values"], self.cast_obs(infos["time_outs"]), self.gamma)
            buf.update_data("rewards", n, shaped_rewards)
            self.dones = dones if dones.dtype == torch.float32 else dones.float()         # :68 (CaT)
            # :69-88 in one launch: running returns / lengths, dones >= 1.0 as the episode end, the three meters,
            # current_* *= (1 - dones), finished lengths zeroed
            rew = rewards if (rewards.dtype == torch.float32 and rewards.is_contiguous()) else rewards.float().contiguous()
            nat.rlg_episode_step(rew, shaped_rewards, self.dones.contiguous(), self.current_rewards,
                                 self.current_shaped_rewards, self.current_lengths, self._meters, self._done_mask)
            if self.algo_observer is not None:
                # an observer wants INDICES (rl_games process_infos): this nonzero() is the reference's host sync,
                # paid only when somebody listens
                idx = self._done_mask.bool().nonzero(as_tuple=False)[::self.num_agents]
                self.algo_observer.process_infos(infos, idx)
        last_values = self.get_values(self.obs)
        mb_fdones, mb_values = buf.tensor_dict["dones"], buf.tensor_dict["values"]
        mb_rewards = buf.tensor_dict["rewards"]
        mb_advs = self.discount_values(self.dones, last_values, mb_fdones, mb_values, mb_rewards)
        mb_returns = mb_advs + mb_values
        batch_dict = buf.get_transformed_list(swap_and_flatten01, self.tensor_list)
        batch_dict["returns"] = swap_and_flatten01(mb_returns)
        batch_dict["played_frames"] = self.batch_size
        batch_dict["step_time"] = step_time
        return batch_dict
