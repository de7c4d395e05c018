"""CleanRL-style PPO for CaT (float dones) on MI355X.

Same entry points as the reference (cleanrl/ppo.py): ``RunningMeanStd``, ``layer_init``,
``Agent(envs)`` with ``get_value / get_action_and_value / forward`` and the 23-key
``state_dict``, and ``PPO(envs, ppo_cfg, run_path)``.  The algorithm is the reference's, step
for step (rollout bookkeeping :201-230, GAE with float dones and a separate time-out mask
:251-277, value normaliser updated twice :287-288, epochs x minibatches with clipped policy /
value losses, global-norm clip, Adam, linear lr anneal :294-354); every tensor operation of it
runs in libcatppo.so (hand-written HIP, see include/catppo.h):

    rollout step   catppo_rms_update + catppo_rms_normalize   (writes obs[step+1] in place)
                   catppo_policy_act                          (writes actions/logprobs/values[step])
                   env.step -> catppo_cat_terms + catppo_cat_step
    after rollout  catppo_value, catppo_gae, 2x (catppo_rms_update + catppo_rms_normalize)
    minibatch      catppo_ppo_minibatch_grad  [RCCL all-reduce of the flat gradient]  catppo_clip_adam

There is no host synchronisation inside an iteration (the reference syncs once per
constraint term per env step and once per minibatch); diagnostics accumulate on the device
and are read once per iteration.  Parameters, gradients and Adam moments live in ONE flat
fp32 buffer each, so the gradient exchange of an env-sharded run is a single all-reduce.
"""
from __future__ import annotations

import math
import os
import time

import numpy as np
import torch
import torch.nn as nn

from cat_envs import native, parallel

DEFAULT_HIDDEN = (512, 256, 128)     # reference Agent (ppo.py:78-95)


# ------------------------------------------------------------------------------------------------
class RunningMeanStd(nn.Module):
    """Running mean / variance with Chan's merge (reference ppo.py:12-62); count starts at 1."""

    def __init__(self, shape=(), epsilon=1e-08, device=None):
        super().__init__()
        self.register_buffer("running_mean", torch.zeros(shape, device=device))
        self.register_buffer("running_var", torch.ones(shape, device=device))
        self.register_buffer("count", torch.ones((), device=device))
        self.epsilon = epsilon
        #: torch.distributed group over which batch moments are summed (env-sharded runs)
        self.dist_group = None

    @property
    def dim(self) -> int:
        return max(1, self.running_mean.numel())

    def forward(self, obs: torch.Tensor, update: bool = True) -> torch.Tensor:
        x = obs if obs.dtype == torch.float32 else obs.float()
        rows = x.reshape(-1, self.dim)
        if rows.stride(-1) != 1:
            rows = rows.contiguous()
        out = torch.empty(rows.shape, device=rows.device)
        self.normalize_into(rows, out, update=update)
        return out.reshape(obs.shape)

    def update(self, x: torch.Tensor):
        rows = x.float().reshape(-1, self.dim)
        self._update_rows(rows if rows.stride(-1) == 1 else rows.contiguous())

    def _update_rows(self, rows: torch.Tensor):
        nat = native.get(rows.device)
        n, d = rows.shape
        group = self.dist_group
        if group is not None and parallel.active(group):
            if not hasattr(self, "_sums"):
                self._sums = torch.zeros(2 * d, dtype=torch.float64, device=rows.device)
            nat.rms_moments(rows, n, d, rows.stride(0), self._sums)
            parallel.global_moment_sums(self._sums, group)
            n_total = n * parallel.world_size(group)                  # equal shards
            nat.rms_merge(self._sums, n_total, d, self.running_mean, self.running_var, self.count)
        else:
            nat.rms_update(rows, n, d, rows.stride(0), self.running_mean, self.running_var, self.count)

    def update_from_moments(self, batch_mean, batch_var, batch_count):
        """Chan merge of externally computed batch moments (reference ppo.py:33-45).  Not on the training path -
        ``update()`` fuses the moment pass and this merge on the device - so plain tensor ops, in place."""
        m, v, c = update_mean_var_count_from_moments(self.running_mean, self.running_var, self.count,
                                                     torch.as_tensor(batch_mean, device=self.running_mean.device),
                                                     torch.as_tensor(batch_var, device=self.running_mean.device),
                                                     batch_count)
        self.running_mean.copy_(m)
        self.running_var.copy_(v)
        self.count.copy_(c)

    def normalize_into(self, rows: torch.Tensor, out: torch.Tensor, update: bool = True):
        """rows (n,d) with unit inner stride -> out (n,d') written in place (d' >= d, padding untouched)"""
        if update:
            self._update_rows(rows)
        n, d = rows.shape
        native.get(rows.device).rms_normalize(rows, n, d, rows.stride(0), self.running_mean, self.running_var,
                                              self.epsilon, out, out.stride(0))


def update_mean_var_count_from_moments(mean, var, count, batch_mean, batch_var, batch_count):
    """(new_mean, new_var, new_count) of Chan's parallel merge, in the reference's operation order
    (ppo.py:48-62): ``M2 = var*count + batch_var*batch_count + delta^2 * count * batch_count / tot``."""
    delta = batch_mean - mean
    tot_count = count + batch_count
    new_mean = mean + delta * batch_count / tot_count
    m2 = var * count + batch_var * batch_count + torch.square(delta) * count * batch_count / tot_count
    return new_mean, m2 / tot_count, tot_count


def layer_init(layer, std=np.sqrt(2), bias_const=0.0):
    torch.nn.init.orthogonal_(layer.weight, std)
    torch.nn.init.constant_(layer.bias, bias_const)
    return layer


class Agent(nn.Module):
    """Actor / critic MLPs + observation / value normalisers (reference ppo.py:71-123).

    ``critic`` and ``actor_mean`` are ordinary ``nn.Sequential`` stacks, so ``state_dict()``
    has the reference's 23 keys and shapes and loads into / from the reference ``Agent``.  Their
    parameters are *views into one flat fp32 buffer* (``self.flat``) laid out by
    ``catppo_mlp_layout_of`` (first-layer rows padded from D to a multiple of 16); the HIP
    kernels read and update that buffer directly.
    """

    def __init__(self, envs, hidden=DEFAULT_HIDDEN, device=None, mlp_precision: str = "fp32"):
        super().__init__()
        if mlp_precision not in ("fp32", "bf16"):
            raise ValueError(f"mlp_precision must be 'fp32' or 'bf16', got {mlp_precision!r}")
        self.mlp_precision = mlp_precision
        obs_shape = envs.unwrapped.single_observation_space["policy"].shape
        act_shape = envs.unwrapped.single_action_space.shape
        self.obs_dim, self.act_dim = int(np.array(obs_shape).prod()), int(np.prod(act_shape))
        self.hidden = tuple(int(h) for h in hidden)
        if device is None:
            device = torch.device("cuda" if torch.cuda.is_available() else "cpu")
        device = torch.device(device)
        self.shape = native.shape_of(self.obs_dim, self.act_dim, self.hidden, mfma_bf16=mlp_precision == "bf16")
        self.layout = native.layout_of(self.shape)

        def mlp(out_dim, out_std):
            dims = [self.obs_dim, *self.hidden]
            layers = []
            for i, o in zip(dims[:-1], dims[1:]):
                layers += [layer_init(nn.Linear(i, o)), nn.ELU()]
            layers.append(layer_init(nn.Linear(dims[-1], out_dim), std=out_std))
            return nn.Sequential(*layers)

        self.critic = mlp(1, 1.0)
        self.actor_mean = mlp(self.act_dim, 0.01)
        self.actor_logstd = nn.Parameter(torch.zeros(1, self.act_dim))
        self.obs_rms = RunningMeanStd(shape=obs_shape)
        self.value_rms = RunningMeanStd(shape=())
        self.flat = None
        self._tie(device)

    # -- flat buffer <-> module parameters ---------------------------------------------------
    def _tie(self, device):
        """move everything to ``device`` and re-point every parameter at its slice of ``flat``"""
        lay, L = self.layout, self.shape.n_hidden
        flat = torch.zeros(lay.n_flat, device=device)
        with torch.no_grad():
            v = flat[lay.off_logstd:lay.off_logstd + self.act_dim].view(1, self.act_dim)
            v.copy_(self.actor_logstd.detach())
            self.actor_logstd = nn.Parameter(v)
            for net, seq in ((0, self.critic), (1, self.actor_mean)):
                for l in range(L + 1):
                    lin = seq[2 * l]
                    out_f, in_f = lin.weight.shape
                    ld = lay.in_dim[l]
                    w = flat[lay.off_w[net][l]:lay.off_w[net][l] + out_f * ld].view(out_f, ld)[:, :in_f]
                    b = flat[lay.off_b[net][l]:lay.off_b[net][l] + out_f]
                    w.copy_(lin.weight.detach())
                    b.copy_(lin.bias.detach())
                    lin.weight, lin.bias = nn.Parameter(w), nn.Parameter(b)
        self.flat = flat
        for rms in (self.obs_rms, self.value_rms):
            for name in ("running_mean", "running_var", "count"):
                setattr(rms, name, getattr(rms, name).to(device))

    def _apply(self, fn, recurse=True):
        probe = fn(torch.zeros(1, device=self.flat.device))
        if probe.device != self.flat.device or probe.dtype != torch.float32:
            if probe.dtype != torch.float32:
                raise TypeError("Agent parameters are fp32 (the HIP kernels compute in fp32)")
            self._tie(probe.device)
        return self

    def load_state_dict(self, state_dict, strict: bool = True, assign: bool = False):
        # copies into the views, i.e. into the flat buffer (never re-assigns parameter storage)
        return super().load_state_dict(state_dict, strict=strict, assign=False)

    @property
    def n_params(self) -> int:
        return int(self.layout.n_params)

    # -- reference API --------------------------------------------------------------------------
    def _padded(self, x: torch.Tensor) -> torch.Tensor:
        x = x.reshape(-1, self.obs_dim)
        if self.layout.obs_pad == self.obs_dim and x.is_contiguous() and x.dtype == torch.float32:
            return x
        xp = torch.zeros(x.shape[0], self.layout.obs_pad, device=x.device)
        xp[:, :self.obs_dim] = x
        return xp

    def get_value(self, x):
        xp = self._padded(x)
        nat = native.get(xp.device)
        nat.mlp_reserve(self.shape, xp.shape[0])
        v = torch.empty(xp.shape[0], device=xp.device)
        nat.value(self.shape, self.flat, xp, xp.shape[0], v)
        return v.unsqueeze(1)

    def get_action_and_value(self, x, action=None, deterministic=False, eps=None):
        """(action, log_prob.sum(1), entropy.sum(1), value (B,1)); ``eps`` optionally supplies the
        N(0,1) noise of ``Normal.sample()`` (otherwise drawn from torch's generator)."""
        xp = self._padded(x)
        n = xp.shape[0]
        nat = native.get(xp.device)
        nat.mlp_reserve(self.shape, n)
        act = torch.empty(n, self.act_dim, device=xp.device)
        logp, val = torch.empty(n, device=xp.device), torch.empty(n, device=xp.device)
        given = None
        if action is not None:
            given = action.reshape(n, self.act_dim).float().contiguous()
        elif not deterministic and eps is None:
            eps = torch.randn(n, self.act_dim, device=xp.device)
        nat.policy_act(self.shape, self.flat, xp, n, None if (given is not None or deterministic) else eps,
                       act, logp, val, given_action=given)
        entropy = (0.5 + 0.5 * math.log(2 * math.pi) + self.actor_logstd.detach()).sum().expand(n)
        return (action if action is not None else act), logp, entropy, val.unsqueeze(1)

    def forward(self, x, deterministic=True):
        action, _, _, _ = self.get_action_and_value(self.obs_rms(x, update=False), deterministic=deterministic)
        return action


# ------------------------------------------------------------------------------------------------
class _JsonlWriter:
    """scalar writer used when tensorboard is not installed (same add_scalar surface)"""

    def __init__(self, log_dir):
        os.makedirs(log_dir, exist_ok=True)
        self._f = open(os.path.join(log_dir, "scalars.jsonl"), "a")

    def add_scalar(self, key, value, step):
        import json
        self._f.write(json.dumps({"key": key, "value": float(value), "step": int(step)}) + "\n")
        self._f.flush()


def _make_writer(ppo_cfg, run_path):
    if ppo_cfg.logger == "wandb":
        from rsl_rl.utils.wandb_utils import WandbSummaryWriter
        return WandbSummaryWriter(log_dir=run_path, flush_secs=10, cfg=ppo_cfg.to_dict())
    if ppo_cfg.logger == "tensorboard":
        try:
            from torch.utils.tensorboard import SummaryWriter as TensorboardSummaryWriter
            return TensorboardSummaryWriter(log_dir=run_path)
        except ImportError:
            return _JsonlWriter(run_path)
    raise AssertionError("logger type not found")


class PPOTrainer:
    """State of one ``PPO()`` run; ``run_iteration`` is one pass of the hot path."""

    def __init__(self, envs, ppo_cfg, run_path=None, writer=None, agent: Agent | None = None):
        c = ppo_cfg
        self.cfg, self.envs, self.run_path, self.writer = c, envs, run_path, writer
        if not torch.cuda.is_available():
            raise RuntimeError("PPO needs a HIP device (MI355X): every step of the update runs in libcatppo.so; "
                               "there is no CPU fallback")
        self.device = torch.device("cuda", torch.cuda.current_device())
        self.nat = native.get(self.device)
        self.T, self.N = int(c.num_steps), int(envs.unwrapped.num_envs)
        self.batch = self.T * self.N
        self.mb = int(c.minibatch_size)
        self.world, self.rank = parallel.world_size(), parallel.rank()
        hidden = tuple(getattr(c, "hidden", None) or DEFAULT_HIDDEN)
        self.agent = agent if agent is not None else Agent(
            envs, hidden=hidden, mlp_precision=str(getattr(c, "mlp_precision", "fp32"))).to(self.device)
        a = self.agent
        self.D, self.A, self.Dp = a.obs_dim, a.act_dim, a.layout.obs_pad
        if parallel.active():
            parallel.broadcast_(a.flat, src=0)                  # replicas start identical (cf. skrl ppo.py:126-131)
            a.obs_rms.dist_group = a.value_rms.dist_group = torch.distributed.group.WORLD
            cm = getattr(envs.unwrapped, "constraint_manager", None)
            if cm is not None and getattr(c, "dist_exact", True):
                cm.dist_group = torch.distributed.group.WORLD
        cm = getattr(envs.unwrapped, "constraint_manager", None)
        if cm is not None and hasattr(cm, "ensure_log_ring"):
            cm.ensure_log_ring(self.T + 2)      # the per-step episode logs are read after the rollout
        n_flat = a.layout.n_flat
        dev, T, N = self.device, self.T, self.N
        z = lambda *s, **k: torch.zeros(*s, device=dev, **k)
        self.grad, self.exp_avg, self.exp_avg_sq = z(n_flat), z(n_flat), z(n_flat)
        self.adam_step = 0
        # time-major rollout buffers; slot T holds the bootstrap observation / dones (reference keeps them in
        # next_obs / next_done / next_true_done)
        self.obs = z(T + 1, N, self.Dp)
        self.actions = z(T, N, self.A)
        self.logprobs, self.rewards, self.values = z(T, N), z(T, N), z(T, N)
        self.dones, self.true_dones = z(T + 1, N), z(T + 1, N)
        self.advantages, self.returns = z(T, N), z(T, N)
        self.values_n, self.returns_n = z(T, N), z(T, N)
        self.next_value = z(N)
        self.noise = z(T, N, self.A)
        self.diag = z(8)
        self.adv_stats = z(2)
        self.hp = native.PpoHparams(float(c.clip_coef), float(c.ent_coef), float(c.vf_coef), int(bool(c.norm_adv)),
                                    int(bool(c.clip_vloss)), 1.0 / (min(self.mb, self.batch) * self.world), 0)
        self.nat.mlp_reserve(a.shape, max(min(self.mb, self.batch), N))
        self.iteration = 0
        self.global_step = 0
        # first observation (ppo.py:186-189)
        first = envs.reset()[0]["policy"]
        a.obs_rms.normalize_into(self._rows(first), self.obs[0])
        self.phase_ms = {}

    @staticmethod
    def _rows(x):
        x = x if x.dtype == torch.float32 else x.float()
        x = x.reshape(x.shape[0], -1)
        return x if x.stride(1) == 1 else x.contiguous()

    # ------------------------------------------------------------------ rollout (ppo.py:201-230)
    def rollout(self, eps_fn=None):
        c, a, nat, T, N = self.cfg, self.agent, self.nat, self.T, self.N
        ep_infos = []
        if eps_fn is None:
            self.noise.normal_()                                 # all N(0,1) draws of the iteration at once
        for step in range(T):
            self.global_step += N * self.world
            eps = self.noise[step] if eps_fn is None else eps_fn(step)
            nat.policy_act(a.shape, a.flat, self.obs[step], N, eps, self.actions[step], self.logprobs[step],
                           self.values[step])
            next_obs, reward, next_done, timeouts, info = self.envs.step(self.actions[step])
            if (reward.dtype == torch.float32 and next_done.dtype == torch.float32 and timeouts.dtype == torch.bool
                    and reward.is_contiguous() and next_done.is_contiguous() and timeouts.is_contiguous()):
                nat.rollout_store(reward, next_done, timeouts, self.rewards[step], self.dones[step + 1],
                                  self.true_dones[step + 1])
            else:                                                # foreign env: dtype conversions in the copies
                self.rewards[step].copy_(reward)
                self.dones[step + 1].copy_(next_done)
                self.true_dones[step + 1].copy_(timeouts)
            if "episode" in info:
                ep_infos.append(info["episode"])
            elif "log" in info:
                packed = info.get("log_packed")
                if packed is None:
                    ep_infos.append(info["log"])
                else:                                 # (keys, device tensor) + the host-side scalars of the log
                    extra = {k: v for k, v in info["log"].items() if not isinstance(v, torch.Tensor)}
                    ep_infos.append((packed[0], packed[1], extra))
            info["true_dones"] = timeouts
            a.obs_rms.normalize_into(self._rows(next_obs["policy"]), self.obs[step + 1])
            if "time_outs" in info:
                if info["time_outs"].any():
                    print("time outs", info["time_outs"].sum())
                    exit(0)
        return ep_infos

    # ------------------------------------------------------------------ GAE + normalisers (:251-288)
    def compute_returns(self):
        c, a, nat, T, N = self.cfg, self.agent, self.nat, self.T, self.N
        nat.value(a.shape, a.flat, self.obs[T], N, self.next_value)
        nat.gae(self.rewards, self.values, self.dones[:T], self.true_dones[:T], self.next_value, self.dones[T],
                self.true_dones[T], c.gamma, c.gae_lambda, self.advantages, self.returns)
        # value_rms is updated with the values and then, a second time, with the returns (:287-288)
        a.value_rms.normalize_into(self.values.view(-1, 1), self.values_n.view(-1, 1))
        a.value_rms.normalize_into(self.returns.view(-1, 1), self.returns_n.view(-1, 1))

    # ------------------------------------------------------------------ update (:294-354)
    def update(self, perm_fn=None):
        c, a, nat = self.cfg, self.agent, self.nat
        B, M = self.batch, min(self.mb, self.batch)
        b_obs = self.obs[:self.T].view(B, self.Dp)
        b_act = self.actions.view(B, self.A)
        b_logp, b_adv = self.logprobs.view(-1), self.advantages.view(-1)
        b_ret, b_val = self.returns_n.view(-1), self.values_n.view(-1)
        vmean, vvar = a.value_rms.running_mean, a.value_rms.running_var
        self.diag.zero_()
        E = int(c.updates_epochs)
        n_mb = (B + M - 1) // M
        exact_adv = parallel.active() and bool(c.norm_adv) and getattr(c, "dist_exact", True)
        perms = [torch.randperm(B, device=self.device) if perm_fn is None else perm_fn(e) for e in range(E)]
        if exact_adv:
            # minibatch advantage mean / unbiased std over ALL ranks (ppo.py:316-318): the moments of every
            # minibatch of the iteration in one launch per epoch, ONE all-reduce, one finishing launch
            if not hasattr(self, "_adv_mom") or self._adv_mom.shape[0] != E * n_mb:
                self._adv_mom = torch.zeros(E * n_mb, 3, dtype=torch.float64, device=self.device)
                self._adv_stats_all = torch.zeros(E * n_mb, 2, device=self.device)
            for e in range(E):
                nat.adv_moments(b_adv, perms[e], M, self._adv_mom[e * n_mb:(e + 1) * n_mb])
            parallel.allreduce_sum_(self._adv_mom)
            nat.adv_stats(self._adv_mom, E * n_mb, self._adv_stats_all)
        self.hp.adv_stats_external = int(exact_adv)
        if not hasattr(self, "_x_g"):
            # packed epoch buffers: one gather launch per epoch, minibatch k = contiguous slice k
            self._parts = (M + nat.GATHER_ROWS - 1) // nat.GATHER_ROWS
            self._x_g = torch.empty(B, self.Dp, device=self.device)
            self._act_g = torch.empty(B, self.A, device=self.device)
            self._scal_g = torch.empty(4 * B, device=self.device)
            self._advp_g = torch.empty(n_mb * self._parts * 2, dtype=torch.float64, device=self.device)
        for epoch in range(E):
            nat.ppo_gather(a.shape, b_obs, b_act, b_logp, b_adv, b_ret, b_val, perms[epoch], M, self._x_g,
                           self._act_g, self._scal_g, self._advp_g)
            for k, start in enumerate(range(0, B, M)):
                m = min(M, B - start)
                self.hp.inv_global_batch = 1.0 / (m * self.world)
                adv_stats = self._adv_stats_all[epoch * n_mb + k] if exact_adv else None
                nat.ppo_minibatch_grad_packed(a.shape, self.hp, a.flat, self._x_g[start:], self._act_g[start:],
                                              self._scal_g[4 * start:], self._advp_g[2 * k * self._parts:], m,
                                              vmean, vvar, adv_stats, self.grad, self.diag)
                parallel.allreduce_sum_(self.grad)              # RCCL SUM of the flat gradient over xGMI
                self.adam_step += 1
                nat.clip_adam(a.flat, self.grad, self.exp_avg, self.exp_avg_sq, a.layout.n_flat,
                              c.max_grad_norm, self.lr, 0.9, 0.999, 1e-5, self.adam_step)

    # ------------------------------------------------------------------ one iteration
    def run_iteration(self, eps_fn=None, perm_fn=None, log: bool = True):
        c = self.cfg
        self.iteration += 1
        it = self.iteration
        self.lr = float(c.learning_rate)
        if c.anneal_lr:
            frac = 1.0 - (it - 1.0) / c.num_iterations
            self.lr = frac * c.learning_rate
        ep_infos = self.rollout(eps_fn)
        self.compute_returns()
        self.update(perm_fn)
        # slot T becomes slot 0 of the next rollout (next_obs / next_done / next_true_done carry over)
        self.obs[0].copy_(self.obs[self.T])
        self.dones[0].copy_(self.dones[self.T])
        self.true_dones[0].copy_(self.true_dones[self.T])
        if not log:
            return None
        if self.world > 1:
            parallel.allreduce_sum_(self.diag)
            self.diag[7] /= self.world
        d = self.diag.cpu().numpy()                              # the one host sync of the iteration
        n_upd = max(d[7], 1.0)
        stats = {"mean_pg_loss": d[0] / n_upd, "mean_v_loss": d[1] / n_upd, "mean_entropy_loss": d[2] / n_upd,
                 "mean_surrogate_loss": d[3] / n_upd, "approx_kl": d[4] / n_upd, "old_approx_kl": d[5] / n_upd,
                 "clipfrac": d[6] / n_upd, "learning_rate": self.lr}
        if self.writer is not None and self.rank == 0:
            self._log_episode_infos(ep_infos, it)
            w = self.writer
            w.add_scalar("Loss/mean_pg_loss", stats["mean_pg_loss"], it)
            w.add_scalar("Loss/mean_entropy_loss", stats["mean_entropy_loss"], it)
            w.add_scalar("Loss/mean_v_loss", stats["mean_v_loss"], it)
            w.add_scalar("Loss/mean_surrogate_loss", stats["mean_surrogate_loss"], it)
            w.add_scalar("Loss/learning_rate", self.lr, it)
        if self.run_path is not None and self.rank == 0 and (it + 1) % c.save_interval == 0:
            torch.save(self.agent.state_dict(), f"{self.run_path}/model_{it}.pt")     # same off-by-one naming
            print("Saved model")
        return stats

    def _log_episode_infos(self, ep_infos, it):
        """mean of every logged key over the steps of the rollout (ppo.py:233-248)"""
        if not ep_infos:
            return
        first = ep_infos[0]
        if isinstance(first, tuple):                             # packed (keys, tensor, host scalars) from CaTEnv
            keys = first[0]
            vals = torch.stack([e[1] for e in ep_infos]).mean(0).cpu().numpy()
            items = list(zip(keys, vals))
            for key in first[2]:                                 # e.g. "Curriculum/<term>" (reference logs them too)
                vs = [float(e[2][key]) for e in ep_infos if key in e[2]]
                items.append((key, sum(vs) / len(vs)))
        else:
            items = []
            for key in first:
                vs = []
                for ep in ep_infos:
                    if key not in ep:
                        continue
                    v = ep[key]
                    v = v if isinstance(v, torch.Tensor) else torch.tensor([float(v)])
                    vs.append(v.reshape(-1).to(self.device).float())
                items.append((key, float(torch.cat(vs).mean())))
        for key, value in items:
            self.writer.add_scalar(key if "/" in key else "Episode/" + key, value, it)


def PPO(envs, ppo_cfg, run_path):
    """Train with CleanRL-style PPO on CaT float dones (reference ppo.py:126-372)."""
    writer = _make_writer(ppo_cfg, run_path)
    if not os.path.exists(run_path):
        os.makedirs(run_path)
    trainer = PPOTrainer(envs, ppo_cfg, run_path, writer)
    print(f"Starting training for {ppo_cfg.num_iterations} steps")
    t0 = time.time()
    for _ in range(1, int(ppo_cfg.num_iterations) + 1):
        trainer.run_iteration()
    torch.cuda.synchronize()
    dt = time.time() - t0
    print(f"[PPO] {trainer.global_step} env steps in {dt:.2f} s ({trainer.global_step / max(dt, 1e-9):,.0f} env-steps/s)")
    return trainer
