"""ORACLE (test infrastructure, NOT product code) - CleanRL-PPO path, torch CPU fp32.

CPU restatement of the reference's PPO arithmetic.  The reference itself is eager
PyTorch, so the restatement uses torch CPU ops (and torch autograd / torch.optim.Adam
for the backward + optimiser, which is exactly what the reference calls); what is
restated here is the *algorithm*: op order, which statistics are updated when, what
is clipped, how float dones enter GAE.  Only ``tests/``, ``__graft_entry__.smoke()``
and ``bench.py``'s ``cpu_baseline`` leg may import this module.

Pinned: checked against vectors produced by the reference's own ``RunningMeanStd``,
``Agent``, GAE loop, minibatch update and the full ``PPO()`` driver run in the build
container (``tests/golden/gen_golden.py`` -> ``tests/golden/ppo_*.npz``; test:
``tests/test_oracle_golden.py``).

Reference: /root/reference/exts/cat_envs/cat_envs/tasks/utils/cleanrl/ppo.py
  :12-62    RunningMeanStd + Chan merge
  :71-123   Agent (two ELU MLPs, state independent log-std, Normal log-prob/entropy)
  :251-277  GAE with float dones and separate time-out mask
  :280-354  flatten, value_rms x2, epochs x minibatches, clipped losses, clip-grad, Adam
"""
from __future__ import annotations

import math

import numpy as np
import torch
import torch.nn.functional as F

HALF_LOG_2PI = math.log(math.sqrt(2 * math.pi))


# ------------------------------------------------------------------ running mean / std
class RMSOracle:
    """ppo.py:12-62.  mean=0, var=1, count=1 initially (count is NOT epsilon)."""

    def __init__(self, shape=(), epsilon: float = 1e-8):
        self.mean = torch.zeros(shape)
        self.var = torch.ones(shape)
        self.count = torch.ones(())
        self.eps = epsilon

    def update(self, x: torch.Tensor) -> None:
        bm = torch.mean(x, dim=0)
        bv = torch.var(x, correction=0, dim=0)
        n = x.shape[0]
        delta = bm - self.mean
        tot = self.count + n
        new_mean = self.mean + delta * n / tot
        m2 = self.var * self.count + bv * n + torch.square(delta) * self.count * n / tot
        self.mean, self.var, self.count = new_mean, m2 / tot, tot

    def normalize(self, x: torch.Tensor) -> torch.Tensor:
        return (x - self.mean) / torch.sqrt(self.var + self.eps)

    def __call__(self, x, update=True):
        if update:
            self.update(x)
        return self.normalize(x)

    def state(self):
        return {"running_mean": self.mean.clone(), "running_var": self.var.clone(),
                "count": self.count.clone()}


# ------------------------------------------------------------------ actor / critic
def _q(t: torch.Tensor) -> torch.Tensor:
    return t.to(torch.bfloat16).to(torch.float32)


class _BF16OperandLinear(torch.autograd.Function):
    """Restatement of the build's bf16 mode (catppo_mlp_shape.mfma_bf16, BASELINE config 5 - NOT a reference
    code path): every hidden-layer GEMM rounds both operands to bf16 (RNE) and accumulates in fp32, in the
    forward (x.w^T), the data gradient (gy.w) and the weight gradient (gy^T.x); bias and its gradient fp32."""

    @staticmethod
    def forward(ctx, x, w, b):
        ctx.save_for_backward(x, w)
        return _q(x) @ _q(w).t() + b

    @staticmethod
    def backward(ctx, gy):
        x, w = ctx.saved_tensors
        return _q(gy) @ _q(w), _q(gy).t() @ _q(x), gy.sum(0)


def mlp_forward(x: torch.Tensor, layers, bf16_hidden: bool = False) -> torch.Tensor:
    """layers = [(W(out,in), b), ...]; ELU(alpha=1) between, none after the last."""
    h = x
    for i, (w, b) in enumerate(layers):
        last = i + 1 == len(layers)
        h = _BF16OperandLinear.apply(h, w, b) if (bf16_hidden and not last) else F.linear(h, w, b)
        if not last:
            h = F.elu(h)
    return h


def gaussian_logp_entropy(mean, logstd, action):
    """Normal(mean, exp(logstd)).log_prob(action).sum(1), .entropy().sum(1)   (ppo.py:106-117)."""
    logstd = logstd.expand_as(mean)
    std = torch.exp(logstd)
    var = std ** 2
    logp = (-((action - mean) ** 2) / (2 * var) - std.log() - HALF_LOG_2PI).sum(1)
    ent = (0.5 + 0.5 * math.log(2 * math.pi) + torch.log(std)).sum(1)
    return logp, ent


class AgentOracle:
    """Parameters as plain tensors under the reference's 23 state_dict keys."""

    def __init__(self, obs_dim: int, act_dim: int, hidden=(512, 256, 128), seed: int = 0, bf16_hidden: bool = False):
        self.bf16_hidden = bool(bf16_hidden)
        g = torch.Generator().manual_seed(seed)
        dims = [obs_dim, *hidden]
        self.p: dict[str, torch.Tensor] = {"actor_logstd": torch.zeros(1, act_dim)}
        for net, out_dim, out_gain in (("critic", 1, 1.0), ("actor_mean", act_dim, 0.01)):
            sizes = list(zip(dims[:-1], dims[1:])) + [(dims[-1], out_dim)]
            for li, (fan_in, fan_out) in enumerate(sizes):
                gain = out_gain if li == len(sizes) - 1 else math.sqrt(2)
                w = torch.empty(fan_out, fan_in)
                torch.nn.init.orthogonal_(w, gain, generator=g)
                self.p[f"{net}.{2 * li}.weight"] = w
                self.p[f"{net}.{2 * li}.bias"] = torch.zeros(fan_out)
        self.obs_rms = RMSOracle((obs_dim,))
        self.value_rms = RMSOracle(())

    def layers(self, net: str):
        idx = sorted({int(k.split(".")[1]) for k in self.p if k.startswith(net + ".")})
        return [(self.p[f"{net}.{i}.weight"], self.p[f"{net}.{i}.bias"]) for i in idx]

    def parameters(self):
        # registration order of the reference Agent: logstd, critic.*, actor_mean.*
        keys = ["actor_logstd"]
        for net in ("critic", "actor_mean"):
            for i in sorted({int(k.split(".")[1]) for k in self.p if k.startswith(net + ".")}):
                keys += [f"{net}.{i}.weight", f"{net}.{i}.bias"]
        return [self.p[k] for k in keys]

    def load(self, sd: dict) -> None:
        for k, v in sd.items():
            v = torch.as_tensor(np.asarray(v)) if not isinstance(v, torch.Tensor) else v
            if k.startswith("obs_rms.") or k.startswith("value_rms."):
                rms = self.obs_rms if k.startswith("obs_rms.") else self.value_rms
                name = k.split(".")[1]
                setattr(rms, {"running_mean": "mean", "running_var": "var", "count": "count"}[name],
                        v.clone().float())
            else:
                self.p[k] = v.clone().float()

    def state_dict(self) -> dict:
        sd = {k: v.detach().clone() for k, v in self.p.items()}
        for pre, rms in (("obs_rms", self.obs_rms), ("value_rms", self.value_rms)):
            for k, v in rms.state().items():
                sd[f"{pre}.{k}"] = v
        return sd

    def get_value(self, x):
        return mlp_forward(x, self.layers("critic"), self.bf16_hidden)

    def get_action_and_value(self, x, action=None, eps=None, deterministic=False):
        mean = mlp_forward(x, self.layers("actor_mean"), self.bf16_hidden)
        if action is None:
            if deterministic:
                action = mean
            else:
                std = torch.exp(self.p["actor_logstd"].expand_as(mean))
                if eps is None:
                    eps = torch.randn_like(mean)
                action = mean + std * eps       # Normal.sample(): loc + scale * N(0,1)
        logp, ent = gaussian_logp_entropy(mean, self.p["actor_logstd"], action)
        return action, logp, ent, self.get_value(x)


# ------------------------------------------------------------------ GAE
def gae(rewards, values, dones, true_dones, next_value, next_done, next_true_done,
        gamma: float, gae_lambda: float):
    """ppo.py:251-277 (time-major (T,N) inputs).  Returns (advantages, returns)."""
    T = rewards.shape[0]
    adv = torch.zeros_like(rewards)
    last = 0
    for t in reversed(range(T)):
        if t == T - 1:
            nn_, tn_, nv = 1.0 - next_done, 1 - next_true_done, next_value
        else:
            nn_, tn_, nv = 1.0 - dones[t + 1], 1 - true_dones[t + 1], values[t + 1]
        delta = rewards[t] + gamma * nv * nn_ * tn_ - values[t]
        adv[t] = last = delta + gamma * gae_lambda * nn_ * tn_ * last
    return adv, adv + values


def gae_numpy_exact(rew, val, done, tdone, nv, nd, ntd, gamma: float, gae_lambda: float):
    """Same recurrence with every fp32 rounding explicit:
       delta = fl(fl(r + fl(fl(fl(g*nv)*nn)*tn)) - v);  A = fl(delta + fl(fl(fl(gl*nn)*tn)*A'))."""
    f = np.float32
    T = rew.shape[0]
    g, gl = f(gamma), f(gamma * gae_lambda)
    adv = np.zeros_like(rew, dtype=f)
    last = np.zeros(rew.shape[1], f)
    for t in reversed(range(T)):
        if t == T - 1:
            nn_, tn_, nxt = (f(1) - nd).astype(f), (f(1) - ntd).astype(f), nv
        else:
            nn_, tn_, nxt = (f(1) - done[t + 1]).astype(f), (f(1) - tdone[t + 1]).astype(f), val[t + 1]
        x = (((g * nxt).astype(f) * nn_).astype(f) * tn_).astype(f)
        delta = ((rew[t] + x).astype(f) - val[t]).astype(f)
        c = ((gl * nn_).astype(f) * tn_).astype(f)
        last = (delta + (c * last).astype(f)).astype(f)
        adv[t] = last
    return adv, (adv + val).astype(f)


def gae_rl_games(fdones, last_values, mb_fdones, mb_values, mb_rewards, gamma: float, tau: float):
    """rl_games ``A2CBase.discount_values`` as called with float dones by the reference
    (rl_games/cat_common.py:96-103).  rl_games (pinned 1.6.1 in the reference's setup) is NOT under
    /root/reference, so this restates its published recurrence - PARITY UNPINNED against rl_games itself;
    it is the CleanRL recurrence above without the time-out channel and is tested bit-equal to
    ``gae(..., true_dones = 0)``, which the reference goldens do pin.  Returns (mb_advs, mb_returns)."""
    T = mb_rewards.shape[0]
    adv = torch.zeros_like(mb_rewards)
    last = 0
    for t in reversed(range(T)):
        if t == T - 1:
            nn_, nv = 1.0 - fdones, last_values
        else:
            nn_, nv = 1.0 - mb_fdones[t + 1], mb_values[t + 1]
        delta = mb_rewards[t] + gamma * nv * nn_ - mb_values[t]
        adv[t] = last = delta + gamma * tau * nn_ * last
    return adv, adv + mb_values


def value_bootstrap(rewards, values, time_outs, gamma: float):
    """rl_games/cat_common.py:59-64: ``shaped_rewards += gamma * values * time_outs.float()``"""
    return rewards + gamma * values * time_outs.float()


def gae_skrl(rewards, dones, values, last_values, discount_factor: float = 0.99, lambda_coefficient: float = 0.95):
    """skrl/ppo.py:397-442 ``compute_gae`` with the CaT float ``not_dones = 1 - dones``.
    Returns (returns, normalised advantages, raw advantages).  Pinned by tests/golden/skrl_gae.npz, which
    gen_golden.py produces by executing the reference's own nested function."""
    advantage = 0
    advantages = torch.zeros_like(rewards)
    not_dones = 1 - dones
    T = rewards.shape[0]
    for i in reversed(range(T)):
        nv = values[i + 1] if i < T - 1 else last_values
        advantage = rewards[i] - values[i] + discount_factor * not_dones[i] * (nv + lambda_coefficient * advantage)
        advantages[i] = advantage
    returns = advantages + values
    normed = (advantages - advantages.mean()) / (advantages.std() + 1e-8)
    return returns, normed, advantages


# ------------------------------------------------------------------ minibatch update
def ppo_minibatch_loss(agent: AgentOracle, mb_obs, mb_actions, mb_logprobs, mb_adv,
                       mb_returns_n, mb_values_n, cfg):
    """ppo.py:300-344 on an already gathered minibatch.  Returns (loss, stats dict)."""
    _, newlogprob, entropy, newvalue = agent.get_action_and_value(mb_obs, mb_actions)
    logratio = newlogprob - mb_logprobs
    ratio = logratio.exp()
    with torch.no_grad():
        old_approx_kl = (-logratio).mean()
        approx_kl = ((ratio - 1) - logratio).mean()
        clipfrac = ((ratio - 1.0).abs() > cfg["clip_coef"]).float().mean()
    adv = mb_adv
    if cfg["norm_adv"]:
        adv = (adv - adv.mean()) / (adv.std() + 1e-8)
    pg1 = -adv * ratio
    pg2 = -adv * torch.clamp(ratio, 1 - cfg["clip_coef"], 1 + cfg["clip_coef"])
    pg_loss = torch.max(pg1, pg2).mean()
    newvalue = agent.value_rms.normalize(newvalue.view(-1))
    if cfg["clip_vloss"]:
        unclipped = (newvalue - mb_returns_n) ** 2
        v_clipped = mb_values_n + torch.clamp(newvalue - mb_values_n, -cfg["clip_coef"], cfg["clip_coef"])
        clipped = (v_clipped - mb_returns_n) ** 2
        v_loss = 0.5 * torch.max(unclipped, clipped).mean()
    else:
        v_loss = 0.5 * ((newvalue - mb_returns_n) ** 2).mean()
    ent = entropy.mean()
    loss = pg_loss - cfg["ent_coef"] * ent + v_loss * cfg["vf_coef"]
    with torch.no_grad():
        # distance of the closest sample to a point where the GRADIENT of the clipped objectives is discontinuous
        # (|ratio - 1| = clip for the surrogate, |newvalue - old value| = clip for the clipped value loss): a sample
        # closer to it than the rounding noise between two implementations (~1e-6) may take the other branch there, and
        # from that optimiser step on the two parameter trajectories differ by more than rounding.  Test diagnostics only.
        margin = ((ratio - 1.0).abs() - cfg["clip_coef"]).abs().min()
        if cfg["clip_vloss"]:
            margin = torch.minimum(margin, ((newvalue - mb_values_n).abs() - cfg["clip_coef"]).abs().min())
    return loss, {"pg_loss": pg_loss.detach(), "v_loss": v_loss.detach(), "entropy": ent.detach(),
                  "loss": loss.detach(), "approx_kl": approx_kl, "old_approx_kl": old_approx_kl,
                  "clipfrac": clipfrac, "clip_boundary_margin": margin,
                  # per-sample quantities the clip branches depend on (PPOOracle.trace: the branch-flip test)
                  "ratio": ratio.detach(), "newvalue_n": newvalue.detach()}


class PPOOracle:
    """One-iteration-at-a-time restatement of ``PPO()`` (ppo.py:126-372).

    ``env`` follows the reference env protocol (reset() / step(action) 5-tuple).
    Noise and permutations can be injected (``eps_fn(step)``, ``perm_fn(epoch)``) so
    that the HIP path and this oracle see identical randomness.
    """

    DEFAULT_CFG = dict(learning_rate=3.0e-4, num_steps=24, num_iterations=2000, gamma=0.99,
                       gae_lambda=0.95, updates_epochs=5, minibatch_size=16384, clip_coef=0.2,
                       ent_coef=0.001, vf_coef=2.0, max_grad_norm=1.0, norm_adv=True,
                       clip_vloss=True, anneal_lr=True)

    def __init__(self, env, num_envs, obs_dim, act_dim, cfg=None, hidden=(512, 256, 128), seed=0,
                 agent: AgentOracle | None = None, rollout_dtype: str = "fp32"):
        self.cfg = dict(self.DEFAULT_CFG)
        self.cfg.update(cfg or {})
        # "fp16": restatement of the build's fp16 rollout planes (BASELINE config 5, not a reference code path):
        # rewards / values / dones / true_dones / next_value are rounded to IEEE half (RNE) when stored, GAE runs in
        # fp32 on the widened values, advantages and returns (= fl32(A + v)) are rounded to half when stored
        self.q = (lambda t: t.half().float()) if rollout_dtype == "fp16" else (lambda t: t)
        self.env, self.N, self.D, self.A = env, num_envs, obs_dim, act_dim
        self.agent = agent or AgentOracle(obs_dim, act_dim, hidden, seed)
        self.params = [p.requires_grad_(True) for p in self.agent.parameters()]
        self.opt = torch.optim.Adam(self.params, lr=self.cfg["learning_rate"], eps=1e-5)
        T = self.cfg["num_steps"]
        z = lambda *s: torch.zeros(*s)
        self.obs, self.actions = z(T, self.N, self.D), z(T, self.N, self.A)
        self.logprobs, self.rewards, self.dones = z(T, self.N), z(T, self.N), z(T, self.N)
        self.true_dones, self.values = z(T, self.N), z(T, self.N)
        self.next_obs = self.agent.obs_rms(env.reset()[0]["policy"])
        self.next_done = torch.zeros(self.N)
        self.next_true_done = torch.zeros(self.N)
        self.iteration = 0
        self.timers = {"rollout": 0.0, "env": 0.0, "gae": 0.0, "update": 0.0}

    def run_iteration(self, eps_fn=None, perm_fn=None, actions_fn=None):
        """``actions_fn(step)``, if given, supplies the action taken at each step (e.g. the device's),
        so that action-dependent constraint terms see identical inputs on both sides."""
        import time
        c = self.cfg
        T, N = c["num_steps"], self.N
        self.iteration += 1
        if c["anneal_lr"]:
            frac = 1.0 - (self.iteration - 1.0) / c["num_iterations"]
            self.opt.param_groups[0]["lr"] = frac * c["learning_rate"]
        t0 = time.perf_counter()
        q = self.q
        for step in range(T):
            self.obs[step], self.dones[step] = self.next_obs, q(self.next_done)
            self.true_dones[step] = self.next_true_done
            with torch.no_grad():
                eps = None if eps_fn is None else eps_fn(step)
                given = None if actions_fn is None else actions_fn(step)
                action, logprob, _, value = self.agent.get_action_and_value(self.next_obs, action=given, eps=eps)
                if given is not None and eps is not None:
                    # the action THIS side would have sampled from the same noise (mu + sigma * eps, ppo.py:111): lets a
                    # caller that replays the device's actions still check the device's sampling arithmetic per step
                    if not hasattr(self, "own_actions"):
                        self.own_actions = torch.zeros_like(self.actions)
                    self.own_actions[step] = self.agent.get_action_and_value(self.next_obs, eps=eps)[0]
            self.values[step], self.actions[step], self.logprobs[step] = q(value.flatten()), action, logprob
            te = time.perf_counter()
            nobs, rew, nd, timeouts, _info = self.env.step(action)
            self.rewards[step] = q(rew)
            self.timers["env"] += time.perf_counter() - te
            self.next_done = nd.to(torch.float)
            self.next_obs = self.agent.obs_rms(nobs["policy"])
            self.next_true_done = timeouts.float()
        t1 = time.perf_counter()
        with torch.no_grad():
            nv = q(self.agent.get_value(self.next_obs).reshape(1, -1))
            adv, returns = gae(self.rewards, self.values, self.dones, self.true_dones, nv[0],
                               q(self.next_done), self.next_true_done, c["gamma"], c["gae_lambda"])
            adv, returns = q(adv), q(returns)
        t2 = time.perf_counter()
        b_obs, b_logp = self.obs.reshape(-1, self.D), self.logprobs.reshape(-1)
        b_act, b_adv = self.actions.reshape(-1, self.A), adv.reshape(-1)
        b_values = self.agent.value_rms(self.values.reshape(-1))      # update #1, then normalise
        b_returns = self.agent.value_rms(returns.reshape(-1))          # update #2, then normalise
        B, M = T * N, c["minibatch_size"]
        sums = {"pg_loss": 0.0, "entropy": 0.0, "v_loss": 0.0, "loss": 0.0}
        last_stats = None
        if getattr(self, "trace", False):
            self.step_trace = []          # (of the current iteration)
        for epoch in range(c["updates_epochs"]):
            inds = torch.randperm(B) if perm_fn is None else perm_fn(epoch)
            # a list of index tensors = the minibatches themselves (env-sharded runs: global minibatch k is the
            # union of the ranks' k-th local minibatches, whose sizes need not be a fixed stride of one permutation)
            mbs = list(inds) if isinstance(inds, (list, tuple)) else [inds[s0:s0 + M] for s0 in range(0, B, M)]
            n_mbs = len(mbs) if isinstance(inds, (list, tuple)) else None
            for mb in mbs:
                loss, st = ppo_minibatch_loss(self.agent, b_obs[mb], b_act[mb], b_logp[mb], b_adv[mb],
                                              b_returns[mb], b_values[mb], c)
                for k in sums:
                    sums[k] = sums[k] + st[k]
                # closest approach of any sample to a clip boundary over the optimiser steps of this oracle (diagnostic)
                self.clip_boundary_margin = min(getattr(self, "clip_boundary_margin", float("inf")),
                                                float(st["clip_boundary_margin"]))
                self.opt.zero_grad()
                loss.backward()
                torch.nn.utils.clip_grad_norm_(self.params, c["max_grad_norm"])
                self.opt.step()
                last_stats = st
                if getattr(self, "trace", False):
                    # per optimiser step: the minibatch, what its clip branches were decided on (evaluated at the
                    # parameters BEFORE the step) and the parameters AFTER it, reference registration order
                    if not hasattr(self, "step_trace"):
                        self.step_trace = []
                    self.step_trace.append(dict(mb=mb.clone(), ratio=st["ratio"].clone(), newvalue_n=st["newvalue_n"].clone(),
                                                old_values_n=b_values[mb].clone(), returns_n=b_returns[mb].clone(),
                                                params=torch.cat([q_.detach().reshape(-1) for q_ in self.agent.parameters()]).clone()))
        t3 = time.perf_counter()
        self.timers["rollout"] += t1 - t0
        self.timers["gae"] += t2 - t1
        self.timers["update"] += t3 - t2
        n_upd = c["updates_epochs"] * (B / M if n_mbs is None else n_mbs)
        out = {f"mean_{k}": float(v) / n_upd for k, v in sums.items()}
        out["lr"] = self.opt.param_groups[0]["lr"]
        out["advantages"], out["returns"] = adv, returns
        out["last"] = last_stats
        return out
