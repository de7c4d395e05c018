"""CleanRL-style PPO for CaT (float dones) on MI355X.

Same entry points as the reference (cleanrl/ppo.py): ``RunningMeanStd``, ``layer_init``,
``Agent(envs)`` with ``get_value / get_action_and_value / forward`` and the 23-key
``state_dict``, and ``PPO(envs, ppo_cfg, run_path)``.  The algorithm is the reference's, step
for step (rollout bookkeeping :201-230, GAE with float dones and a separate time-out mask
:251-277, value normaliser updated twice :287-288, epochs x minibatches with clipped policy /
value losses, global-norm clip, Adam, linear lr anneal :294-354); every tensor operation of it
runs in libcatppo.so (hand-written HIP, see include/catppo.h):

    iteration      catppo_iter_begin                          (iteration counter, lr schedule: device state)
    rollout step   catppo_policy_step                         (ONE launch at 2049-4096 rows - all hidden layers + heads
                                                               per 32-row workgroup - else 3 GEMM launches + head:
                                                               actions/logprobs/values[step]; Philox action noise inside)
                   env.step_into -> catppo_rollout_pre/_post  (simulator state advance, terms, CaT, resets, buffer rows,
                                                               obs normaliser: 3 launches; a foreign env takes the
                                                               unfused calls)
    after rollout  catppo_value, catppo_gae, 2x (catppo_rms_update + catppo_rms_normalize)
    epoch          catppo_ppo_gather_ex                       (keyed on-device permutation, no index array)
    minibatch      catppo_ppo_minibatch_grad_packed  [catppo_allreduce: RCCL SUM of the flat gradient]
                   catppo_clip_adam_dev                       (lr / step count from the device state)
                   single process: both in one call, catppo_ppo_minibatch_step_packed (one launch fewer)
    [KL-adaptive]  catppo_kl_mean [catppo_allreduce] catppo_kl_adaptive_lr   after every epoch, no host sync
The update phase is replayed from a hipGraph (``graph_update``; collectives of an env-sharded run included, with a
reported eager fallback when the capture fails).

There is no host synchronisation inside an iteration (the reference syncs once per
constraint term per env step and once per minibatch); diagnostics accumulate on the device
and are read once per iteration.  Parameters, gradients and Adam moments live in ONE flat
fp32 buffer each, so the gradient exchange of an env-sharded run is a single all-reduce.
"""
from __future__ import annotations

import ctypes as C
import math
import os
import time

import numpy as np
import torch
import torch.nn as nn

from cat_envs import native, parallel

DEFAULT_HIDDEN = (512, 256, 128)     # reference Agent (ppo.py:78-95)
#: arithmetic of the hidden-layer GEMMs -> catppo_mlp_shape.mfma_bf16
MLP_PRECISIONS = {"fp32": 0, "bf16": 1, "bf16x3": 2}


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
        x = obs if obs.dtype in (torch.float32, torch.float16) else obs.float()
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
                self._n_global = {}
            nat.rms_moments_ex(rows, n, d, rows.stride(0), self._sums)
            parallel.global_moment_sums(self._sums, group)
            # true global row count (shards may differ by one env): exchanged once per batch size, not assumed
            if n not in self._n_global:
                cnt = torch.tensor([float(n)], dtype=torch.float64, device=rows.device)
                parallel.allreduce_sum_(cnt, group)
                self._n_global[n] = float(cnt.item())
            nat.rms_merge(self._sums, self._n_global[n], d, self.running_mean, self.running_var, self.count)
        else:
            nat.rms_update_ex(rows, n, d, rows.stride(0), self.running_mean, self.running_var, self.count)

    def update_normalize_pair(self, rows_a, out_a, rows_b, out_b):
        """``normalize_into(rows_a, out_a)`` followed by ``normalize_into(rows_b, out_b)`` (the value normaliser's two
        updates per iteration, reference ppo.py:287-288) with ONE exchange between ranks: the batch moments of the two
        batches do not depend on each other - only the two Chan merges are sequential - so an env-sharded run sums
        both moment pairs in a single all-reduce (round 5; one collective per iteration instead of two).  Same
        arithmetic, same order, as the two separate calls."""
        group = self.dist_group
        if group is None or not parallel.active(group) or os.environ.get("CATPPO_VRMS_PAIR", "1") == "0":   # (A/B switch)
            self.normalize_into(rows_a, out_a)
            self.normalize_into(rows_b, out_b)
            return
        nat = native.get(rows_a.device)
        (na, d), (nb, db) = rows_a.shape, rows_b.shape
        assert d == db == self.dim
        if not hasattr(self, "_sums_pair"):
            self._sums_pair = torch.zeros(4 * d, dtype=torch.float64, device=rows_a.device)
            self._n_global = getattr(self, "_n_global", {})
        sa, sb = self._sums_pair[:2 * d], self._sums_pair[2 * d:]
        nat.rms_moments_ex(rows_a, na, d, rows_a.stride(0), sa)
        nat.rms_moments_ex(rows_b, nb, d, rows_b.stride(0), sb)
        parallel.global_moment_sums(self._sums_pair, group)
        for n in (na, nb):
            if n not in self._n_global:
                cnt = torch.tensor([float(n)], dtype=torch.float64, device=rows_a.device)
                parallel.allreduce_sum_(cnt, group)
                self._n_global[n] = float(cnt.item())
        nat.rms_merge(sa, self._n_global[na], d, self.running_mean, self.running_var, self.count)
        self.normalize_into(rows_a, out_a, update=False)
        nat.rms_merge(sb, self._n_global[nb], d, self.running_mean, self.running_var, self.count)
        self.normalize_into(rows_b, out_b, update=False)

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
        native.get(rows.device).rms_normalize_ex(rows, n, d, rows.stride(0), self.running_mean, self.running_var,
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
        if mlp_precision not in MLP_PRECISIONS:
            raise ValueError(f"mlp_precision must be one of {sorted(MLP_PRECISIONS)}, got {mlp_precision!r}")
        self.mlp_precision = mlp_precision
        obs_shape = envs.unwrapped.single_observation_space["policy"].shape
        act_shape = envs.unwrapped.single_action_space.shape
        self.obs_dim, self.act_dim = int(np.array(obs_shape).prod()), int(np.prod(act_shape))
        self.hidden = tuple(int(h) for h in hidden)
        if device is None:
            device = torch.device("cuda" if torch.cuda.is_available() else "cpu")
        device = torch.device(device)
        self.shape = native.shape_of(self.obs_dim, self.act_dim, self.hidden, mfma_bf16=MLP_PRECISIONS[mlp_precision])
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


class RolloutSink:
    """Where an env's fused step (``step_into``) delivers this rollout step: the rollout-buffer rows
    rewards[step] / dones[step+1] / true_dones[step+1], and the observation normaliser whose statistics are
    updated with the raw next observation and whose output lands in obs[step+1].  Addresses are computed from base
    pointers (no tensor slicing on the per-step host path)."""

    def __init__(self, trainer: "PPOTrainer"):
        t = self.t = trainer
        self.step = 0
        rms = t.agent.obs_rms
        self._rms = (rms.running_mean.data_ptr(), rms.running_var.data_ptr(), rms.count.data_ptr(),
                     native.f32(rms.epsilon))
        isz = t.rewards.element_size()
        self._row = t.N * isz
        self._p_rew, self._p_done, self._p_td = t.rewards.data_ptr(), t.dones.data_ptr(), t.true_dones.data_ptr()
        self._p_obs, self._obs_row = t.obs.data_ptr(), t.N * t.Dp * 4
        self._dtype = native.F16 if t.plane_dtype == torch.float16 else native.F32
        self._struct = None
        #: process group over which the env all-reduces the observation moment sums of the fused step (None: this
        #: rank's rows are the whole batch).  Independent of ``dist_exact``: like the unfused path, whose
        #: ``obs_rms.dist_group`` is always the world, the normaliser sees the GLOBAL batch in every sharded run.
        self.obs_group = rms.dist_group if (rms.dist_group is not None and parallel.active(rms.dist_group)) else None

    def fill(self, st):
        k = self.step
        if st is not self._struct:               # constants of the run: once per argument block
            self._struct = st
            st.plane_dtype = self._dtype
            st.obs_mean, st.obs_var, st.obs_count, st.obs_eps = self._rms
            # the divisor of the moment sums must match what was summed: all ranks' rows only if they are exchanged
            st.obs_rows_total = self.t.n_envs_global if self.obs_group is not None else float(self.t.N)
            st.obs_out_ld = self.t.Dp
        row = self._row
        st.rewards_t = self._p_rew + k * row
        st.dones_t1 = self._p_done + (k + 1) * row
        st.true_dones_t1 = self._p_td + (k + 1) * row
        st.obs_out = self._p_obs + (k + 1) * self._obs_row


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
        self.rng = str(getattr(c, "rng", "device"))
        if self.rng not in ("device", "torch"):
            raise ValueError(f"rng must be 'device' or 'torch', got {self.rng!r}")
        rd = str(getattr(c, "rollout_dtype", "fp32"))
        if rd not in ("fp32", "fp16"):
            raise ValueError(f"rollout_dtype must be 'fp32' or 'fp16', got {rd!r}")
        self.plane_dtype = torch.float16 if rd == "fp16" else torch.float32
        self.gae_mode = str(getattr(c, "gae_mode", "serial"))
        sched = getattr(c, "lr_schedule", None)
        if sched is None:
            sched = "linear" if c.anneal_lr else "fixed"
        if sched not in ("linear", "fixed", "adaptive"):
            raise ValueError(f"lr_schedule must be None, 'linear', 'fixed' or 'adaptive', got {sched!r}")
        self.lr_schedule = sched
        self.n_envs_global = float(self.N)
        rows_per_rank, me = [self.batch], 0
        if parallel.active():
            parallel.init_native_comm(self.nat)                 # RCCL under the C ABI (catppo_comm_init)
            parallel.broadcast_(a.flat, src=0)                  # replicas start identical (cf. skrl ppo.py:126-131)
            a.obs_rms.dist_group = a.value_rms.dist_group = torch.distributed.group.WORLD
            cm = getattr(envs.unwrapped, "constraint_manager", None)
            if cm is not None and getattr(c, "dist_exact", True):
                cm.dist_group = torch.distributed.group.WORLD
            counts = parallel.gather_counts(self.N, self.device)      # shards may differ by one env
            self.n_envs_global = float(sum(counts))
            rows_per_rank, me = [self.T * n for n in counts], (self.rank if len(counts) > 1 else 0)
            mbs = parallel.gather_counts(self.mb, self.device)
            if len(set(mbs)) != 1:
                raise ValueError(f"minibatch_size differs between ranks: {mbs}")
        # ONE minibatch schedule for all ranks (ragged shards would otherwise disagree on the number of gradient
        # all-reduces), with the true global row count of every minibatch for the 1/M_global loss scaling
        self.n_mb, m_r, per_rank, self._mb_rows_global = parallel.minibatch_plan(rows_per_rank, self.mb)
        self.M, self._mb_rows = m_r[me], per_rank[me]
        cm = getattr(envs.unwrapped, "constraint_manager", None)
        if cm is not None and hasattr(cm, "ensure_log_ring"):
            cm.ensure_log_ring(self.T + 2)      # the per-step episode logs are read after the rollout
        n_flat = a.layout.n_flat
        dev, T, N = self.device, self.T, self.N
        z = lambda *s, **k: torch.zeros(*s, device=dev, **k)
        pz = lambda *s: torch.zeros(*s, device=dev, dtype=self.plane_dtype)
        self.grad, self.exp_avg, self.exp_avg_sq = z(n_flat), z(n_flat), z(n_flat)
        self.adam_step = 0
        # device-resident iteration state: lr, Adam step count, RNG counters (catppo_iter_state)
        self.state = self.nat.iter_state_new(int(getattr(c, "seed", 0)) * 1000003 + 977 * self.rank + 1,
                                             float(c.learning_rate))
        self.kl_buf = z(1)
        # time-major rollout buffers; slot T holds the bootstrap observation / dones (reference keeps them in
        # next_obs / next_done / next_true_done).  The six (T,N) planes GAE touches are fp32 or fp16.
        self.obs = z(T + 1, N, self.Dp)
        self.actions = z(T, N, self.A)
        self.logprobs = z(T, N)
        self.rewards, self.values = pz(T, N), pz(T, N)
        self.dones, self.true_dones = pz(T + 1, N), pz(T + 1, N)
        self.advantages, self.returns = pz(T, N), pz(T, N)
        self.values_n, self.returns_n = z(T, N), z(T, N)
        self.next_value = pz(N)
        self._act_rows = list(self.actions.unbind(0))
        self.noise = z(T, N, self.A) if self.rng == "torch" else None
        self.record_noise = False        # device rng: keep the noise / permutations used (parity tests replay them)
        self.noise_rec = None
        self.perm_rec = None
        self.diag = z(8)
        self.adv_stats = z(2)
        self.hp = native.PpoHparams(float(c.clip_coef), float(c.ent_coef), float(c.vf_coef), int(bool(c.norm_adv)),
                                    int(bool(c.clip_vloss)), 1.0 / self._mb_rows_global[0], 0)
        self.nat.mlp_reserve(a.shape, max(self.M, N))
        self.iteration = 0
        self.global_step = 0
        # fused env step (two launches) when the env offers it
        env_u = envs.unwrapped
        self.sink = None
        if bool(getattr(c, "fused_rollout", True)) and hasattr(env_u, "step_into") and \
                getattr(env_u, "can_step_into", lambda: False)() and os.environ.get("CATPPO_FUSED_ROLLOUT", "1") != "0":
            self.sink = RolloutSink(self)
        dt = getattr(c, "defer_rollout_tail", None)
        if os.environ.get("CATPPO_ROLLOUT_DEFER_TAIL") is not None:
            dt = os.environ["CATPPO_ROLLOUT_DEFER_TAIL"] != "0"
        #: catppo_rollout_defer_tail around the env steps of a rollout (fused path only)
        self.defer_tail = True if dt is None else bool(dt)
        # hipGraph replay of the update phase
        g = getattr(c, "graph_update", None)
        env_g = os.environ.get("CATPPO_GRAPH_UPDATE")
        if env_g is not None:
            g = env_g == "1"
        if g is None:
            # replay pays at every minibatch size: -5 % of an iteration at 2048 rows (launch bound), -1.3 % at 16384
            # (9.57 -> 9.41 ms of update phase at cfg2, round 3: ~330 launches whose host-side enqueue and inter-launch
            # gaps the graph removes); round 2 only enabled it up to 4096 rows
            g = True
        dist_on_torch = parallel.active() and not parallel.native_comm_active()
        # RCCL collectives inside the captured graph (they are stream operations under the C ABI): exercised on a world of
        # one (tests) - but no run with real peers has been possible on the one-GPU development boxes, a hang or an
        # asynchronous fault inside a REPLAYED graph cannot be caught and turned into the eager fallback, and the update
        # phase of an env-sharded run is GPU bound (graph replay buys ~1 % at 16384-row minibatches).  So with real peers
        # it is OPT-IN again (round 5): CATPPO_GRAPH_COMM=1 / cfg ``graph_comm=True`` once a node has shown it
        # works.  When on: the ranks agree (a vote over the rendezvous group) on whether EVERY rank captured and replayed;
        # otherwise all of them drop to eager launches together on a fresh communicator (``graph_fallback``).
        gc = getattr(c, "graph_comm", None)
        if os.environ.get("CATPPO_GRAPH_COMM") is not None:
            gc = os.environ["CATPPO_GRAPH_COMM"] != "0"
        dist_in_graph_ok = self.world == 1 or bool(gc)
        self.graph_update = bool(g) and self.rng == "device" and not dist_on_torch and dist_in_graph_ok
        #: why the update phase left the graph path (None: it did not)
        self.graph_fallback = None
        # gradient all-reduce in per-layer buckets on the library's side stream, under the backward launches of the
        # layers below (catppo_set_grad_overlap; cf. skrl/ppo.py:534-537, which reduces after the whole backward).
        # Needs libcatppo's own communicator.  OPT-IN (cfg ``grad_overlap`` / CATPPO_GRAD_OVERLAP=1): measured on a world of
        # one with every exchange point forced on (profiles/r4_grad_overlap_world1.txt) the two fork / join pairs and the
        # per-bucket fold launches cost +29 us per optimiser step (9.50 -> 10.38 ms of update phase at cfg2) - more than a
        # 1.5 MB ring all-reduce over xGMI is expected to take serialised, so it only pays on a fabric
        # where that all-reduce is slower than ~30 us.
        go = getattr(c, "grad_overlap", None)
        env_go = os.environ.get("CATPPO_GRAD_OVERLAP")
        if env_go is not None:
            go = {"1": True, "2": "tail", "tail": "tail"}.get(env_go, False)
        if go is None:
            go = False
        # (round 5: "tail" / CATPPO_GRAD_OVERLAP=2 = the no-extra-launch form, catppo_set_grad_overlap(ctx, 2): measured
        # in profiles/r5_grad_overlap_tail_world1.txt)
        self.grad_overlap = False
        if parallel.active() and parallel.native_comm_active():
            self.grad_overlap = self.nat.set_grad_overlap(go)
        self._grad_overlap_mode = go
        # single process: an optimiser step is ONE library call (catppo_ppo_minibatch_step_packed) whose fold launches
        # emit the squared gradient norm of the clip - the launch that re-read the gradient for it is gone.  Not with
        # exchange points on (the gradient all-reduce sits between fold and clip) nor with the side-stream experiment.
        ocs = getattr(c, "one_call_step", None)
        if os.environ.get("CATPPO_ONE_CALL_STEP") is not None:
            ocs = os.environ["CATPPO_ONE_CALL_STEP"] == "1"
        self.one_call_step = (True if ocs is None else bool(ocs)) and not parallel.active() and \
            os.environ.get("CATPPO_SIDE_STREAM", "0") != "1"
        self._graph_id = None
        self._eager_updates_left = 1 if parallel.active() else 0
        self.graph_nodes = 0
        self.stream = torch.cuda.Stream(device=dev) if self.graph_update else None
        # first observation (ppo.py:186-189)
        first = envs.reset()[0]["policy"]
        a.obs_rms.normalize_into(self._rows(first), self.obs[0])
        if self.stream is not None:
            self.stream.wait_stream(torch.cuda.current_stream(dev))
        self.time_phases = False
        self._phase_events = []

    def phase_summary(self, reset: bool = True):
        """mean device milliseconds per iteration of the three phases (needs ``time_phases = True``); synchronises"""
        torch.cuda.synchronize()
        n = max(len(self._phase_events), 1)
        out = {"rollout_ms": 0.0, "gae_and_normalisers_ms": 0.0, "update_ms": 0.0, "iterations": len(self._phase_events)}
        for e in self._phase_events:
            out["rollout_ms"] += e[0].elapsed_time(e[1]) / n
            out["gae_and_normalisers_ms"] += e[1].elapsed_time(e[2]) / n
            out["update_ms"] += e[2].elapsed_time(e[3]) / n
        if reset:
            self._phase_events = []
        return out

    @staticmethod
    def _rows(x):
        x = x if x.dtype == torch.float32 else x.float()
        x = x.reshape(x.shape[0], -1)
        return x if x.stride(1) == 1 else x.contiguous()

    # ------------------------------------------------------------------ rollout (ppo.py:201-230)
    def rollout(self, eps_fn=None):
        c, a, nat, T, N = self.cfg, self.agent, self.nat, self.T, self.N
        ep_infos = []
        if eps_fn is None and self.rng == "torch":
            self.noise.normal_()                                 # all N(0,1) draws of the iteration at once
        if self.record_noise and self.noise_rec is None:
            self.noise_rec = torch.zeros(T, N, self.A, device=self.device)
        env_u = self.envs.unwrapped
        # per-step host path: raw addresses instead of tensor slices, the library called directly
        lib, h, shp = nat.lib, nat.h, C.byref(a.shape)
        p_flat, p_state = a.flat.data_ptr(), self.state.data_ptr()
        p_obs, p_act, p_lp, p_val = self.obs.data_ptr(), self.actions.data_ptr(), self.logprobs.data_ptr(), \
            self.values.data_ptr()
        s_obs, s_act, s_lp, s_val = N * self.Dp * 4, N * self.A * 4, N * 4, N * self.values.element_size()
        vdt = native.F16 if self.plane_dtype == torch.float16 else native.F32
        act_rows = self._act_rows
        use_rng = eps_fn is None and self.rng != "torch"
        # the one-workgroup tail of the fused env step (running maxima / normaliser state / episode log: read by the NEXT
        # env step and by the logging below, not by the policy forward) rides in the next step's first launch instead of
        # standing between catppo_rollout_post and the forward (catppo_rollout_defer_tail; CATPPO_ROLLOUT_DEFER_TAIL=0: A/B)
        defer = self.sink is not None and self.defer_tail
        if defer:
            nat.rollout_defer_tail(True)
        try:
            for step in range(T):
                self.global_step += N * self.world
                if use_rng:                                          # Philox noise inside the head kernel
                    rc = lib.catppo_policy_step(
                        h, shp, p_flat, p_obs + step * s_obs, N, None, None, p_state, step,
                        self.noise_rec[step].data_ptr() if self.record_noise else None, p_act + step * s_act,
                        p_lp + step * s_lp, p_val + step * s_val, vdt, nat._stream())
                else:
                    eps = self.noise[step] if eps_fn is None else eps_fn(step)
                    rc = lib.catppo_policy_step(h, shp, p_flat, p_obs + step * s_obs, N, eps.data_ptr(), None, None, 0, None,
                                                p_act + step * s_act, p_lp + step * s_lp, p_val + step * s_val, vdt,
                                                nat._stream())
                if rc:
                    nat._ok(rc)
                if self.sink is not None:
                    self.sink.step = step
                    next_obs, reward, next_done, timeouts, info = env_u.step_into(act_rows[step], self.sink)
                else:
                    next_obs, reward, next_done, timeouts, info = self.envs.step(self.actions[step])
                    if (reward.dtype == torch.float32 and next_done.dtype == torch.float32 and timeouts.dtype == torch.bool
                            and reward.is_contiguous() and next_done.is_contiguous() and timeouts.is_contiguous()):
                        nat.rollout_store_ex(reward, next_done, timeouts, self.rewards[step], self.dones[step + 1],
                                             self.true_dones[step + 1])
                    else:                                            # foreign env: dtype conversions in the copies
                        self.rewards[step].copy_(reward)
                        self.dones[step + 1].copy_(next_done)
                        self.true_dones[step + 1].copy_(timeouts)
                    a.obs_rms.normalize_into(self._rows(next_obs["policy"]), self.obs[step + 1])
                if "episode" in info:
                    ep_infos.append(info["episode"])
                elif "log" in info:
                    packed = info.get("log_packed")
                    if packed is None:
                        ep_infos.append(info["log"])
                    else:                                 # (keys, device tensor) + the host-side scalars of the log
                        host = info.get("log_host")
                        extra = dict(host) if host is not None else \
                            {k: v for k, v in info["log"].items() if not isinstance(v, torch.Tensor)}
                        ep_infos.append((packed[0], packed[1], extra))
                info["true_dones"] = timeouts
                if "time_outs" in info:
                    if info["time_outs"].any():
                        print("time outs", info["time_outs"].sum())
                        exit(0)
        finally:
            if defer:
                nat.rollout_defer_tail(False)          # = flush: everything the steps wrote is current from here on
        return ep_infos

    # ------------------------------------------------------------------ GAE + normalisers (:251-288)
    def compute_returns(self):
        c, a, nat, T, N = self.cfg, self.agent, self.nat, self.T, self.N
        nat.value_ex(a.shape, a.flat, self.obs[T], N, self.next_value)
        args = (self.rewards, self.values, self.dones[:T], self.true_dones[:T], self.next_value, self.dones[T],
                self.true_dones[T], c.gamma, c.gae_lambda, self.advantages, self.returns)
        if self.plane_dtype == torch.float16:
            nat.gae_f16(*args)
        elif self.gae_mode == "scan":
            nat.gae_mode(native.GAE_SCAN, *args)
        else:
            nat.gae(*args)
        # value_rms is updated with the values and then, a second time, with the returns (:287-288); env-sharded runs
        # exchange both batches' moments in one all-reduce
        a.value_rms.update_normalize_pair(self.values.view(-1, 1), self.values_n.view(-1, 1),
                                          self.returns.view(-1, 1), self.returns_n.view(-1, 1))

    # ------------------------------------------------------------------ update (:294-354)
    def _update_buffers(self):
        if not hasattr(self, "_x_g"):
            # packed epoch buffers: one gather launch per epoch, minibatch k = contiguous slice k
            B, M, n_mb = self.batch, self.M, self.n_mb
            self._parts = (M + self.nat.GATHER_ROWS - 1) // self.nat.GATHER_ROWS
            self._x_g = torch.empty(B, self.Dp, device=self.device)
            self._act_g = torch.empty(B, self.A, device=self.device)
            self._scal_g = torch.empty(4 * B, device=self.device)
            self._advp_g = torch.empty(n_mb * self._parts * 2, dtype=torch.float64, device=self.device)
            E = int(self.cfg.updates_epochs)
            self._advp_all = torch.empty(E * n_mb * self._parts * 2, dtype=torch.float64, device=self.device)
            self._adv_mom = torch.zeros(E * n_mb, 3, dtype=torch.float64, device=self.device)
            self._adv_stats_all = torch.zeros(E * n_mb, 2, device=self.device)

    def _update_body(self, perms):
        """the launches of one update phase (E epochs x minibatches).  ``perms``: list of index tensors (injected /
        torch.randperm) or None = keyed on-device permutation.  Contains no host synchronisation, no allocation and
        only library calls (+ RCCL through the C ABI), so it can be captured into a hipGraph."""
        c, a, nat = self.cfg, self.agent, self.nat
        B, M, n_mb = self.batch, self.M, self.n_mb
        b_obs = self.obs[:self.T].view(B, self.Dp)
        b_act = self.actions.view(B, self.A)
        b_logp, b_adv = self.logprobs.view(-1), self.advantages.view(-1)
        b_ret, b_val = self.returns_n.view(-1), self.values_n.view(-1)
        vmean, vvar = a.value_rms.running_mean, a.value_rms.running_var
        E = int(c.updates_epochs)
        exact_adv = parallel.active() and bool(c.norm_adv) and getattr(c, "dist_exact", True)
        self.hp.adv_stats_external = int(exact_adv)
        # keyed on-device permutations: the minibatch advantage moments of ALL epochs are a function of (seed, iteration,
        # epoch) and the advantages alone, so they are formed before the first gather and exchanged in ONE all-reduce per
        # iteration (round 5; one per epoch before).  Injected / torch permutations keep the per-epoch route.
        adv_upfront = exact_adv and perms is None and os.environ.get("CATPPO_ADV_UPFRONT", "1") != "0"
        if adv_upfront:
            nat.adv_moments_keyed(b_adv, self.state, E, B, M, self._advp_all, self._adv_mom)
            parallel.allreduce_sum_(self._adv_mom)
            nat.adv_stats(self._adv_mom, E * n_mb, self._adv_stats_all)
        # trace_params (tests: the branch-flip analysis of tests/test_gpu_parity_sizes.py): the flat parameters after every
        # optimiser step of this update phase, [E * n_mb, n_flat] (device-to-device copies in stream order)
        trace = None
        if getattr(self, "trace_params", False):
            if getattr(self, "param_trace", None) is None or self.param_trace.shape[0] != E * n_mb:
                self.param_trace = torch.empty(E * n_mb, a.layout.n_flat, device=self.device)
            trace = self.param_trace
            self.param_trace_start = a.flat.clone()
            # (and what this update phase reads: the next rollout's first rows overwrite the buffers' step 0 before a test looks)
            self.trace_batch = dict(obs=b_obs.clone(), actions=b_act.clone(), logprobs=b_logp.clone(), values_n=b_val.clone(),
                                    advantages=b_adv.clone(), returns_n=b_ret.clone())
        for epoch in range(E):
            rec = self.perm_rec[epoch] if self.perm_rec is not None else None
            nat.ppo_gather_ex(a.shape, b_obs, b_act, b_logp, b_adv, b_ret, b_val, B, M, self._x_g, self._act_g,
                              self._scal_g, self._advp_g, inds=None if perms is None else perms[epoch],
                              st=self.state, epoch=epoch, inds_out=rec if perms is None else None)
            if exact_adv and not adv_upfront:
                # minibatch advantage mean / unbiased std over ALL ranks (ppo.py:316-318): the moments of every
                # minibatch of the epoch from the chunk sums the gather just wrote (any plane precision, no index
                # array), ONE all-reduce, one finishing launch
                mom = self._adv_mom[epoch * n_mb:(epoch + 1) * n_mb]
                nat.adv_moments_parts(self._advp_g, self._parts, B, M, mom)
                parallel.allreduce_sum_(mom)
                nat.adv_stats(mom, n_mb, self._adv_stats_all[epoch * n_mb:(epoch + 1) * n_mb])
            for k in range(n_mb):
                start, m = k * M, self._mb_rows[k]
                self.hp.inv_global_batch = 1.0 / self._mb_rows_global[k]     # mean over the GLOBAL minibatch
                adv_stats = self._adv_stats_all[epoch * n_mb + k] if exact_adv else None
                if self.one_call_step:
                    # single process: gradient, clip and Adam in one call (the fold launches emit the squared norm)
                    nat.ppo_minibatch_step_packed(a.shape, self.hp, a.flat, self._x_g[start:], self._act_g[start:],
                                                  self._scal_g[4 * start:], self._advp_g[2 * k * self._parts:], m,
                                                  vmean, vvar, adv_stats, self.grad, self.diag, self.exp_avg,
                                                  self.exp_avg_sq, c.max_grad_norm, 0.9, 0.999, 1e-5, self.state)
                    if trace is not None:
                        trace[epoch * n_mb + k].copy_(a.flat)
                    continue
                nat.ppo_minibatch_grad_packed(a.shape, self.hp, a.flat, self._x_g[start:], self._act_g[start:],
                                              self._scal_g[4 * start:], self._advp_g[2 * k * self._parts:], m,
                                              vmean, vvar, adv_stats, self.grad, self.diag)
                if not self.grad_overlap:                       # (else: reduced bucket by bucket inside the call above)
                    parallel.allreduce_sum_(self.grad)          # RCCL SUM of the flat gradient over xGMI
                nat.clip_adam_dev(a.flat, self.grad, self.exp_avg, self.exp_avg_sq, a.layout.n_flat,
                                  c.max_grad_norm, 0.9, 0.999, 1e-5, self.state)
                if trace is not None:
                    trace[epoch * n_mb + k].copy_(a.flat)
            if self.lr_schedule == "adaptive":
                # KL-adaptive learning rate after every epoch (skrl/ppo.py:558-567), entirely on the device
                nat.kl_mean(self.state, self.diag, self.kl_buf)
                parallel.allreduce_sum_(self.kl_buf)            # per-rank values SUM to the global mean KL
                nat.kl_adaptive_lr(self.state, self.kl_buf, float(getattr(c, "kl_threshold", 0.01)))
        return E * n_mb

    def update(self, perm_fn=None):
        c = self.cfg
        B = self.batch
        E = int(c.updates_epochs)
        self._update_buffers()
        self.diag.zero_()
        if self.record_noise and self.perm_rec is None:
            self.perm_rec = torch.zeros(E, B, dtype=torch.int64, device=self.device)
        perms = None
        if perm_fn is not None:
            perms = [perm_fn(e) for e in range(E)]
        elif self.rng == "torch":
            perms = [torch.randperm(B, device=self.device) for e in range(E)]
        tracing = bool(getattr(self, "trace_params", False))      # (a traced update phase runs eagerly)
        if self.graph_update and perms is None and not self.record_noise and not tracing and self._eager_updates_left > 0:
            # env-sharded runs: the FIRST update phase runs eagerly, so that RCCL's first collectives of every size
            # (channel set-up, lazily loaded kernels, possibly allocations) happen outside a stream capture
            self._eager_updates_left -= 1
        elif self.graph_update and perms is None and not self.record_noise and not tracing:
            if self._graph_id is not None:
                try:
                    self.nat.graph_launch(self._graph_id)
                    self.adam_step += self._graph_steps
                    return
                except RuntimeError as e:                        # workspace grew: the graph was dropped
                    self._graph_id = None
                    if parallel.active() and self.world > 1:
                        #  the re-capture below ends in a vote (all_gather_object) that only re-capturing ranks
                        # enter: peers whose replay succeeded are already in the next rollout's collectives, so a rank
                        # that loses its graph alone must not walk into that vote - it fails loudly instead of hanging
                        raise RuntimeError(f"rank {self.rank}: replay of the captured update phase failed ({e}) in an "
                                           f"env-sharded run of {self.world} ranks; re-capturing on one rank would wait in a "
                                           "vote its peers never join") from e
            try:
                self.nat.graph_begin()
                try:
                    self._graph_steps = self._update_body(None)
                except BaseException:
                    # a failure between begin and end leaves a PARTIAL capture: never instantiate or replay it (ending
                    # the capture may itself fail once the capture is invalidated)
                    self.nat.graph_abort()
                    raise
                gid, self.graph_nodes = self.nat.graph_end()
            except RuntimeError as e:
                gid, err = None, e
            else:
                err = None
            if parallel.active() and self.world > 1:
                # every rank learns whether EVERY rank holds a graph before anybody replays: a rank whose capture failed
                # after enqueueing collectives may have left its communicator ahead of its peers'  - then
                # nobody replays, and all ranks continue eagerly on a FRESH communicator
                votes = [None] * self.world
                torch.distributed.all_gather_object(votes, None if err is None else f"{type(err).__name__}: {err}")
                bad = [(i, v) for i, v in enumerate(votes) if v is not None]
                if bad:
                    if gid is not None:
                        self.nat.graph_destroy(gid)
                    gid = None
                    err = err or RuntimeError("capture failed on " + "; ".join(f"rank {i}: {v}" for i, v in bad))
                    parallel.reinit_native_comm()
            if err is None:
                # nothing has executed so far (a capture records, it does not run)
                steps_before = int(self.nat.iter_state_read(self.state).adam_step) if parallel.active() else None
                try:
                    self.nat.graph_launch(gid)
                except RuntimeError as e:
                    err = e
                    # a failing launch normally leaves the phase undone - but a deferred asynchronous error, or a
                    # partially enqueued graph, would make the eager re-run below apply optimiser steps twice on this
                    # rank only.  The device-side Adam step count says which: re-run only if it has not moved.
                    if parallel.active():
                        torch.cuda.synchronize()
                        moved = int(self.nat.iter_state_read(self.state).adam_step) - steps_before
                        self.nat.graph_destroy(gid)
                        if moved != 0:
                            raise RuntimeError(f"hipGraph replay of the update phase failed after {moved} optimiser steps "
                                               f"had executed ({e}); refusing to re-run the phase eagerly") from e
            if err is not None:
                self._graph_id = None
                if not parallel.active():
                    raise err                                    # single process: the error surfaces as before
                # env-sharded run: collectives inside a graph are the one thing a one-GPU box cannot prove.  Fall back
                # to eager launches (same launches, same order, same collectives)
                self.graph_update = False
                self.graph_fallback = f"{type(err).__name__}: {err}"
                import sys
                print(f"[catppo] rank {self.rank}: hipGraph capture / first replay of the update phase failed, "
                      f"falling back to eager launches: {self.graph_fallback}", file=sys.stderr)
            else:
                self._graph_id = gid
                self.adam_step += self._graph_steps
                return
        self.adam_step += self._update_body(perms)

    # ------------------------------------------------------------------ one iteration
    def run_iteration(self, eps_fn=None, perm_fn=None, log: bool = True):
        if self.stream is None:
            return self._run_iteration(eps_fn, perm_fn, log)
        cur = torch.cuda.current_stream(self.device)
        self.stream.wait_stream(cur)
        with torch.cuda.stream(self.stream):                     # graph capture needs a non-default stream
            out = self._run_iteration(eps_fn, perm_fn, log)
        cur.wait_stream(self.stream)
        return out

    def _run_iteration(self, eps_fn=None, perm_fn=None, log: bool = True):
        c = self.cfg
        self.iteration += 1
        it = self.iteration
        self.lr = float(c.learning_rate)
        sched = {"fixed": native.LR_FIXED, "linear": native.LR_LINEAR, "adaptive": native.LR_KEEP}[self.lr_schedule]
        if self.lr_schedule == "linear":
            frac = 1.0 - (it - 1.0) / c.num_iterations
            self.lr = frac * c.learning_rate                     # host mirror of the device-side schedule (logging)
        self.nat.iter_begin(self.state, float(c.learning_rate), int(c.num_iterations), sched)
        ev = None
        if self.time_phases:                                     # HIP events on the stream the launches go to
            ev = [torch.cuda.Event(enable_timing=True) for _ in range(4)]
            ev[0].record()
        ep_infos = self.rollout(eps_fn)
        if ev:
            ev[1].record()
        self.compute_returns()
        if ev:
            ev[2].record()
        self.update(perm_fn)
        if ev:
            ev[3].record()
            self._phase_events.append(ev)
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
        if self.lr_schedule == "adaptive":
            self.lr = float(self.nat.iter_state_read(self.state).lr)
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
