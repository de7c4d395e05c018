"""KL-adaptive learning-rate schedule, device side.

Reference call site: skrl/ppo.py:558-567 - after every learning epoch the mean of the minibatch KL estimates is
all-reduced over the ranks (``all_reduce(kl, SUM) / world_size``) and handed to ``KLAdaptiveLR.step(kl)``.  The
scheduler class itself lives in skrl (>= 1.4.2, not vendored): its published rule (``kl_factor`` 2, ``lr_factor`` 1.5,
``min_lr`` 1e-6, ``max_lr`` 1e-2; threshold 0.01 in skrl_ppo_cfg.yaml:49-51; rl_games' ``lr_schedule: adaptive`` with
``kl_threshold: 0.008``, rl_games_cat_solo.yaml:64-66, is the same rule) is restated here - PARITY UNPINNED against
skrl / rl_games themselves.

The host never sees the KL: ``step`` enqueues ``catppo_kl_mean`` -> [``catppo_allreduce``] -> ``catppo_kl_adaptive_lr``
on the current HIP stream, and the next ``catppo_clip_adam_dev`` reads the new rate from the device-resident
``catppo_iter_state``.  ``PPOTrainer`` uses the same three calls when ``lr_schedule = "adaptive"``.
"""
from __future__ import annotations

import torch

from cat_envs import native, parallel


class KLAdaptiveLR:
    """``KLAdaptiveLR(state, kl_threshold=0.008)`` - ``state`` is the device tensor made by ``Native.iter_state_new``.

    ``step(diag)``: ``diag`` is the 8-float diagnostics vector the minibatch kernels accumulate (``diag[4]`` = sum of
    the per-minibatch approx-KL means, ``diag[7]`` = minibatch count); the schedule consumes what was added since
    its previous call.  ``step_kl(kl)`` takes a ready-made 1-element device tensor instead."""

    def __init__(self, state: torch.Tensor, kl_threshold: float = 0.008, kl_factor: float = 2.0,
                 lr_factor: float = 1.5, min_lr: float = 1e-6, max_lr: float = 1e-2, group=None):
        self.state, self.group = state, group
        self.kl_threshold, self.kl_factor, self.lr_factor = float(kl_threshold), float(kl_factor), float(lr_factor)
        self.min_lr, self.max_lr = float(min_lr), float(max_lr)
        self.nat = native.get(state.device)
        self._kl = torch.zeros(1, device=state.device)

    def step(self, diag: torch.Tensor):
        self.nat.kl_mean(self.state, diag, self._kl)
        self.step_kl(self._kl)

    def step_kl(self, kl: torch.Tensor):
        parallel.allreduce_sum_(kl, self.group)          # per-rank values already carry the 1/world factor
        self.nat.kl_adaptive_lr(self.state, kl, self.kl_threshold, self.kl_factor, self.lr_factor, self.min_lr,
                                self.max_lr)

    def get_last_lr(self):
        """host copy (synchronises; logging only)"""
        return [float(self.nat.iter_state_read(self.state).lr)]
