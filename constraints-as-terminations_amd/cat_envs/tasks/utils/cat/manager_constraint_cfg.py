"""Configuration term of the constraint manager.

Same surface as the reference ``ConstraintTermCfg`` (cat/manager_constraint_cfg.py:23-27):
``func`` (callable ``f(env, **params) -> (N,) | (N,C)`` tensor, positive = violated),
``params`` (inherited) and ``max_p`` (maximum termination probability of the term).
"""
from __future__ import annotations

from collections.abc import Callable

import torch

from cat_envs.shim import MISSING, ManagerTermBaseCfg, configclass


@configclass
class ConstraintTermCfg(ManagerTermBaseCfg):
    func: Callable[..., torch.Tensor] = MISSING
    max_p: float = MISSING
