"""Curriculum terms (reference: cat/curriculums.py:21-42)."""
from __future__ import annotations

from collections.abc import Sequence


def modify_constraint_p(env, env_ids: Sequence[int], term_name: str, num_steps: int, init_max_p: float):
    """Anneal a constraint's ``max_p``: the expected time-to-termination goes linearly from 20 steps
    to ``1/init_max_p`` steps over ``num_steps`` env steps.  Host-side double arithmetic; the new
    value reaches the GPU as the next ``catppo_cat_step``'s ``term_dp`` kernel argument."""
    progress = min(env.common_step_counter / num_steps, 1.0)
    horizon_start = 20
    horizon_end = 1 / init_max_p
    max_p = 1 / (horizon_start + progress * (horizon_end - horizon_start))
    manager = env.constraint_manager
    term_cfg = manager.get_term_cfg(term_name)
    term_cfg.max_p = max_p
    manager.set_term_cfg(term_name, term_cfg)
    return max_p
