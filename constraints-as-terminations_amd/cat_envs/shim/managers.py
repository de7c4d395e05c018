"""ManagerBase / term cfg / SceneEntityCfg fallbacks (isaaclab's are used when importable)."""
from __future__ import annotations

import inspect
import re
from typing import Any, Callable, Sequence

from .configclass import MISSING, configclass

try:  # pragma: no cover
    from isaaclab.managers import SceneEntityCfg  # type: ignore
    from isaaclab.managers.manager_base import ManagerBase, ManagerTermBase  # type: ignore
    from isaaclab.managers.manager_term_cfg import CurriculumTermCfg, ManagerTermBaseCfg  # type: ignore
    HAVE_ISAACLAB = True
except Exception:
    HAVE_ISAACLAB = False

    @configclass
    class ManagerTermBaseCfg:
        func: Callable | Any = MISSING
        params: dict = {}

    @configclass
    class CurriculumTermCfg(ManagerTermBaseCfg):
        pass

    class SceneEntityCfg:
        """Names are resolved to ids against ``scene[name].joint_names / body_names``."""

        def __init__(self, name: str, joint_names=None, joint_ids=slice(None), body_names=None,
                     body_ids=slice(None), preserve_order: bool = False):
            self.name, self.joint_names, self.body_names = name, joint_names, body_names
            self.joint_ids, self.body_ids, self.preserve_order = joint_ids, body_ids, preserve_order

        @staticmethod
        def _match(patterns, names, preserve_order):
            if isinstance(patterns, str):
                patterns = [patterns]
            if preserve_order:
                ids = [i for p in patterns for i, n in enumerate(names) if re.fullmatch(p, n)]
            else:
                ids = [i for i, n in enumerate(names) if any(re.fullmatch(p, n) for p in patterns)]
            if not ids:
                raise ValueError(f"no name matches {patterns} in {names}")
            return ids

        def resolve(self, scene) -> None:
            entity = scene[self.name]
            if self.joint_names is not None:
                ids = self._match(self.joint_names, entity.joint_names, self.preserve_order)
                self.joint_ids = slice(None) if ids == list(range(len(entity.joint_names))) else ids
            if self.body_names is not None:
                ids = self._match(self.body_names, entity.body_names, self.preserve_order)
                self.body_ids = slice(None) if ids == list(range(len(entity.body_names))) else ids

        def __repr__(self):
            return (f"SceneEntityCfg({self.name!r}, joint_names={self.joint_names}, body_names={self.body_names})")

    class ManagerTermBase:
        def __init__(self, cfg, env):
            self.cfg, self._env = cfg, env

        @property
        def num_envs(self):
            return self._env.num_envs

        @property
        def device(self):
            return self._env.device

        def reset(self, env_ids: Sequence[int] | None = None) -> None:
            pass

        def __call__(self, *args, **kwargs):
            raise NotImplementedError

    class ManagerBase:
        def __init__(self, cfg, env):
            self.cfg = cfg
            self._env = env
            self._prepare_terms()

        @property
        def num_envs(self) -> int:
            return self._env.num_envs

        @property
        def device(self):
            return self._env.device

        def _prepare_terms(self):
            raise NotImplementedError

        def _resolve_common_term_cfg(self, term_name: str, term_cfg, min_argc: int = 1):
            """resolve SceneEntityCfg params against the scene, instantiate class-based terms and
            check that every non-default argument of the term function is supplied."""
            for value in term_cfg.params.values():
                if isinstance(value, SceneEntityCfg) and hasattr(self._env, "scene"):
                    value.resolve(self._env.scene)
            func = term_cfg.func
            if inspect.isclass(func) and issubclass(func, ManagerTermBase):
                term_cfg.func = func = func(cfg=term_cfg, env=self._env)
            if not callable(func):
                raise AttributeError(f"The term '{term_name}' is not callable. Received: {func}")
            target = func.__call__ if isinstance(func, ManagerTermBase) else func
            try:
                sig = inspect.signature(target)
            except (TypeError, ValueError):
                return
            names = [p.name for p in sig.parameters.values()
                     if p.kind in (p.POSITIONAL_ONLY, p.POSITIONAL_OR_KEYWORD, p.KEYWORD_ONLY)
                     and p.default is inspect.Parameter.empty]
            missing = [n for n in names[min_argc:] if n not in term_cfg.params]
            if missing:
                raise ValueError(f"The term '{term_name}' expects mandatory parameters: {missing}"
                                 f" and received: {list(term_cfg.params)}.")
