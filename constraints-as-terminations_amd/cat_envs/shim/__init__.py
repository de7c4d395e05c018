"""Minimal stand-ins for the IsaacLab symbols the CaT hot path touches.

When IsaacLab is importable its own classes are used; otherwise these fallbacks give the same
surface: ``configclass`` (dataclass tolerant of un-annotated overrides, mutable defaults,
``MISSING`` placeholders; ``to_dict/replace/copy``), ``ManagerBase`` / ``ManagerTermBase`` /
``ManagerTermBaseCfg``, ``SceneEntityCfg`` (regex joint / body name resolution) and a tiny task
registry replacing ``gym.register`` / ``gym.make`` / ``load_cfg_from_registry``.
"""
from .configclass import MISSING, configclass  # noqa: F401
from .managers import (CurriculumTermCfg, ManagerBase, ManagerTermBase, ManagerTermBaseCfg,  # noqa: F401
                       SceneEntityCfg)
from .registry import (apply_overrides, hydra_task_config, load_cfg_from_registry, make, register,  # noqa: F401
                       registry)
