"""Import harness for the upstream reference (THIS CONTAINER ONLY).

Used exclusively by ``tests/golden/gen_golden.py`` to execute the reference's own
functions as the oracle and freeze their outputs as ``.npz`` vectors.  Nothing
from ``/root/reference`` is copied: the modules are imported in place, with stub
modules standing in for the packages the image lacks (isaaclab, prettytable,
tensorboard).  The GPU box has no ``/root/reference``; nothing outside the
generator imports this file.

Recipe follows SURVEY.md Appendix A.
"""
from __future__ import annotations

import dataclasses
import importlib
import importlib.util
import os
import sys
import types

REF_ROOT = "/root/reference"
UTILS = os.path.join(REF_ROOT, "exts/cat_envs/cat_envs/tasks/utils")

sys.dont_write_bytecode = True  # the reference tree is writable; leave it untouched


def have_reference() -> bool:
    return os.path.isdir(UTILS)


def _install_stubs() -> None:
    if "isaaclab" in sys.modules:
        return

    def mod(name):
        m = types.ModuleType(name)
        sys.modules[name] = m
        return m

    isaaclab = mod("isaaclab")
    managers = mod("isaaclab.managers")
    manager_base = mod("isaaclab.managers.manager_base")
    manager_term_cfg = mod("isaaclab.managers.manager_term_cfg")
    utils = mod("isaaclab.utils")
    prettytable = mod("prettytable")

    class ManagerTermBase:  # class-based terms expose reset(env_ids)
        def __init__(self, cfg=None, env=None):
            self.cfg, self._env = cfg, env

        def reset(self, env_ids=None):
            pass

    class ManagerBase:
        def __init__(self, cfg, env):
            self.cfg = cfg
            self._env = env
            self._prepare_terms()

        @property
        def num_envs(self):
            return self._env.num_envs

        @property
        def device(self):
            return self._env.device

        def _resolve_common_term_cfg(self, term_name, term_cfg, min_argc=1):
            pass

    @dataclasses.dataclass
    class ManagerTermBaseCfg:
        func: object = None
        params: dict = dataclasses.field(default_factory=dict)

    class SceneEntityCfg:
        def __init__(self, name, joint_names=None, body_names=None, joint_ids=slice(None),
                     body_ids=slice(None), preserve_order=False):
            self.name = name
            self.joint_names = joint_names
            self.body_names = body_names
            self.joint_ids = joint_ids
            self.body_ids = body_ids
            self.preserve_order = preserve_order

    def configclass(cls):
        # the reference's ConstraintTermCfg redeclares ``func`` / adds ``max_p`` with MISSING
        # defaults; a permissive dataclass is enough for the oracle runs.
        ann = dict(getattr(cls, "__annotations__", {}))
        for k in ann:
            if not hasattr(cls, k) or getattr(cls, k) is dataclasses.MISSING:
                setattr(cls, k, None)
        return dataclasses.dataclass(cls)

    class PrettyTable:
        def __init__(self):
            self.rows, self.align, self.title, self.field_names = [], {}, "", []

        def add_row(self, r):
            self.rows.append(r)

        def get_string(self):
            return "\n".join(str(r) for r in self.rows)

    manager_base.ManagerBase = ManagerBase
    manager_base.ManagerTermBase = ManagerTermBase
    manager_term_cfg.ManagerTermBaseCfg = ManagerTermBaseCfg
    managers.SceneEntityCfg = SceneEntityCfg
    managers.ManagerTermBase = ManagerTermBase
    managers.manager_base = manager_base
    managers.manager_term_cfg = manager_term_cfg
    utils.configclass = configclass
    isaaclab.managers = managers
    isaaclab.utils = utils
    prettytable.PrettyTable = PrettyTable

    # tensorboard is absent: PPO() lazily imports torch.utils.tensorboard.SummaryWriter
    tb = types.ModuleType("torch.utils.tensorboard")

    class SummaryWriter:
        last = None

        def __init__(self, log_dir=None, **kw):
            self.scalars = []
            SummaryWriter.last = self

        def add_scalar(self, key, value, step):
            self.scalars.append((key, float(value), int(step)))

    tb.SummaryWriter = SummaryWriter
    sys.modules["torch.utils.tensorboard"] = tb


def load_ref_ppo():
    """reference cleanrl/ppo.py as a module (needs no stubs except tensorboard for PPO())."""
    _install_stubs()
    spec = importlib.util.spec_from_file_location("ref_ppo", os.path.join(UTILS, "cleanrl/ppo.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


def load_ref_cat():
    """reference cat/ package: (constraint_manager, manager_constraint_cfg, constraints, curriculums)."""
    _install_stubs()
    pkg = types.ModuleType("ref_cat")
    pkg.__path__ = [os.path.join(UTILS, "cat")]
    sys.modules["ref_cat"] = pkg
    cm = importlib.import_module("ref_cat.constraint_manager")
    cfg = importlib.import_module("ref_cat.manager_constraint_cfg")
    cs = importlib.import_module("ref_cat.constraints")
    cu = importlib.import_module("ref_cat.curriculums")
    return cm, cfg, cs, cu


def tensorboard_stub():
    return sys.modules["torch.utils.tensorboard"].SummaryWriter
