"""Task registry standing in for gym.register / gym.make / load_cfg_from_registry."""
from __future__ import annotations

import importlib

registry: dict = {}


def register(id: str, entry_point, kwargs: dict | None = None, **_ignored):
    registry[id] = {"entry_point": entry_point, "kwargs": dict(kwargs or {})}


def _resolve(spec):
    if not isinstance(spec, str):
        return spec
    mod, _, attr = spec.partition(":")
    return getattr(importlib.import_module(mod), attr)


def load_cfg_from_registry(task_name: str, entry_point_key: str):
    if task_name not in registry:
        raise KeyError(f"task '{task_name}' is not registered (known: {sorted(registry)})")
    spec = registry[task_name]["kwargs"].get(entry_point_key)
    if spec is None:
        raise ValueError(f"task '{task_name}' has no '{entry_point_key}'")
    cfg_cls = _resolve(spec)
    return cfg_cls() if isinstance(cfg_cls, type) else cfg_cls


def make(id: str, cfg=None, **kwargs):
    if id not in registry:
        raise KeyError(f"task '{id}' is not registered (known: {sorted(registry)})")
    ep = _resolve(registry[id]["entry_point"])
    if cfg is None:
        cfg = load_cfg_from_registry(id, "env_cfg_entry_point")
    return ep(cfg=cfg, **kwargs)
