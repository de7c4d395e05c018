"""``configclass`` fallback (isaaclab.utils.configclass is used when available)."""
from __future__ import annotations

import copy
import dataclasses
from typing import Any


class _Missing:
    """Placeholder for "must be set by the user" (isaaclab uses dataclasses.MISSING as a value)."""
    _inst = None

    def __new__(cls):
        if cls._inst is None:
            cls._inst = super().__new__(cls)
        return cls._inst

    def __repr__(self):
        return "MISSING"

    def __bool__(self):
        return False

    def __deepcopy__(self, memo):
        return self


MISSING = _Missing()


def _is_config_member(name: str, value: Any) -> bool:
    if name.startswith("__"):
        return False
    if isinstance(value, (staticmethod, classmethod, property)):
        return False
    if callable(value) and not isinstance(value, type) and getattr(value, "__qualname__", "").count(".") \
            and type(value).__name__ == "function":
        # plain functions defined in the class body are methods, not fields
        return False
    return True


def _to_dict(obj):
    if dataclasses.is_dataclass(obj) and not isinstance(obj, type):
        return {f.name: _to_dict(getattr(obj, f.name)) for f in dataclasses.fields(obj)}
    if isinstance(obj, dict):
        return {k: _to_dict(v) for k, v in obj.items()}
    if isinstance(obj, (list, tuple)):
        return type(obj)(_to_dict(v) for v in obj)
    if callable(obj) and hasattr(obj, "__module__") and hasattr(obj, "__name__"):
        return f"{obj.__module__}:{obj.__name__}"
    if obj is None or isinstance(obj, (bool, int, float, str)):
        return obj
    if isinstance(obj, slice):
        return f"slice({obj.start},{obj.stop},{obj.step})"
    if hasattr(obj, "__dict__"):      # plain config objects such as SceneEntityCfg
        return {k: _to_dict(v) for k, v in vars(obj).items() if not k.startswith("_")}
    return str(obj)


def configclass(cls=None, **kwargs):
    def wrap(c):
        ann = dict(c.__dict__.get("__annotations__", {}))
        # nested classes (e.g. ObservationsCfg.PolicyCfg) stay class attributes, not fields
        for name, value in list(c.__dict__.items()):
            if not _is_config_member(name, value) or isinstance(value, type):
                continue
            if type(value).__name__ == "function":
                continue
            if name not in ann:
                ann[name] = type(value) if value is not None else Any   # un-annotated override
        c.__annotations__ = ann
        for name in list(ann):
            if name not in c.__dict__:
                continue
            value = c.__dict__[name]
            if value is dataclasses.MISSING:
                setattr(c, name, MISSING)
            elif isinstance(value, dataclasses.Field):
                continue
            elif isinstance(value, (list, dict, set)) or (dataclasses.is_dataclass(value) and not isinstance(value, type)):
                setattr(c, name, dataclasses.field(default_factory=lambda v=value: copy.deepcopy(v)))
        c = dataclasses.dataclass(c, **kwargs)
        c.to_dict = lambda self: _to_dict(self)
        c.replace = lambda self, **kw: dataclasses.replace(self, **kw)
        c.copy = lambda self: copy.deepcopy(self)
        return c
    return wrap if cls is None else wrap(cls)


try:  # prefer the real thing when IsaacLab is installed
    from isaaclab.utils import configclass as _real_configclass  # type: ignore  # noqa: F401
    from dataclasses import MISSING as _DC_MISSING  # noqa: F401
    configclass = _real_configclass  # noqa: F811
    MISSING = _DC_MISSING  # noqa: F811
except Exception:  # pragma: no cover - IsaacLab is absent on AMD boxes
    pass
