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


def apply_overrides(cfgs: dict, overrides):
    """``env.a.b=value`` / ``agent.x=value``: the dotted override syntax hydra resolves in the reference
    (isaaclab_tasks.utils.hydra, used by scripts/clean_rl/train.py:92).  Values are Python literals where they parse
    (``1024``, ``0.5``, ``True``, ``[512,256,128]``), strings otherwise; an unknown root or attribute is an error,
    like hydra's "Could not override" - a typo must not be ignored."""
    import ast
    for ov in overrides:
        key, sep, raw = ov.partition("=")
        if not sep:
            raise ValueError(f"override '{ov}' is not of the form key=value")
        root, *path = key.split(".")
        if root not in cfgs or not path:
            raise KeyError(f"override '{ov}': unknown root '{root}' (known: {sorted(cfgs)})")
        obj = cfgs[root]
        for name in path[:-1]:
            if not hasattr(obj, name):
                raise AttributeError(f"override '{ov}': '{type(obj).__name__}' has no field '{name}'")
            obj = getattr(obj, name)
        if not hasattr(obj, path[-1]):
            raise AttributeError(f"override '{ov}': '{type(obj).__name__}' has no field '{path[-1]}'")
        try:
            value = ast.literal_eval(raw)
        except (ValueError, SyntaxError):
            value = raw
        setattr(obj, path[-1], value)


def hydra_task_config(task_name: str, agent_cfg_entry_point: str):
    """Decorator with the contract of ``isaaclab_tasks.utils.hydra.hydra_task_config`` (reference
    scripts/clean_rl/train.py:92): resolve the task's env and agent configs from the registry, apply the ``key=value``
    overrides left in ``sys.argv[1:]`` (the script moves them there after argparse, train.py:57-61), then call the
    wrapped ``main(env_cfg, agent_cfg, *args, **kwargs)``.  Overrides are therefore applied BEFORE the body of ``main``
    runs, so the non-hydra command-line flags the body applies (--num_envs, --seed, ...) win over them - the
    reference's precedence.  No hydra involved: the same dotted syntax, resolved by ``apply_overrides``."""
    import functools
    import sys

    def decorator(func):
        @functools.wraps(func)
        def wrapper(*args, **kwargs):
            env_cfg = load_cfg_from_registry(task_name, "env_cfg_entry_point")
            agent_cfg = load_cfg_from_registry(task_name, agent_cfg_entry_point)
            # only `key=value` tokens are overrides.  Anything else left on the command line (the VALUE of an unknown
            # `--flag value` pair that parse_known_args passed through, a stray positional) is reported and skipped, as
            # train.py did before the decorator existed - it must not abort the run as a malformed override
            tokens = [a for a in sys.argv[1:] if not a.startswith("-")]
            stray = [a for a in tokens if "=" not in a]
            if stray:
                print(f"[WARN] ignoring command-line tokens that are neither flags nor key=value overrides: {stray}",
                      file=sys.stderr)
            apply_overrides({"env": env_cfg, "agent": agent_cfg}, [a for a in tokens if "=" in a])
            return func(env_cfg, agent_cfg, *args, **kwargs)
        return wrapper
    return decorator
