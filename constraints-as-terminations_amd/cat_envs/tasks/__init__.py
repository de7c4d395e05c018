"""Task packages.  Importing this module registers the Solo12 CaT tasks (reference:
exts/cat_envs/cat_envs/tasks/__init__.py imports every task package for its gym.register side effect)."""
from .locomotion.velocity.config import solo12  # noqa: F401
