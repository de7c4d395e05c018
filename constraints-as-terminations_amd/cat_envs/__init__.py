"""cat_envs - MI355X-native drop-in for the CaT + CleanRL-PPO hot path of
Gepetto/constraints-as-terminations.  Import paths mirror the reference's
``cat_envs.tasks.utils.{cat,cleanrl}`` so that ``scripts/clean_rl/train.py`` reads the same.

All arithmetic of the path runs in ``lib/libcatppo.so`` (hand-written HIP for gfx950) reached
through :mod:`cat_envs.native`; there is no CPU fallback.
"""
__version__ = "0.1.0"
