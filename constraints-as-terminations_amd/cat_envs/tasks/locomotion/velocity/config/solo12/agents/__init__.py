from . import clean_rl_ppo_cfg  # noqa: F401
