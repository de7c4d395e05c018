"""Task registration (reference: solo12/__init__.py:16-39): same task ids and entry-point keys.
gymnasium's registry is used when it is installed, the package-local one otherwise."""
from cat_envs.shim import register
from cat_envs.tasks.utils.cat.cat_env import CaTEnv

from . import agents

_KW = {
    "clean_rl_cfg_entry_point": f"{agents.__name__}.clean_rl_ppo_cfg:Solo12FlatPPORunnerCfg",
}

register(
    id="Isaac-Velocity-CaT-Flat-Solo12-v0",
    entry_point=CaTEnv,
    disable_env_checker=True,
    kwargs={"env_cfg_entry_point": f"{__name__}.cat_flat_env_cfg:Solo12FlatEnvCfg", **_KW},
)

register(
    id="Isaac-Velocity-CaT-Flat-Solo12-Play-v0",
    entry_point=CaTEnv,
    disable_env_checker=True,
    kwargs={"env_cfg_entry_point": f"{__name__}.cat_flat_env_cfg:Solo12FlatEnvCfg_PLAY", **_KW},
)
