from .ppo import PPO, Agent, PPOTrainer, RunningMeanStd  # noqa: F401
from .rl_cfg import CleanRlPpoActorCriticCfg  # noqa: F401
