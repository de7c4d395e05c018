"""Return computation of the reference's rl_games front end on the HIP kernels.

rl_games itself (the A2C agent, its models, schedulers and experience buffer base classes) is a
third-party package the reference only subclasses; it is out of scope here.  What the reference
*changes* in it for CaT - float dones in the experience buffer, ``value_bootstrap`` reward shaping with
time-outs, ``discount_values`` on float dones (rl_games/cat_common.py:35-112, cat_experience.py) - is
provided as device functions with the same argument meaning.
"""
from .cat_common import CaTA2CAgent, bootstrap_time_outs, discount_values  # noqa: F401
from .cat_experience import CaTExperienceBuffer, CaTVectorizedReplayBuffer, swap_and_flatten01  # noqa: F401
from .rl_games import RlGamesVecEnvWrapperCaT  # noqa: F401
