"""Return computation of the reference's skrl front end on the HIP kernels.
skrl (agent base class, models, memories, KL-adaptive scheduler) is third-party and out of scope; the
CaT-specific part - ``compute_gae`` with float ``not_dones`` and whole-batch advantage normalisation
(skrl/ppo.py:397-442) - is provided with the same signature, and the KL-adaptive learning-rate step incl. its KL
all-reduce (skrl/ppo.py:558-567) as a device-side schedule."""
from .returns import compute_gae  # noqa: F401
from .schedulers import KLAdaptiveLR  # noqa: F401
