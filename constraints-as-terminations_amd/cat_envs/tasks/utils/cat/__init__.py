from .constraint_manager import CaT, ConstraintManager  # noqa: F401
from .manager_constraint_cfg import ConstraintTermCfg  # noqa: F401
