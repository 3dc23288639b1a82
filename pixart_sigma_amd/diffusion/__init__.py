from .dpm_solver import DPMS  # noqa: F401
from .iddpm import IDDPM  # noqa: F401
