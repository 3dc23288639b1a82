from .dpm_solver import DPMS  # noqa: F401
from .iddpm import IDDPM  # noqa: F401
from .sa_solver import SASolverSampler  # noqa: F401
