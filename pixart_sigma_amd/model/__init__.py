from .builder import MODELS, build_model  # noqa: F401
from .nets import *  # noqa: F401,F403
