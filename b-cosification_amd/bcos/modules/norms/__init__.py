from . import uncentered_norms  # noqa: F401
from .uncentered_norms import *  # noqa: F401,F403
from .utils import *  # noqa: F401,F403
