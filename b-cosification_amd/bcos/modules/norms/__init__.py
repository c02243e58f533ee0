from . import centered_norms, uncentered_norms  # noqa: F401
from .centered_norms import *  # noqa: F401,F403
from .uncentered_norms import *  # noqa: F401,F403
from .utils import *  # noqa: F401,F403
