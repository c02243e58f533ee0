from .experiment_utils import *  # noqa: F401,F403
from .loading_utils import *  # noqa: F401,F403
from .metric_utils import *  # noqa: F401,F403
