from .batchnorm_uncentered import *  # noqa: F401,F403
