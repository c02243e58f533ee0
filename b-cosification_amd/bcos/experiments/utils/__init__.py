"""Checkpoint / config lookup for B-cosified models: the inference-side subset of the
reference's bcos/experiments/utils -- no trainer, datamodules, metrics files or CLI."""
from .config_utils import *  # noqa: F401,F403
from .experiment_utils import *  # noqa: F401,F403
