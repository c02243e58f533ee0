"""Drop-in `bcos` package for MI355X: same import paths as shrebox/B-cosification's `bcos`
(modules, common, models, data.transforms) with every B-cos operator dispatched to the
hand-written HIP kernels in libbcos_hip.so (see bcos_hip/, include/bcos_hip.h).

Unlike the reference's bcos/__init__.py:5-20 nothing heavy is imported eagerly.
"""
from . import common, modules  # noqa: F401
from .common import BcosUtilMixin, explanation_mode, gradient_to_image  # noqa: F401
