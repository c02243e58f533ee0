"""bcos.modules -- the operator layer (reference bcos/modules/__init__.py), HIP-backed."""
from . import norms  # noqa: F401
from .bcosconv2d import BcosConv2d, BcosConv2dWithScale, NormedConv2d  # noqa: F401
from .bcoslinear import BcosLinear, NormedLinear  # noqa: F401
from .common import BcosSequential, DetachableModule  # noqa: F401
from .logitlayer import LogitLayer  # noqa: F401
from .norms import *  # noqa: F401,F403
from .bcosattnpool import BcosAttentionPool2d  # noqa: F401,E402
