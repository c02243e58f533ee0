"""DetachableModule / BcosSequential -- mirrors bcos/modules/common.py:8-51 of the reference."""
from torch import nn

from bcos.common import BcosUtilMixin

__all__ = ["DetachableModule", "BcosSequential"]


class DetachableModule(nn.Module):
    """Base class of every module whose dynamic weights can be frozen ("explanation mode").

    `detach` is the flag the reference toggles through `explanation_mode` (bcos/common.py:369-377);
    here it additionally tells the fused HIP forward to emit the per-element scale that the
    input-gradient kernel consumes.
    """

    def __init__(self):
        super().__init__()
        self.detach = False

    def set_explanation_mode(self, activate: bool = True) -> None:
        self.detach = activate

    @property
    def is_in_explanation_mode(self) -> bool:
        return self.detach


class BcosSequential(BcosUtilMixin, nn.Sequential):
    """nn.Sequential + the explanation helpers (reference: bcos/modules/common.py:37-51)."""

    def __init__(self, *args):
        super().__init__(*args)

    @classmethod
    def from_standard_module(cls, mod):
        # rebuilt from the values only, i.e. children are renumbered 0..n-1 (CLIP downsample keys)
        return cls(*mod._modules.values())
