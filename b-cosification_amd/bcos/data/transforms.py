"""AddInverse input encoding (reference bcos/data/transforms.py:42-55): [r,g,b] -> [r,g,b,1-r,1-g,1-b]."""
import torch
from torch import Tensor, nn

__all__ = ["AddInverse"]


class AddInverse(nn.Module):
    """Concatenates 1 - x along the channel dimension `dim` (default -3).  Pure data formatting; in the fused
    engine this is folded into the input-preparation kernel (bcos_prep_input, add_inverse=1)."""

    def __init__(self, dim: int = -3):
        super().__init__()
        self.dim = dim

    def forward(self, in_tensor: Tensor) -> Tensor:
        return torch.cat([in_tensor, 1 - in_tensor], dim=self.dim)
