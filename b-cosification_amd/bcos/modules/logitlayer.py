"""LogitLayer: x / T + bias (reference bcos/modules/logitlayer.py:13-36).  In the fused engine
(bcos_hip/engine.py) this is folded into the global-average-pool kernel."""
from typing import Optional

import torch.nn as nn
from torch import Tensor

__all__ = ["LogitLayer"]


class LogitLayer(nn.Module):
    def __init__(self, logit_temperature: Optional[float] = None, logit_bias: Optional[float] = None):
        super().__init__()
        self.logit_bias = logit_bias
        self.logit_temperature = logit_temperature

    def forward(self, in_tensor: Tensor) -> Tensor:
        out = in_tensor
        if self.logit_temperature is not None:
            out = out / self.logit_temperature
        if self.logit_bias is not None:
            out = out + self.logit_bias
        return out

    def extra_repr(self) -> str:
        parts = []
        if self.logit_temperature is not None:
            parts.append(f"logit_temperature={self.logit_temperature}")
        if self.logit_bias is not None:
            parts.append(f"logit_bias={self.logit_bias}")
        return ", ".join(parts)
