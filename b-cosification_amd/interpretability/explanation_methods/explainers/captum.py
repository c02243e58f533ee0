"""IxG / Grad explainers with captum's semantics but without captum (reference
interpretability/explanation_methods/explainers/captum.py:29-37): for B-cos models in explanation mode, IxG is the
model-inherent explanation x * W(x)."""
from interpretability.explanation_methods.utils import InputXGradientBase

__all__ = ["IxG", "Grad"]


class IxG(InputXGradientBase):
    multiply_by_inputs = True

    def __init__(self, model):
        super().__init__(model)


class Grad(InputXGradientBase):
    multiply_by_inputs = False

    def __init__(self, model):
        super().__init__(model)
