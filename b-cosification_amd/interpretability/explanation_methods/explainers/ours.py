"""The model-inherent B-cos explanations (reference interpretability/explanation_methods/explainers/ours.py:8-76)."""
import torch

from interpretability.explanation_methods.utils import ExplainerBase

__all__ = ["Ours", "OursRelative"]


def Ours(model):
    assert hasattr(model, "attribute_selection"), \
        "model requires a 'attribute_selection' attribute for our explanation method!"
    return model


class _MeanSubtracted(torch.nn.Module):
    def __init__(self, model):
        super().__init__()
        self.model = model

    def forward(self, *args, **kwargs):
        out = self.model(*args, **kwargs)
        assert out.dim() == 2, f"model output must be 2D (batch_size, num_classes) but is {out.ndim}D"
        return out - out.mean(dim=1, keepdim=True)


class OursRelative(ExplainerBase):
    """Mean-corrected explanations: Input x Gradient of y - mean_k y."""

    def __init__(self, model):
        assert hasattr(model, "explanation_mode"), \
            "model requires a 'explanation_mode' attribute for our (relative) explanation method!"
        super().__init__(model)
        from interpretability.explanation_methods.explainers.captum import IxG
        self.explainer = IxG(_MeanSubtracted(model))

    def attribute(self, image, target, **kwargs):
        with self.model.explanation_mode():
            return self.explainer.attribute(image, target)

    def attribute_selection(self, image, targets, **kwargs):
        return torch.cat([self.attribute(image, t) for t in targets], dim=0)
