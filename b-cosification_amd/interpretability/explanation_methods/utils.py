"""Explainer base classes (reference interpretability/explanation_methods/utils.py:37-99).

`InputXGradientBase` restates what the reference obtains from captum==0.7.0 `InputXGradient` / `Saliency`
(interpretability/explanation_methods/explainers/captum.py:3-32; captum is a third-party package that is not part of
the reference tree): for a batch x [N,C,H,W] and per-sample targets t, the gradient of sum_n y[n, t_n] w.r.t. x, times
x for Input x Gradient.  When the wrapped model carries a fused engine (bcos_hip.engine.attach) and is in explanation
mode, the whole batch is explained by one fused forward + input-gradient pass instead of autograd over the modules.
"""
import numpy as np
import torch

__all__ = ["ExplainerBase", "InputXGradientBase", "CaptumDerivative"]


class ExplainerBase:
    def __init__(self, model):
        self.model = model

    def attribute(self, img, target, **kwargs):
        raise NotImplementedError("Need attribution method")

    def attribute_selection(self, img, tgts):
        raise NotImplementedError("Need attribution for selection of targets method")


def _as_target_tensor(target, n, device):
    t = torch.as_tensor(target, device=device, dtype=torch.int64).reshape(-1)
    if t.numel() == 1 and n > 1:
        t = t.expand(n)
    if t.numel() != n:
        raise ValueError(f"need one target per sample: got {t.numel()} targets for {n} samples")
    return t.contiguous()


class InputXGradientBase(ExplainerBase):
    multiply_by_inputs = True

    def __init__(self, model, **configs):
        super().__init__(model)
        self.configs = configs

    def _gradient(self, img, target):
        model = self.model
        engine = getattr(model, "_bcos_engine", None)
        in_expl_mode = any(getattr(m, "detach", False) for m in model.modules()) if isinstance(model, torch.nn.Module) else False
        if engine is not None and getattr(engine, "supports_explain", False) and in_expl_mode:
            return engine.explain(img, target)["dynamic_linear_weights"]
        x = img.detach().clone().requires_grad_(True)
        with torch.enable_grad():
            out = model(x)
            (grad,) = torch.autograd.grad(out.gather(1, target.view(-1, 1)).sum(), x)
        return grad

    def attribute(self, img, target, **kwargs):
        t = _as_target_tensor(target, img.shape[0], img.device)
        grad = self._gradient(img, t)
        return img.detach() * grad if self.multiply_by_inputs else grad

    def attribute_selection(self, img, targets):
        """[N, n_targets] targets -> attributions [N * n_targets, C, H, W] (reference utils.py:83-99)."""
        if isinstance(targets, torch.Tensor):
            targets = targets.detach().cpu().numpy()
        targets = np.array(targets, dtype=int).reshape(len(img), -1)
        out = torch.zeros(*targets.shape[:2], *img.shape[1:], dtype=torch.float32, device=img.device)
        for j in range(targets.shape[1]):
            out[:, j] = self.attribute(img, target=targets[:, j].tolist()).detach()
        return out.reshape(-1, *img.shape[1:])


CaptumDerivative = InputXGradientBase   # name used by the reference for the same role
