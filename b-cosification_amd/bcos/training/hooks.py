"""The exponent B as a trained / scheduled quantity (reference bcos/training/hooks.py:7-35, bcos/training/trainer.py:447-474).

The reference turns `mod.b` of every B-cos layer into an nn.Parameter (start value + 1e-6, so that the `b == 1` early return of
the layers is not taken) and, for the "linear_b" recipe, registers a gradient hook that REPLACES the exponent's gradient by
`-batch_size` until B reaches its end value -- an optimiser step then moves every B by lr * batch_size: a linear ramp in the
number of samples seen -- and by zero afterwards.  The layers of this package differentiate the exponent themselves
(csrc/bcos_train.hip: bgrad of bcos_train_scale_bwd; bcos/modules/_hipfn.py: learnable_b), so the hook has a gradient to replace.
"""
import torch
import torch.nn as nn

__all__ = ["Hook", "forward_hook_fn", "setup_b_parameters"]


class Hook:
    """Gradient hook of one layer's exponent: call signature and effect of the reference's `Hook` (hooks.py:7-24)."""

    def __init__(self, mod, start=1, end=2):
        self.mod, self.start, self.end = mod, start, end

    def __call__(self, grad):
        b = self.mod.b
        if b < self.start:                         # below the ramp: put it back on its first point
            b.data = torch.tensor(float(self.start + 1e-6), device=b.device, dtype=b.dtype)
        if b >= self.end:                          # the ramp is over: the exponent stays where it is
            return torch.zeros_like(grad)
        return torch.full_like(grad, -float(self.mod.batch_size))


def forward_hook_fn(module, inputs, output):
    """Records the batch size seen by a module (the ramp's step length); inputs may be nested once (DenseNet blocks)."""
    first = inputs[0]
    if not torch.is_tensor(first):
        first = first[0]
    module.batch_size = first.size(0)


def setup_b_parameters(model: nn.Module, bcosify_args: dict):
    """trainer.py:447-474: unless `fix_b`, make the exponent of every layer that has one a parameter starting at
    `b_at_start` + 1e-6, record batch sizes on every module, and with `linear_b` attach the ramp hook
    (`b_at_start` -> `b_at_end`).  Returns the list of the new parameters (for the optimiser's param groups)."""
    made = []
    if bcosify_args is None or bcosify_args.get("fix_b", False):
        return made
    start = bcosify_args.get("b_at_start", 1)
    for mod in model.modules():
        mod.register_forward_hook(forward_hook_fn)
        if hasattr(mod, "b"):
            ref = next(iter(mod.parameters()), None)
            dev = ref.device if ref is not None else torch.device("cpu")
            mod.b = nn.Parameter(torch.tensor(float(start) + 1e-6, device=dev), requires_grad=True)
            if bcosify_args.get("linear_b", False):
                mod.b.register_hook(Hook(mod, start=start, end=bcosify_args.get("b_at_end", 2)))
            made.append(mod.b)
    return made
