"""Trainer-side pieces that touch the hot path's modules: the learnable / scheduled exponent B.
The trainer itself (PyTorch-Lightning module, losses, optimisers) is out of scope."""
from .hooks import Hook, forward_hook_fn, setup_b_parameters  # noqa: F401
