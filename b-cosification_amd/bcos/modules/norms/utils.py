"""NoBias / Unaffine layer-factory wrappers (reference bcos/modules/norms/utils.py:18-88)."""
from functools import wraps

__all__ = ["NoBias", "Unaffine"]


def _wrap(make_layer, suffix, drop_weight):
    @wraps(make_layer)
    def init(*args, **kwargs):
        norm = make_layer(*args, **kwargs)
        assert norm.bias is not None, "It makes no sense to use this wrapper if you set affine=False!"
        norm.bias = None
        if drop_weight:
            norm.weight = None
        base = norm.__class__.__name__
        norm._get_name = lambda: base + suffix
        return norm

    for attr in ("__name__", "__qualname__"):
        if hasattr(make_layer, attr):
            setattr(init, attr, getattr(make_layer, attr) + suffix)
    return init


def NoBias(make_layer):
    """Layer factory whose product has `bias = None`."""
    return _wrap(make_layer, "NoBias", drop_weight=False)


def Unaffine(make_layer):
    """Layer factory whose product has neither weight nor bias."""
    return _wrap(make_layer, "Unaffine", drop_weight=True)
