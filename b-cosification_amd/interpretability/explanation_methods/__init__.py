from .explainers import get_explainer  # noqa: F401
