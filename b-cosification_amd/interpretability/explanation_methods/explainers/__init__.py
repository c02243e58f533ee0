"""Explainer registry (reference interpretability/explanation_methods/explainers/__init__.py:62-104)."""
import warnings

from interpretability.explanation_methods.explainers.captum import Grad, IxG
from interpretability.explanation_methods.explainers.ours import Ours, OursRelative
from interpretability.explanation_methods.explanation_configs import explainer_configs

explainer_map = {"IxG": IxG, "Grad": Grad, "Ours": Ours, "OursRelative": OursRelative}
"""Mapping from explainer name to explainer class."""

OUT_OF_SCOPE = ("Occlusion", "RISE", "LIME", "GCam", "IntGrad", "GB", "DeepLIFT")


def get_explainer(model, explainer_name, config_name, **config_overrides):
    try:
        explainer_config = explainer_configs[explainer_name][config_name]
        updated = {**explainer_config, **config_overrides}
        try:
            return explainer_map[explainer_name](model, **updated)
        except TypeError:
            warnings.warn(f"Ignoring overrides {config_overrides} for explainer config!")
            return explainer_map[explainer_name](model, **explainer_config)
    except KeyError:
        hint = " (a comparison baseline that is out of scope of the MI355X hot path)" if explainer_name in OUT_OF_SCOPE else ""
        raise KeyError(f"Explainer '{explainer_name}' with config '{config_name}' not found{hint}!")
