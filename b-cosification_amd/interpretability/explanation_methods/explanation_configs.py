"""Named configurations of the explainers that are part of the MI355X hot path (reference
interpretability/explanation_methods/explanation_configs.py:5-30 lists more: the perturbation / baseline explainers
RISE, LIME, Occlusion, IntGrad, GB, DeepLIFT, GCam are comparison methods that only call `model(x)` many times and
are out of scope)."""

explainer_configs = {
    "Ours": {"default": {}},
    "OursRelative": {"default": {}},
    "IxG": {"default": {}},
    "Grad": {"default": {}},
}
