"""Config helpers of the experiment tables (reference bcos/experiments/utils/config_utils.py).

  update_config(old, new)                     recursive dict merge used by every experiment_parameters.py (:38-66)
  create_configs_with_different_seeds(...)    "<name>-seed=<N>" copies with config["seed"] = N (:227-257)
  get_configs_and_model_factory(ds, net)      (CONFIGS, get_model) of bcos.experiments.<ds>.<net> by import path (:140-177)
  sanitize_config(cfg)                        nested dict of primitives for logging (:186-222)
The CLI printer (`configs_cli`) belongs to the training launcher and is not provided.
"""
import copy
from importlib import import_module
from typing import Any, Callable, Dict, List, Tuple, Union

from .structure_constants import (BASE_EXPERIMENTS_DIRECTORY, CONFIGS_MODULE, CONFIGS_VAR_NAME, MODEL_FACTORY_MODULE,
                                  MODEL_FACTORY_VAR_NAME, ROOT)

__all__ = ["update_config", "create_configs_with_different_seeds", "get_configs_and_model_factory", "sanitize_config",
           "ALLOWED_SANITIZED_DATA_TYPES"]

ALLOWED_SANITIZED_DATA_TYPES = (str, int, float, bool, tuple, list, type(None))


def update_config(old_config: Dict, new_config: Dict) -> Dict:
    """A deep copy of `old_config` with `new_config` merged in: sub-dicts are merged key by key, everything else is
    replaced.  Replacing an existing sub-dict by a non-dict is refused (AssertionError, as in the reference)."""
    merged = copy.deepcopy(old_config)
    for key, value in new_config.items():
        if isinstance(merged.get(key), dict):
            assert isinstance(value, dict), "Trying to overwrite a dict with something in a config!"
            merged[key] = update_config(merged[key], value)
        else:
            merged[key] = value
    return merged


def create_configs_with_different_seeds(configs: Dict[str, Dict[str, Any]], seeds: Union[List[int], int]) -> Dict[str, Dict[str, Any]]:
    out = {}
    for seed in ([seeds] if isinstance(seeds, int) else list(seeds)):
        for name, cfg in configs.items():
            cfg = copy.deepcopy(cfg)
            cfg["seed"] = seed
            out[f"{name}-seed={seed}"] = cfg
    return out


def get_configs_and_model_factory(dataset: str, base_network: str) -> Tuple[Dict, Callable]:
    base = ".".join([ROOT, BASE_EXPERIMENTS_DIRECTORY, dataset, base_network])
    modules = {}
    for what in (MODEL_FACTORY_MODULE, CONFIGS_MODULE):
        try:
            modules[what] = import_module(f"{base}.{what}")
        except ModuleNotFoundError:
            print(f"Unable to import '{base}.{what}'")
            raise
    return getattr(modules[CONFIGS_MODULE], CONFIGS_VAR_NAME), getattr(modules[MODEL_FACTORY_MODULE], MODEL_FACTORY_VAR_NAME)


def sanitize_config(config_dict: Dict) -> Dict:
    """Primitives stay, sub-dicts recurse, objects offering `__to_config__()` are expanded, anything else becomes its repr."""
    clean = {}
    for key, value in config_dict.items():
        if isinstance(value, dict):
            value = sanitize_config(value)
        elif not isinstance(value, ALLOWED_SANITIZED_DATA_TYPES):
            value = sanitize_config(value.__to_config__()) if hasattr(value, "__to_config__") else repr(value)
        clean[key] = value
    return clean
