"""`Experiment`: locate a trained B-cosified model by (dataset, base_network, experiment_name) and build + load it.

Inference-side restatement of bcos/experiments/utils/experiment_utils/experiment_utils.py:27-256:
`Experiment("ImageNet", "bcosification", "resnet_50").load_trained_model()` resolves
<base_directory>/ImageNet/bcosification/resnet_50/last.ckpt, looks the model config up in
`bcos.experiments.<dataset>.<base_network>.experiment_parameters.CONFIGS`, builds the network with that package's
`model.get_model` and loads the checkpoint with zero key edits.  Training-side members (trainer, datamodule, metrics)
are not provided.
"""
from pathlib import Path
from typing import Any, Dict, Optional, Union

from ..config_utils import get_configs_and_model_factory, update_config
from .loading_utils import get_state_dict_and_training_ckpt_from_save_dir

__all__ = ["Experiment"]


class Experiment:
    def __init__(self, path_or_dataset: Union[str, Path], base_network: Optional[str] = None,
                 experiment_name: Optional[str] = None, base_directory: Union[str, Path] = "./experiments"):
        if base_network is None and experiment_name is None:            # a path .../<dataset>/<base_network>/<name>
            parts = Path(path_or_dataset).parts
            if len(parts) < 3:
                raise ValueError("Experiment path must end in <dataset>/<base_network>/<experiment_name>")
            base_directory = Path(*parts[:-3]) if len(parts) > 3 else Path(".")
            path_or_dataset, base_network, experiment_name = parts[-3:]
        elif base_network is None or experiment_name is None:
            raise ValueError("give either a path or dataset, base_network and experiment_name")
        self.base_directory = Path(base_directory)
        self.dataset, self.base_network, self.experiment_name = str(path_or_dataset), base_network, experiment_name
        self.save_dir = self.base_directory / self.dataset / self.base_network / self.experiment_name
        configs, self._model_factory = get_configs_and_model_factory(self.dataset, self.base_network)
        if self.experiment_name not in configs:
            raise KeyError(f"Unknown experiment '{self.experiment_name}' for {self.dataset}/{self.base_network}")
        self.config: Dict[str, Any] = configs[self.experiment_name]

    def get_model(self, **kwargs):
        """The network of this experiment; keyword arguments override entries of its `model` section (deep merge)."""
        return self._model_factory(update_config(self.config["model"], kwargs))

    def load_trained_model(self, reload: str = "last", verbose: bool = False, ema: bool = False, return_training_ckpt_if_possible: bool = False):
        model = self.get_model()
        state_dict, ckpt = get_state_dict_and_training_ckpt_from_save_dir(self.save_dir, reload=reload, ema=ema, verbose=verbose)
        model.load_state_dict(state_dict)
        model.eval()
        return (model, ckpt) if return_training_ckpt_if_possible else model
