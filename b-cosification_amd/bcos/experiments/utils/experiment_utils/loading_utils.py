"""State-dict readers for the checkpoint containers the reference writes.

Restates the inference-side behaviour of bcos/experiments/utils/experiment_utils/loading_utils.py:
  * PyTorch-Lightning checkpoints {"state_dict": {"model.<key>": ..., "ema.module.<key>": ...}, "epoch",
    "pytorch-lightning_version"} (:78-107): model weights under the "model." prefix, EMA under "ema.module.";
  * simple training checkpoints {"model_state_dict": ...} (:110-133);
  * stripped flat state dicts written by scripts/strip_checkpoints.py:52-84 (plain {key: tensor});
  * directory convention <save_dir>/last.ckpt and epoch=<N>-*.ckpt (:47-75, structure_constants.py:15).
  * reload = "best" / "best_any" (:180-199, 273-321): the epoch with the highest validation accuracy in
    <save_dir>/metrics/eval_acc1[.ema].gz (metric_utils.Metrics), "best_any" switching to the EMA weights when their best
    accuracy is higher.
"""
from pathlib import Path
from typing import Any, Dict, Optional, Tuple, Union

import torch

__all__ = ["ReloadTypes", "EMANotFound", "change_state_dict_keys", "device_safe_load_state_dict_from_path",
           "load_model_state_dict_from_training_ckpt", "get_last_checkpoint_path_in_save_dir",
           "get_state_dict_and_training_ckpt_from_save_dir", "CHECKPOINT_LAST_FILENAME"]

PathLike = Union[str, Path]
StateDictType = Dict[str, Any]
CHECKPOINT_LAST_FILENAME = "last.ckpt"
MODEL_PREFIX, EMA_PREFIX = "model.", "ema.module."


class EMANotFound(KeyError):
    pass


class ReloadTypes:
    BEST, BEST_ANY, LAST, EPOCH = "best", "best_any", "last", "epoch_"

    @classmethod
    def validate(cls, value: str) -> bool:
        return value in (cls.BEST, cls.BEST_ANY, cls.LAST) or value.startswith(cls.EPOCH)


def change_state_dict_keys(state_dict: StateDictType, prefix_filter: str = "", new_prefix: str = "") -> StateDictType:
    """Keep the entries whose key starts with `prefix_filter` and replace that prefix by `new_prefix`."""
    return {new_prefix + k[len(prefix_filter):]: v for k, v in state_dict.items() if k.startswith(prefix_filter)}


def device_safe_load_state_dict_from_path(path: PathLike) -> StateDictType:
    """torch.load onto the CPU irrespective of the device the checkpoint was written from."""
    return torch.load(str(path), map_location="cpu", weights_only=False)


def _is_pl(ckpt) -> bool:
    return isinstance(ckpt, dict) and "state_dict" in ckpt and "epoch" in ckpt and "pytorch-lightning_version" in ckpt


def _is_simple(ckpt) -> bool:
    return isinstance(ckpt, dict) and "model_state_dict" in ckpt


def load_model_state_dict_from_training_ckpt(training_ckpt: StateDictType, ema: bool = False) -> StateDictType:
    """-> {"<key>": tensor} with the keys of BcosifyNetwork.state_dict() ("model.conv1.linear.weight", ...): the PL
    prefix "model." (or "ema.module.") is the LightningModule attribute, the remaining "model." belongs to BcosifyNetwork."""
    if _is_pl(training_ckpt) and _is_simple(training_ckpt):
        raise ValueError("ambiguous checkpoint: both a Lightning and a simple container")
    if _is_pl(training_ckpt):
        sd = training_ckpt["state_dict"]
        prefix = EMA_PREFIX if ema else MODEL_PREFIX
        if ema and not any(k.startswith(prefix) for k in sd):
            raise EMANotFound("EMA state dict not found in training checkpoint!")
        return change_state_dict_keys(sd, prefix_filter=prefix)
    if _is_simple(training_ckpt):
        return training_ckpt["model_state_dict"]
    if isinstance(training_ckpt, dict) and training_ckpt and all(torch.is_tensor(v) for v in training_ckpt.values()):
        return training_ckpt                      # stripped checkpoint: already the flat model state dict
    raise NotImplementedError("Unsupported checkpoint format!")


def get_last_checkpoint_path_in_save_dir(save_dir: PathLike) -> Path:
    path = Path(save_dir) / CHECKPOINT_LAST_FILENAME
    if not path.exists():
        raise FileNotFoundError(f"Could not find last checkpoint in {save_dir}!")
    return path


def get_state_dict_and_training_ckpt_from_save_dir(save_dir: PathLike, reload: str = "last", ema: bool = False,
                                                   verbose: bool = False) -> Tuple[StateDictType, StateDictType]:
    save_dir = Path(save_dir)
    if not save_dir.exists():
        raise FileNotFoundError(f"Directory '{save_dir}' does not exist!")
    if not save_dir.is_dir():
        raise ValueError(f"'{save_dir}' is not a directory!")
    if not ReloadTypes.validate(reload):
        raise ValueError(f"Unknown reload type: '{reload}'")
    if reload == ReloadTypes.LAST:
        ckpt = device_safe_load_state_dict_from_path(get_last_checkpoint_path_in_save_dir(save_dir))
    else:
        if reload in (ReloadTypes.BEST, ReloadTypes.BEST_ANY):
            epoch, ema = _determine_best_epoch_and_ema_status(save_dir, ema, reload)
        else:
            epoch = int(reload.split("_")[1])
        ckpt = load_training_checkpoint_for_epoch_in(save_dir, epoch)
    sd = load_model_state_dict_from_training_ckpt(ckpt, ema=ema)
    if verbose and isinstance(ckpt, dict) and "epoch" in ckpt:
        print(f"Loaded epoch: {ckpt['epoch']}" + (" (EMA)" if ema else ""))
    return sd, ckpt


def load_training_checkpoint_for_epoch_in(save_dir: PathLike, epoch: int) -> StateDictType:
    """<save_dir>/epoch=<N>-*.ckpt, the per-epoch files of the trainer's checkpoint callback."""
    try:
        path = next(Path(save_dir).glob(f"epoch={epoch}-*.ckpt"))
    except StopIteration:
        raise FileNotFoundError(f"Tried loading checkpoint for epoch {epoch} but none was found in {save_dir}!")
    return device_safe_load_state_dict_from_path(path)


def _determine_best_epoch_and_ema_status(save_dir: PathLike, ema: bool, reload: str) -> Tuple[int, bool]:
    """"best": the best epoch of the EMA weights if `ema` else of the plain weights; "best_any": whichever of the two
    reached the higher validation accuracy (plain weights on a tie or when no EMA metrics exist)."""
    from .metric_utils import Metrics, MetricsNotFoundError
    try:
        metrics = Metrics.from_experiment_dir(save_dir)
    except MetricsNotFoundError:
        raise MetricsNotFoundError("Unable to find metrics! These are required to find the best checkpoint!")
    if reload == ReloadTypes.BEST:
        epoch, _ = metrics.get_best_epoch_and_accuracy_ema() if ema else metrics.get_best_epoch_and_accuracy()
        return epoch, ema
    epoch, acc = metrics.get_best_epoch_and_accuracy()
    try:
        epoch_ema, acc_ema = metrics.get_best_epoch_and_accuracy_ema()
    except EMANotFound:
        epoch_ema, acc_ema = -1, -1.0
    return (epoch_ema, True) if acc_ema > acc else (epoch, False)
