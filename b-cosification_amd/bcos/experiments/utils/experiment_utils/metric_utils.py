"""Reader of the metrics files a training run leaves next to its checkpoints.

Restates what reload="best" / "best_any" needs from bcos/experiments/utils/experiment_utils/metric_utils.py:17-148:
<save_dir>/metrics/<name>.gz are text tables (numpy.savetxt) of rows [epoch, value]; the validation accuracy lives
under "eval_acc1", its EMA counterpart under "eval_acc1_ema".  The torchmetrics classes of that file belong to the
trainer and are not built.
"""
from pathlib import Path
from typing import Dict, Tuple, Union

import numpy as np

__all__ = ["Metrics", "MetricsNotFoundError"]

PathLike = Union[str, Path]


class MetricsNotFoundError(FileNotFoundError):
    pass


class Metrics(dict):
    VALIDATION_KEY = "eval_acc1"
    EMA_VALIDATION_KEY = "eval_acc1_ema"

    def __init__(self, tables: Dict[str, np.ndarray]):
        super().__init__({name: np.atleast_2d(np.asarray(t, dtype=np.float64)) for name, t in tables.items()})

    @classmethod
    def from_metrics_dir(cls, metrics_dir: PathLike) -> "Metrics":
        metrics_dir = Path(metrics_dir)
        if not metrics_dir.exists():
            raise MetricsNotFoundError(f"Metrics directory '{metrics_dir}' does not exist!")
        return cls({f.stem: np.loadtxt(f) for f in sorted(metrics_dir.glob("*.gz"))})

    @classmethod
    def from_experiment_dir(cls, exp_dir: PathLike) -> "Metrics":
        exp_dir = Path(exp_dir)
        if not exp_dir.exists():
            raise MetricsNotFoundError(f"Experiment directory '{exp_dir}' does not exist!")
        return cls.from_metrics_dir(exp_dir / "metrics")

    def find_best_epoch_and_metric_value_for(self, metric_key: str, mode: str = "max") -> Tuple[int, float]:
        """(epoch, value) of the extremal row; ties go to the first such row, as argmax / argmin do."""
        if mode not in ("max", "min"):
            raise ValueError(f"Unknown mode={mode!r}")
        table = self[metric_key]
        row = table[int(table[:, 1].argmax() if mode == "max" else table[:, 1].argmin())]
        return int(row[0]), float(row[1])

    def get_best_epoch_and_accuracy(self) -> Tuple[int, float]:
        return self.find_best_epoch_and_metric_value_for(self.VALIDATION_KEY)

    def get_best_epoch_and_accuracy_ema(self) -> Tuple[int, float]:
        from .loading_utils import EMANotFound
        if self.EMA_VALIDATION_KEY not in self:
            raise EMANotFound("EMA metrics not found!")
        return self.find_best_epoch_and_metric_value_for(self.EMA_VALIDATION_KEY)
