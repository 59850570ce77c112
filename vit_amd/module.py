"""`ViTLModule` / `BaseLightningModule`: the module surface the reference's trainer drives (src/vit.py:58-215,
src/basemodule.py:143-196), over the MI355X `MyViT`.

What is kept from the reference is the SURFACE: method names (`forward`, `training_step`, `validation_step`, `test_step`,
`configure_optimizers`, the `on_*` hooks), attributes (`model`, `config`, `loss_name`, `monitor_metric`, `sweep`,
`noise_level`, `task_type`, `val_dict` / `test_dict`), the batch contracts (3-tuple `(flux, error, labels)`; 4-tuple
`(noisy, flux, error, labels)` from the eval datasets, spec_datasets.py:28-34) and the logged names
(`{loss_name}_loss`, `{val|test}_{loss_name}_loss`, `*_mae`, `*_mse`, `*_r2`, `*_acc`, `val_bias_median`, `val_p90`,
`val_beta`).  The bodies are this build's own:
  * one model evaluation per eval batch (the reference runs the model a second time in validation_step / test_step only to
    fetch predictions, vit.py:133-148, 204-212 -- same values in eval mode);
  * epoch statistics (median bias, 90th percentile of |residual|, slope of the pred-vs-label line) in torch on the
    collected predictions;
  * metrics from vit_amd.metrics (torchmetrics is not in this image), whose DDP reduction sums states, not values;
  * training noise through the `vit_add_noise` kernel.
With `lightning` installed the classes subclass `lightning.LightningModule` and run under `L.Trainer`; otherwise a small
stand-in base provides `log` / `trainer` / `logger` for vit_amd.trainer.
"""
from __future__ import annotations

import warnings
from typing import Any, Dict, Optional, Tuple

import torch
import torch.nn as nn

from .builder import get_model
from .metrics import Accuracy, MeanAbsoluteError, MeanSquaredError, R2Score
from .optimizer import OptModule

try:  # pragma: no cover - not installed in the build image
    import lightning as L

    _Base = L.LightningModule
    HAVE_LIGHTNING = True
except Exception:  # noqa: BLE001
    HAVE_LIGHTNING = False

    class _Base(nn.Module):
        """The few LightningModule attributes the path touches."""

        def __init__(self):
            super().__init__()
            self.trainer = None
            self.logger = None
            self._logged: Dict[str, Any] = {}
            self.current_epoch = 0

        def log(self, name, value, on_step=None, on_epoch=None, prog_bar=False, **kw):
            if self.trainer is not None and hasattr(self.trainer, "_log"):
                self.trainer._log(name, value, on_step=on_step, on_epoch=on_epoch)
            else:
                self._logged[name] = value

        def save_hyperparameters(self, *a, **k):
            pass


_CLS_ALIASES = frozenset({"classification", "cls", "class"})


def _normalize_task(config) -> str:
    """'cls' or 'reg' from model.task_type (legacy key model.task; default 'cls'): vit.py:20-27."""
    model_cfg = config.get("model") or {}
    raw = model_cfg.get("task_type") or model_cfg.get("task") or "cls"
    return "cls" if str(raw).lower() in _CLS_ALIASES else "reg"


class BaseLightningModule(_Base):
    def __init__(self, model=None, config={}):
        super().__init__()
        self.model = model
        self.loss_name = model.loss_name
        self.callbacks = []
        self.config = config
        self.sweep = False

    def configure_optimizers(self):
        """Optimizer (+ scheduler) from the `opt` section (basemodule.py:152-182).  Two adjustments are made to that
        section first: a plateau scheduler needs a validation metric, so without `data.val_path` it is dropped; a
        one-cycle scheduler needs the run length, derived from data.num_samples / train.batch_size / train.ep."""
        cfg = self.config
        opt = dict(cfg.get("opt") or {})
        opt["monitor_metric"] = getattr(self, "monitor_metric", self.loss_name)
        schedule = str(opt.get("lr_sch") or "").lower()
        data = cfg.get("data") or {}
        if "plateau" in schedule and not data.get("val_path"):
            warnings.warn("opt.lr_sch is a plateau scheduler but data.val_path is not set: there is no validation metric "
                          "to monitor, so the run continues WITHOUT a learning-rate scheduler", stacklevel=2)
            del opt["lr_sch"]
        if "onecycle" in schedule:
            train = cfg.get("train") or {}
            per_step = int(train.get("batch_size", 64))
            opt["steps_per_epoch"] = -(-int(data.get("num_samples", 32000)) // per_step)
            opt["epochs"] = int(train.get("ep", 100))
        return OptModule.from_config(opt)(self.model)


class ViTLModule(BaseLightningModule):
    def __init__(self, model=None, config={}):
        super().__init__(model=model or self.get_model(config), config=config)
        self.save_hyperparameters(ignore=["model"])
        self.task_type = _normalize_task(config)
        self.noise_level = (config.get("noise") or {}).get("noise_level", 0.0)
        if self.task_type == "cls":
            self.accuracy = Accuracy()
            self.monitor_metric = "acc"
            self._metrics = {"acc": self.accuracy}
        else:
            self.mae, self.mse, self.r2 = MeanAbsoluteError(), MeanSquaredError(), R2Score()
            self.monitor_metric = "mae"
            self._metrics = {"mae": self.mae, "mse": self.mse, "r2": self.r2}

    def get_model(self, config):
        return get_model(config)

    def epoch_metrics(self) -> Dict[str, Any]:
        """name -> running metric object (vit_amd.trainer resets / syncs / computes them per evaluation epoch)."""
        return self._metrics

    # ------------------------------------------------------------------ steps
    def forward(self, flux, labels, loss_only=True):
        outputs = self.model(flux, labels=labels)
        return outputs.loss if loss_only else outputs

    def _augment(self, flux: torch.Tensor, error: torch.Tensor) -> torch.Tensor:
        """Training-time noise (vit.py:86-88): flux + N(0,1) * error * noise_level, drawn on the device by `vit_add_noise`.
        The per-step seed comes from torch's seeded CPU generator, so seed_everything(42) fixes the whole run."""
        if not self.noise_level or self.noise_level <= 0:
            return flux
        from . import functional as vf

        if error is None:
            raise ValueError("noise.noise_level > 0 needs the batch's `error` tensor; the loader left it on the host because the "
                             "dataset was built with noise_level 0 (SpecLoader(ship_error=True) ships it regardless)")
        seed = int(torch.randint(0, 2 ** 62, (1,)).item())
        return vf.add_noise(flux.contiguous().float(), error.contiguous().float(), float(self.noise_level), seed)

    def training_step(self, batch, batch_idx):
        flux, error, labels = batch
        loss = self(self._augment(flux, error), labels, loss_only=True)
        self.log(f"{self.loss_name}_loss", loss, on_step=True, on_epoch=True, prog_bar=True)
        return loss

    def _eval_inputs(self, batch) -> Tuple[torch.Tensor, torch.Tensor]:
        """(model input, labels) of an eval batch.  Eval datasets carry a fixed-seed noisy copy first (4-tuple); it is
        used only when the run has noise switched on."""
        labels = batch[-1]
        if len(batch) == 4:
            return (batch[0] if self.noise_level > 0 else batch[1]), labels
        if len(batch) == 3:
            return batch[0], labels
        raise ValueError(f"eval batch must be (flux, error, labels) or (noisy, flux, error, labels); got {len(batch)} items")

    def _shared_eval_step(self, batch, prefix):
        x, labels = self._eval_inputs(batch)
        outputs = self.forward(x, labels, loss_only=False)
        self.log(f"{prefix}_{self.loss_name}_loss", outputs.loss, on_step=False, on_epoch=True)
        scored = outputs.logits if self.task_type == "cls" else outputs.logits.squeeze()
        for name, metric in self._metrics.items():
            self.log(f"{prefix}_{name}", metric(scored, labels), on_step=False, on_epoch=True, prog_bar=(name == "acc"))
        store: Optional[dict] = getattr(self, f"{prefix}_dict", None)
        if self.task_type == "reg" and store is not None:
            store["preds"].append(scored.detach().float().cpu())
            store["labels"].append(labels.detach().float().cpu())
        return outputs.loss

    def validation_step(self, batch, batch_idx):
        return self._shared_eval_step(batch, "val")

    def test_step(self, batch, batch_idx):
        return self._shared_eval_step(batch, "test")

    # ------------------------------------------------------------------ epoch-level regression statistics
    def on_validation_start(self):
        if self.task_type == "reg":
            self.val_dict = {"preds": [], "labels": []}

    def on_test_start(self):
        if self.task_type == "reg":
            self.test_dict = {"preds": [], "labels": []}

    @staticmethod
    def _residual_stats(pred: torch.Tensor, label: torch.Tensor) -> Dict[str, float]:
        """Median of (pred - label), 90th percentile of |pred - label| (linear interpolation between order statistics)
        and the least-squares slope of pred against label, in float64."""
        p, t = pred.double(), label.double()
        res = p - t
        # torch.median returns the lower of the two middle values for an even count; the statistic wanted is their mean
        bias = float(torch.quantile(res, 0.5))
        p90 = float(torch.quantile(res.abs(), 0.9))
        tc = t - t.mean()
        denom = float((tc * tc).sum())
        slope = float((tc * (p - p.mean())).sum() / denom) if denom > 0 else float("nan")
        return {"bias_median": bias, "p90": p90, "beta": slope}

    def on_validation_epoch_end(self):
        """val_bias_median / val_p90 / val_beta per regression target (suffix _i when there are several): vit.py:157-187."""
        store = getattr(self, "val_dict", None)
        if self.task_type != "reg" or not store or not store["preds"]:
            return
        n_rows = sum(int(l.shape[0]) if l.dim() > 0 else 1 for l in store["labels"])
        preds = torch.cat([p.reshape(-1) for p in store["preds"]]).reshape(n_rows, -1)
        labels = torch.cat([l.reshape(-1) for l in store["labels"]]).reshape(n_rows, -1)
        for col in range(preds.shape[1]):
            tag = "" if preds.shape[1] == 1 else f"_{col}"
            for key, value in self._residual_stats(preds[:, col], labels[:, col]).items():
                self.log(f"val_{key}{tag}", value, on_epoch=True)
        self.val_dict = {"preds": [], "labels": []}
