"""`ViTLModule` / `BaseLightningModule`: the module surface the reference's trainer calls (src/vit.py:58-215,
src/basemodule.py:143-196), on top of the MI355X `MyViT`.

Same methods and step semantics: `forward(flux, labels, loss_only=True)`, `training_step` (optional on-the-fly noise:
flux + randn_like(flux) * error * noise_level, vit.py:83-92), `_shared_eval_step` / `validation_step` / `test_step`
with the 3- and 4-tuple batch contracts (spec_datasets.py:28-34), `configure_optimizers` (ReduceLROnPlateau guard and
OneCycle step injection, basemodule.py:152-182), `self.log(...)` names (`{loss_name}_loss`, `val_mae`, `val_mse`,
`val_r2`, `val_acc`, ...), `monitor_metric`.  When `lightning` is installed the classes subclass
`lightning.LightningModule` and run under `L.Trainer` unchanged; otherwise a minimal stand-in base with the same
`log` / `trainer` / `logger` attributes is used by vit_amd.trainer (Lightning is not part of this image).
"""
from __future__ import annotations

from typing import Any, Dict

import torch
import torch.nn as nn

from .builder import get_model
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


def _normalize_task(config):
    """vit.py:20-27"""
    m = (config.get("model", {}) or {})
    task = (m.get("task_type") or m.get("task") or "cls").lower()
    return "cls" if task in ("classification", "cls", "class") else "reg"


class _Mean:
    """Running metric with torchmetrics' update/compute/reset shape (device-side sums, no host sync per step)."""

    def __init__(self):
        self.reset()

    def reset(self):
        self.s = {}
        self.n = 0

    def _add(self, k, v):
        self.s[k] = self.s.get(k, 0) + v


class MeanAbsoluteError(_Mean):
    def __call__(self, preds, target):
        d = (preds.float().reshape(-1) - target.float().reshape(-1)).abs()
        self._add("abs", d.sum())
        self.n += d.numel()
        return d.mean()

    def compute(self):
        return self.s["abs"] / self.n


class MeanSquaredError(_Mean):
    def __call__(self, preds, target):
        d = (preds.float().reshape(-1) - target.float().reshape(-1)) ** 2
        self._add("sq", d.sum())
        self.n += d.numel()
        return d.mean()

    def compute(self):
        return self.s["sq"] / self.n


class R2Score(_Mean):
    def __call__(self, preds, target):
        p, t = preds.float().reshape(-1), target.float().reshape(-1)
        self._add("st", t.sum()); self._add("stt", (t * t).sum()); self._add("res", ((t - p) ** 2).sum())
        self.n += t.numel()
        return self._r2(t.sum(), (t * t).sum(), ((t - p) ** 2).sum(), t.numel())

    @staticmethod
    def _r2(st, stt, res, n):
        tot = stt - st * st / n
        return 1 - res / tot

    def compute(self):
        return self._r2(self.s["st"], self.s["stt"], self.s["res"], self.n)


class Accuracy(_Mean):
    def __call__(self, logits, target):
        c = (logits.argmax(-1) == target).float()
        self._add("c", c.sum())
        self.n += c.numel()
        return c.mean()

    def compute(self):
        return self.s["c"] / self.n


class BaseLightningModule(_Base):
    def __init__(self, model=None, config={}):
        super().__init__()
        self.model = model
        self.loss_name = model.loss_name
        self.callbacks = []
        self.config = config
        self.sweep = False

    def configure_optimizers(self):
        """basemodule.py:152-182"""
        opt_config = {**self.config.get("opt", {})}
        opt_config["monitor_metric"] = getattr(self, "monitor_metric", self.loss_name)
        lr_sch = opt_config.get("lr_sch", "").lower()
        data_config = self.config.get("data", {})
        has_val_path = bool(data_config.get("val_path"))
        if "plateau" in lr_sch and not has_val_path:
            print("[WARNING] ReduceLROnPlateau requires validation data ('data.val_path' in config) but none configured.")
            print("[WARNING] Disabling learning rate scheduler. Consider adding validation data or using a different scheduler.")
            opt_config.pop("lr_sch", None)
        if "onecycle" in lr_sch:
            train_config = self.config.get("train", {})
            batch_size = train_config.get("batch_size", 64)
            num_samples = data_config.get("num_samples", 32000)
            epochs = train_config.get("ep", 100)
            steps_per_epoch = (num_samples + batch_size - 1) // batch_size
            opt_config["steps_per_epoch"] = steps_per_epoch
            opt_config["epochs"] = epochs
            print(f"[OneCycleLR] Calculated steps_per_epoch={steps_per_epoch}, epochs={epochs}")
        return OptModule.from_config(opt_config)(self.model)


class ViTLModule(BaseLightningModule):
    def __init__(self, model=None, config={}):
        model = model or self.get_model(config)
        super().__init__(model=model, config=config)
        self.save_hyperparameters(ignore=["model"])
        self.task_type = _normalize_task(config)
        self.noise_level = (config.get("noise", {}) or {}).get("noise_level", 0.0)
        if self.task_type == "cls":
            self.accuracy = Accuracy()
            self.monitor_metric = "acc"
        else:
            self.mae, self.mse, self.r2 = MeanAbsoluteError(), MeanSquaredError(), R2Score()
            self.monitor_metric = "mae"

    def get_model(self, config):
        return get_model(config)

    def forward(self, flux, labels, loss_only=True):
        outputs = self.model(flux, labels=labels)
        return outputs.loss if loss_only else outputs

    def training_step(self, batch, batch_idx):
        flux, error, labels = batch
        if self.noise_level > 0:
            # vit.py:86-88; on the device through vit_add_noise (no arithmetic of the path runs in torch), one fresh
            # seed per step drawn from torch's seeded CPU generator so that seed_everything(42) fixes the whole run
            from . import functional as vf

            seed = int(torch.randint(0, 2 ** 62, (1,)).item())
            noisy = vf.add_noise(flux.contiguous().float(), error.contiguous().float(), self.noise_level, seed)  # raises off-GPU
            loss = self(noisy, labels, loss_only=True)
        else:
            loss = self(flux, labels, loss_only=True)
        self.log(f"{self.loss_name}_loss", loss, on_step=True, on_epoch=True, prog_bar=True)
        return loss

    def _shared_eval_step(self, batch, prefix):
        if len(batch) == 4:
            noisy, flux, error, labels = batch
            outputs = self.forward(noisy if self.noise_level > 0 else flux, labels, loss_only=False)
        else:
            flux, error, labels = batch
            outputs = self.forward(flux, labels, loss_only=False)
        loss = outputs.loss
        self.log(f"{prefix}_{self.loss_name}_loss", loss, on_step=False, on_epoch=True)
        if self.task_type == "cls":
            acc = self.accuracy(outputs.logits, labels)
            self.log(f"{prefix}_acc", acc, on_step=False, on_epoch=True, prog_bar=True)
        else:
            preds = outputs.logits.squeeze()
            self.log(f"{prefix}_mae", self.mae(preds, labels), on_step=False, on_epoch=True)
            self.log(f"{prefix}_mse", self.mse(preds, labels), on_step=False, on_epoch=True)
            self.log(f"{prefix}_r2", self.r2(preds, labels), on_step=False, on_epoch=True)
        self._last_eval_outputs = outputs
        return loss

    def validation_step(self, batch, batch_idx):
        loss = self._shared_eval_step(batch, "val")
        if self.task_type == "reg" and hasattr(self, "val_dict"):
            # the reference runs the model a second time here to fetch the predictions (vit.py:133-148); the values are
            # identical in eval mode, so the outputs of the first pass are re-used
            self.val_dict["preds"].append(self._last_eval_outputs.logits.squeeze().detach().cpu())
            self.val_dict["labels"].append(batch[-1].detach().cpu())
        return loss

    def on_validation_start(self):
        if self.task_type == "reg":
            self.val_dict = {"preds": [], "labels": []}

    def on_validation_epoch_end(self):
        """Epoch-level bias / p90 / slope (vit.py:157-187)."""
        if self.task_type != "reg" or not hasattr(self, "val_dict") or not self.val_dict["preds"]:
            return
        import numpy as np

        all_preds = torch.cat([p.reshape(p.shape[0], -1) if p.dim() > 0 else p.reshape(1, 1) for p in self.val_dict["preds"]], 0).numpy()
        all_labels = torch.cat([l.reshape(l.shape[0], -1) for l in self.val_dict["labels"]], 0).numpy()
        for i in range(all_preds.shape[1]):
            residuals = all_preds[:, i] - all_labels[:, i]
            coeffs = np.polyfit(all_labels[:, i], all_preds[:, i], 1)
            suffix = "" if all_preds.shape[1] == 1 else f"_{i}"
            self.log(f"val_bias_median{suffix}", float(np.median(residuals)), on_epoch=True)
            self.log(f"val_p90{suffix}", float(np.percentile(np.abs(residuals), 90)), on_epoch=True)
            self.log(f"val_beta{suffix}", float(coeffs[0]), on_epoch=True)
        self.val_dict = {"preds": [], "labels": []}

    def on_test_start(self):
        if self.task_type == "reg":
            self.test_dict = {"preds": [], "labels": []}

    def test_step(self, batch, batch_idx):
        loss = self._shared_eval_step(batch, "test")
        if self.task_type == "reg" and hasattr(self, "test_dict"):
            self.test_dict["preds"].append(self._last_eval_outputs.logits.squeeze().detach().cpu())
            self.test_dict["labels"].append(batch[-1].detach().cpu())
        return loss
