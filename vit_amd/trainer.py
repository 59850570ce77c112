"""Minimal trainer reproducing the step semantics the reference obtains from `BaseTrainer(L.Trainer)` /
`SpecTrainer` (src/basemodule.py:203-251, src/vit.py:349-435) -- Lightning itself is not part of this image:

  seed (scripts/run.py:27-30) -> for epoch: for batch: zero_grad -> training_step (fwd) -> backward (DDP all-reduce
  of the flat gradient buffer overlapped with it when devices > 1: hardware_utils.py:95) -> clip global grad-norm to
  `train.grad_clip` or 0.5 (basemodule.py:244) -> optimizer.step();  validation every epoch (basemodule.py:249);
  ReduceLROnPlateau on `val_{monitor}` / epoch- or step-interval schedulers (opt/optimizer.py:150-172);
  EarlyStopping(patience 500, vit.py:365,417-424); `fast_dev_run` when `train.debug` (basemodule.py:245).

'train.precision': '32' (default, as in the reference) -> fp32-class kernels (split-bf16 x3 GEMMs, fp32 attention);
'bf16-mixed' -> bf16 MFMA operands with fp32 master weights / residual stream / statistics (the throughput path).
"""
from __future__ import annotations

import math
import time
from typing import Any, Dict, Iterable, Optional

import torch

from . import ddp as ddp_mod
from .optimizer import FusedAdamW

__all__ = ["Trainer", "select_accelerator_and_devices", "get_training_strategy", "seed_everything"]


def select_accelerator_and_devices(num_gpus: Optional[int] = None):
    """hardware_utils.py:44-83 without the CUDA/MPS/nvidia-smi branches: MI355X GPUs or nothing."""
    if torch.cuda.is_available() and torch.cuda.device_count() > 0:
        return "gpu", (num_gpus if num_gpus and num_gpus > 0 else torch.cuda.device_count())
    return "cpu", 1


def get_training_strategy(device_count: int) -> str:
    """hardware_utils.py:86-95"""
    return "ddp" if device_count and device_count > 1 else "auto"


def seed_everything(seed: int = 42):
    import random

    import numpy as np

    random.seed(seed)
    np.random.seed(seed % (2 ** 32))
    torch.manual_seed(seed)
    return seed


def _to_device(batch, device):
    return tuple(t.to(device, non_blocking=True) if torch.is_tensor(t) else t for t in batch)


class Trainer:
    def __init__(self, config: Dict[str, Any], device: Optional[torch.device] = None, verbose: bool = True):
        """`config` is the `train` section (as the reference passes it: vit.py:359)."""
        self.max_epochs = config.get("ep", 10)
        self.gradient_clip_val = config.get("grad_clip", 0.5)
        self.fast_dev_run = bool(config.get("debug", False))
        self.precision = str(config.get("precision", "32"))
        self.patience = int(config.get("patience", 500))
        self.rank, self.local_rank, self.world = ddp_mod.init_distributed()
        self.acc, self.device0 = select_accelerator_and_devices(config.get("gpus"))
        self.strategy = get_training_strategy(self.world)
        self.device = device or torch.device("cuda", self.local_rank)
        self.backend = torch.distributed.get_backend() if self.world > 1 else None
        self.verbose = verbose and self.rank == 0
        self.logged: Dict[str, float] = {}
        self._epoch_acc: Dict[str, list] = {}
        self.global_step = 0
        self.current_epoch = 0
        self.should_stop = False
        self.history = []

    # LightningModule.log lands here
    def _log(self, name, value, on_step=None, on_epoch=None):
        self._epoch_acc.setdefault(name, []).append(value.detach() if torch.is_tensor(value) else value)

    def _flush_epoch_logs(self):
        out = {}
        for k, vals in self._epoch_acc.items():
            ts = [v.float().reshape(()) if torch.is_tensor(v) else torch.tensor(float(v)) for v in vals]
            dev = next((t.device for t in ts if t.is_cuda), torch.device("cpu"))
            out[k] = float(torch.stack([t.to(dev) for t in ts]).mean())
        self._epoch_acc = {}
        return out

    def _setup(self, module):
        module.trainer = self
        module.to(self.device)
        if hasattr(module.model, "set_precision"):
            module.model.set_precision(self.precision)  # basemodule.py:233: precision=str(train.precision or '32')
        conf = module.configure_optimizers()
        if isinstance(conf, dict):
            self.optimizer = conf["optimizer"]
            self.sched_cfg = conf.get("lr_scheduler")
        else:
            self.optimizer, self.sched_cfg = conf, None
        self.optimizers = [self.optimizer]
        eng = module.model.engine
        self.reducer = None
        if self.world > 1:
            eng._ensure_device_state()
            ddp_mod.broadcast_parameters(eng.flat)
            eng._shadow_version = -1
            self.reducer = ddp_mod.GradAllReducer(lambda: eng.grads, eng.layout.buckets())
            eng.grad_ready_cb = self.reducer.bucket_ready
        if isinstance(self.optimizer, FusedAdamW):
            self.optimizer.set_grad_clip(self.gradient_clip_val)
        # PreprocessorFreezeCallback (src/prepca/callbacks.py): warmup.freeze_epochs > 0 freezes the input preprocessor for
        # that many epochs, -1 for good, 0 never.  (As in the reference, the optimizer was built from the parameters that
        # existed at configure time: a preprocessor unfrozen later becomes trainable for autograd but is only stepped if
        # the optimizer is rebuilt.)
        warm = (getattr(module, "config", {}) or {}).get("warmup", {}) or {}
        self.freeze_epochs = int(warm.get("freeze_epochs", 0) or 0)
        self._unfrozen = False
        if self.freeze_epochs != 0 and hasattr(module.model, "set_preprocessor_trainable"):
            module.model.set_preprocessor_trainable(False)

    def training_step(self, module, batch, batch_idx):
        """One optimisation step with the reference's ordering."""
        self.optimizer.zero_grad(set_to_none=True)
        loss = module.training_step(batch, batch_idx)
        loss.backward()
        if self.reducer is not None:
            self.reducer.finish()
            for p in getattr(self.optimizer, "_extras", []):  # a trainable preprocessor lives outside the flat buffer
                if p.grad is not None:
                    torch.distributed.all_reduce(p.grad, op=torch.distributed.ReduceOp.SUM)
                    p.grad.div_(self.world)
        if not isinstance(self.optimizer, FusedAdamW) and self.gradient_clip_val:
            torch.nn.utils.clip_grad_norm_([p for p in module.parameters() if p.grad is not None], self.gradient_clip_val)
        self.optimizer.step()
        if self.sched_cfg and self.sched_cfg.get("interval") == "step":
            self.sched_cfg["scheduler"].step()
        self.global_step += 1
        return loss

    @torch.no_grad()
    def validate(self, module, loader, prefix="val"):
        module.eval()
        for m in ("mae", "mse", "r2", "accuracy"):
            if hasattr(module, m):
                getattr(module, m).reset()
        hook = getattr(module, f"on_{'validation' if prefix == 'val' else 'test'}_start", None)
        if hook:
            hook()
        step = module.validation_step if prefix == "val" else module.test_step
        for i, batch in enumerate(loader):
            step(_to_device(batch, self.device), i)
            if self.fast_dev_run:
                break
        hook = getattr(module, f"on_{'validation' if prefix == 'val' else 'test'}_epoch_end", None)
        if hook:
            hook()
        logs = self._flush_epoch_logs()
        # torchmetrics semantics: epoch value = compute() over the whole epoch, not the mean of batch values
        for name, attr in (("mae", "mae"), ("mse", "mse"), ("r2", "r2"), ("acc", "accuracy")):
            if hasattr(module, attr) and getattr(module, attr).n:
                logs[f"{prefix}_{name}"] = float(getattr(module, attr).compute())
        if self.world > 1:
            keys = sorted(logs)
            t = torch.tensor([logs[k] for k in keys], dtype=torch.float64, device=self.device)
            torch.distributed.all_reduce(t)
            logs = {k: float(v) / self.world for k, v in zip(keys, t)}
        self.logged.update(logs)
        return logs

    def fit(self, module, train_loader: Iterable, val_loader: Optional[Iterable] = None):
        self._setup(module)
        best, bad = None, 0
        monitor = f"val_{getattr(module, 'monitor_metric', 'loss')}"
        mode_max = monitor.endswith("acc")
        epochs = 1 if self.fast_dev_run else self.max_epochs
        for epoch in range(epochs):
            self.current_epoch = module.current_epoch = epoch
            if self.freeze_epochs > 0 and epoch >= self.freeze_epochs and not self._unfrozen and \
                    hasattr(module.model, "set_preprocessor_trainable"):
                module.model.set_preprocessor_trainable(True)  # callbacks.py:31-46
                self._unfrozen = True
                if self.verbose:
                    print(f"[trainer] epoch {epoch}: unfreezing the input preprocessor")
            if hasattr(train_loader, "set_epoch"):
                train_loader.set_epoch(epoch)
            module.train()
            t0 = time.time()
            n = 0
            for i, batch in enumerate(train_loader):
                self.training_step(module, _to_device(batch, self.device), i)
                n += 1
                if self.fast_dev_run:
                    break
            logs = self._flush_epoch_logs()
            if val_loader is not None:
                logs.update(self.validate(module, val_loader, "val"))
            if self.sched_cfg and self.sched_cfg.get("interval", "epoch") != "step":
                sch = self.sched_cfg["scheduler"]
                if self.sched_cfg.get("reduce_on_plateau"):
                    if monitor in logs:  # strict=False: skip silently when the metric is missing
                        sch.step(logs[monitor])
                else:
                    sch.step()
            logs["lr"] = self.optimizer.param_groups[0]["lr"]
            logs["epoch_time_s"] = time.time() - t0
            self.history.append(logs)
            self.logged.update(logs)
            if self.verbose:
                print(f"[epoch {epoch}] " + " ".join(f"{k}={v:.5g}" for k, v in sorted(logs.items())))
            if monitor in logs:  # EarlyStopping(monitor, patience, mode)
                v = logs[monitor]
                if best is None or (v > best if mode_max else v < best):
                    best, bad = v, 0
                else:
                    bad += 1
                    if bad >= self.patience:
                        break
        return self.history

    def test(self, module, loader):
        module.trainer = self
        module.to(self.device)
        return self.validate(module, loader, "test")
