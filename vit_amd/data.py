"""Host-side dataset contract of the path: what the reference's `RegSpecDataset` /
`ClassSpecDataset` hand to `training_step` / `_shared_eval_step` (src/dataloader/spec_datasets.py:12-110,
src/dataloader/base.py:219-326), over arrays already in memory.

  * flux is clipped at zero on load (base.py:236);
  * regression labels are optionally normalised -- 'minmax' (the baseline's choice), 'standard' / 'zscore' -- with the
    statistics of the TRAINING split re-used for val / test (spec_datasets.py:73-91);
  * classification labels are `log_g > 2.5` (spec_datasets.py:24);
  * val / test splits carry noisy spectra generated ONCE with a fixed seed, `flux + randn * error * noise_level` under
    torch.manual_seed(42) (base.py:312-326), and yield 4-tuples (noisy, flux, error, labels); training yields 3-tuples and
    the module adds fresh noise on the device every step (vit.py:86-88 -> vit_add_noise).

The reference reads HDF5 (`dataset/arrays/{flux,error}/value`, a pandas parameter table; base.py:227-297); h5py is not
part of this image, so `from_hdf5` is a thin optional reader and everything else works from tensors / .npz files.
"""
from __future__ import annotations

from typing import Iterator, Optional, Sequence, Tuple

import torch

__all__ = ["SpecDataset", "SpecLoader", "SpecDataModule"]


class SpecDataset:
    def __init__(self, flux, error, params, *, task: str = "reg", stage: str = "train", label_norm: str = "none",
                 noise_level: float = 0.0, stats: Optional[dict] = None, noise_seed: int = 42, eps: float = 1e-8):
        self.flux = torch.as_tensor(flux, dtype=torch.float32).clip(min=0.0)  # base.py:236
        self.error = torch.as_tensor(error, dtype=torch.float32)
        if self.error.shape != self.flux.shape:
            raise ValueError("flux and error must have the same shape")
        if self.error.dim() == 2 and self.error.shape[1] > 1 and bool(self.error.isnan().any()):  # base.py:210-215, 238-239
            self.error = self.error.clone()
            if bool(self.error[:, 0].isnan().any()):
                self.error[:, 0] = self.error[:, 1]
            if bool(self.error[:, -1].isnan().any()):
                self.error[:, -1] = self.error[:, -2]
        params = torch.as_tensor(params)
        self.task, self.stage, self.label_norm, self.noise_level = task, stage, label_norm, float(noise_level)
        self.stats = dict(stats or {})
        if task == "cls":
            self.labels = (params.float() > 2.5).long()  # spec_datasets.py:24 (log_g threshold)
        elif task == "reg":
            self.labels = params.float()
            self._normalize(eps)
        else:
            raise ValueError(f"Unsupported task_type '{task}'")
        self.noisy = None
        if stage in ("val", "test", "validate") and self.noise_level > 0:  # base.py:312-326
            gen_state = torch.random.get_rng_state()
            torch.manual_seed(noise_seed)
            self.noisy = self.flux + torch.randn_like(self.flux) * self.error * self.noise_level
            torch.random.set_rng_state(gen_state)

    def _normalize(self, eps):
        kind = self.label_norm
        if kind not in ("standard", "zscore", "minmax"):
            return
        is_train = self.stage in (None, "fit", "train")
        st = self.stats
        if kind in ("standard", "zscore"):
            if is_train or "mean" not in st:
                st["mean"] = self.labels.mean(dim=0)
                st["std"] = self.labels.std(dim=0, unbiased=False)
            std = torch.where(st["std"].abs() < eps, torch.ones_like(st["std"]), st["std"])
            self.labels = (self.labels - st["mean"]) / std
        else:
            if is_train or "min" not in st:
                st["min"] = self.labels.min(dim=0).values
                st["max"] = self.labels.max(dim=0).values
            den = st["max"] - st["min"]
            den = torch.where(den.abs() < eps, torch.ones_like(den), den)
            self.labels = (self.labels - st["min"]) / den

    def __len__(self) -> int:
        return self.flux.shape[0]

    def __getitem__(self, idx):
        if self.noisy is not None:
            return self.noisy[idx], self.flux[idx], self.error[idx], self.labels[idx]
        return self.flux[idx], self.error[idx], self.labels[idx]

    @classmethod
    def from_npz(cls, path, param_keys: Sequence[str] = ("log_g",), num_samples: Optional[int] = None, **kw):
        import numpy as np

        raw = np.load(path)  # allow_pickle stays False: arrays only
        for k in ("flux", "error", *param_keys):
            if k not in raw.files:
                raise KeyError(f"Requested array '{k}' not found in {path}: {list(raw.files)}")
        cols = [torch.from_numpy(np.asarray(raw[k][:num_samples])).float() for k in param_keys]
        params = cols[0] if len(cols) == 1 else torch.stack(cols, dim=1)
        return cls(raw["flux"][:num_samples], raw["error"][:num_samples], params, **kw)

    @classmethod
    def from_hdf5(cls, path, num_samples: Optional[int] = None, param_keys: Sequence[str] = ("log_g",), **kw):
        try:
            import h5py
            import pandas as pd
        except ImportError as e:  # pragma: no cover - not available in this image
            raise ImportError("from_hdf5 needs h5py and pandas (the reference's file layout, base.py:227-297)") from e
        with h5py.File(path, "r") as f:
            flux = torch.tensor(f["dataset/arrays/flux/value"][:num_samples])
            error = torch.tensor(f["dataset/arrays/error/value"][:num_samples])
        df = pd.read_hdf(path)[:num_samples]
        cols = [torch.tensor(df[k].values).float() for k in param_keys]
        return cls(flux, error, cols[0] if len(cols) == 1 else torch.stack(cols, dim=1), **kw)


class SpecLoader:
    """Batches with the reference's DataLoader semantics that matter to the step (basemodule.py:76-85): shuffle on the
    training split unless debugging, DistributedSampler-style sharding over ranks, whole-tensor indexing (no workers)."""

    def __init__(self, ds: SpecDataset, batch_size: int, shuffle: bool = False, seed: int = 42, drop_last: bool = False):
        self.ds, self.bs, self.shuffle, self.seed, self.epoch, self.drop_last = ds, batch_size, shuffle, seed, 0, drop_last

    def set_epoch(self, e: int) -> None:
        self.epoch = e

    def __iter__(self) -> Iterator[Tuple[torch.Tensor, ...]]:
        import torch.distributed as dist

        from .ddp import shard_indices

        rank = dist.get_rank() if dist.is_initialized() else 0
        world = dist.get_world_size() if dist.is_initialized() else 1
        idx = shard_indices(len(self.ds), rank, world, self.epoch, self.shuffle, self.seed)
        for i in range(0, len(idx), self.bs):
            j = idx[i:i + self.bs]
            if self.drop_last and len(j) < self.bs:
                break
            yield self.ds[j]


class SpecDataModule:
    """What `ViTDataModule.from_config(config)` + `BaseDataModule.setup / *_dataloader` give the reference's trainer
    (src/vit.py:29-50, src/basemodule.py:38-127), over this module's `SpecDataset` / `SpecLoader`:

      data.file_path  -> training split (first `data.num_samples` rows)            shuffled unless `train.debug`
      data.val_path   -> validation split (first `data.num_test_samples` rows)     batch = min(train.batch_size, len(val))
      data.test_path  -> test split (first `data.num_test_samples` rows)
      data.param ('log_g' | 'a,b' | [a, b]), data.label_norm, noise.noise_level as in the reference's dataset classes; the
      label statistics of the TRAINING split are re-used on val / test (vit.py:42-50); val / test spectra carry their
      fixed-seed noise (base.py:312-326).

    Files: `.npz` with arrays `flux`, `error` and one array per parameter name (this image has no h5py), or the reference's
    HDF5 layout (`dataset/arrays/{flux,error}/value` + a pandas parameter table: base.py:227-297) when h5py imports."""

    def __init__(self, config: dict):
        self.config = config
        data, train = config.get("data", {}) or {}, config.get("train", {}) or {}
        model = config.get("model", {}) or {}
        task = (model.get("task_type") or model.get("task") or "cls").lower()
        self.task = "cls" if task in ("classification", "cls", "class") else "reg"  # vit.py:20-26
        self.paths = {"train": data.get("file_path") or data.get("train_path"), "val": data.get("val_path"),
                      "test": data.get("test_path")}
        self.num_samples = data.get("num_samples")
        self.num_test_samples = data.get("num_test_samples")
        prm = data.get("param")
        if isinstance(prm, str):
            prm = [x.strip() for x in prm.split(",") if x.strip()]
        self.param_keys = list(prm) if prm else []
        if self.task == "reg" and not self.param_keys:
            raise ValueError("Regression requires 'data.param' to be set in the config (string, comma-separated string, or "
                             "list).")  # spec_datasets.py:52-57
        if self.task == "cls" and not self.param_keys:
            self.param_keys = ["log_g"]  # spec_datasets.py:24
        self.label_norm = data.get("label_norm", "none")
        self.noise_level = float((config.get("noise", {}) or {}).get("noise_level", 0.0) or 0.0)
        self.batch_size = int(train.get("batch_size", 64))
        self.debug = bool(train.get("debug", False))
        self.train = self.val = self.test = None

    @classmethod
    def from_config(cls, config: dict) -> "SpecDataModule":
        return cls(config)

    def available(self, stage: str = "train") -> bool:
        import os

        path = self.paths.get(stage)
        return bool(path) and os.path.exists(path)

    def _load(self, stage: str, stats):
        import os

        path = self.paths["train" if stage == "train" else stage]
        if not path or not os.path.exists(path):
            raise FileNotFoundError(f"[{stage}] Data file not found: {path}")  # base.py:224-225
        n = self.num_samples if stage == "train" else self.num_test_samples
        kw = dict(task=self.task, stage=stage, label_norm=self.label_norm, noise_level=self.noise_level, stats=stats)
        print(f"[{stage}] loading data from {path}, num_samples={n}")
        if path.endswith((".h5", ".hdf5", ".hdf")):
            return SpecDataset.from_hdf5(path, num_samples=n, param_keys=self.param_keys, **kw)
        return SpecDataset.from_npz(path, param_keys=self.param_keys, num_samples=n, **kw)

    def setup(self, stage: str = "fit"):
        if stage in ("fit", None):
            self.train = self._load("train", None)
            if self.paths["val"]:
                self.val = self._load("val", self.train.stats)
        elif stage == "test":
            stats = self.train.stats if self.train is not None else None
            if stats is None and self.label_norm in ("standard", "zscore", "minmax") and self.task == "reg":
                # evaluation only: the statistics still come from the training split (the reference's test-only entry builds
                # the training dataset first for the same reason: Experiment.__init__ -> data_module.setup('fit'))
                stats = self._load("train", None).stats
            self.test = self._load("test", stats)
        return self

    def train_dataloader(self) -> "SpecLoader":
        return SpecLoader(self.train, self.batch_size, shuffle=not self.debug)

    def val_dataloader(self):
        if self.val is None or len(self.val) == 0:
            print("[WARNING] Validation dataset is None or empty - validation will be skipped")  # basemodule.py:88-94
            return None
        return SpecLoader(self.val, min(max(self.batch_size, 1), len(self.val)))

    def test_dataloader(self) -> "SpecLoader":
        return SpecLoader(self.test, self.batch_size)
