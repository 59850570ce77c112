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

__all__ = ["SpecDataset", "SpecLoader"]


class SpecDataset:
    def __init__(self, flux, error, params, *, task: str = "reg", stage: str = "train", label_norm: str = "none",
                 noise_level: float = 0.0, stats: Optional[dict] = None, noise_seed: int = 42, eps: float = 1e-8):
        self.flux = torch.as_tensor(flux, dtype=torch.float32).clip(min=0.0)  # base.py:236
        self.error = torch.as_tensor(error, dtype=torch.float32)
        if self.error.shape != self.flux.shape:
            raise ValueError("flux and error must have the same shape")
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
    def from_npz(cls, path, param_keys: Sequence[str] = ("log_g",), **kw):
        import numpy as np

        raw = np.load(path)
        cols = [torch.from_numpy(raw[k]).float() for k in param_keys]
        params = cols[0] if len(cols) == 1 else torch.stack(cols, dim=1)
        return cls(raw["flux"], raw["error"], params, **kw)

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
