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

__all__ = ["SpecDataset", "SpecLoader", "SpecDataModule"]  # _Stager, _step_reads: the input path behind SpecLoader


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


def _step_reads(ds: "SpecDataset") -> Tuple[bool, ...]:
    """Which items of a batch tuple the step functions read (vit_amd/module.py; reference src/vit.py:83-92, 94-110):
    training (flux, error, labels): `error` only feeds the noise injection, dead when noise_level == 0;
    evaluation 4-tuples (noisy, flux, error, labels): `noisy` or `flux`, never `error`; 3-tuples: flux.  Items the
    step never reads are NOT moved to the device (their slot in the tuple is None): at ViT-B/16 224^2 the training split's
    `error` is half of the 103 MB the reference's loader ships per step."""
    if ds.noisy is not None:
        return (True, True, False, True)  # the module picks noisy or clean by ITS noise level (module.py: _eval_inputs)
    train = ds.stage in (None, "fit", "train")
    return (True, bool(train and ds.noise_level > 0), True)


class _Stager:
    """Host-resident split -> device batches, `depth` batches ahead of the step (the reference: DataLoader(pin_memory,
    persistent_workers), src/basemodule.py:76-85).  A worker thread gathers the rows of batch k + depth into a PINNED
    staging buffer (`numpy.take`: one thread, GIL released) and enqueues their H2D copy on a copy stream of its own;
    the consumer only makes its stream wait for that copy's event.  `depth + 1` slots of (pinned, device) buffers; a
    slot is refilled only after the step that read it has been enqueued (event recorded when the consumer lets go of it),
    so a batch stays valid until the next-but-`depth` one is asked for."""

    def __init__(self, cols, batches, device, depth: int = 2, buffers=None, carry=None):
        import queue
        import threading

        self.cols, self.batches, self.device = cols, batches, device
        self.nslots = depth + 1
        rows = max((len(j) for j in batches), default=0)
        if buffers is None:
            buffers = self.make_buffers(cols, rows, device, self.nslots)
        self.pinned, self.dev = buffers
        # `carry` = (ready, freed) events of the stager that used these buffers in the previous epoch: its last copies out of
        # the pinned buffers and the last steps that read the device buffers order this epoch's first refills
        self.ready, self.freed = carry if carry is not None else ([torch.cuda.Event() for _ in range(self.nslots)],
                                                                  [None] * self.nslots)
        self.stream = torch.cuda.Stream(device=device)
        self.free_q: "queue.Queue" = queue.Queue()
        self.ready_q: "queue.Queue" = queue.Queue()
        for sl in range(self.nslots):
            self.free_q.put(sl)
        self.held = None
        self.stop = False
        self.error = None
        self.thread = threading.Thread(target=self._work, name="vit_amd-stager", daemon=True)
        self.thread.start()

    @staticmethod
    def make_buffers(cols, rows, device, nslots):
        """(pinned, device) staging buffers, one set per slot; the loader keeps them across epochs (page-locking 3 x 51 MB
        is tens of milliseconds)."""
        pinned = [[None if c is None else torch.empty((rows,) + tuple(c.shape[1:]), dtype=c.dtype, pin_memory=True)
                   for c in cols] for _ in range(nslots)]
        dev = [[None if c is None else torch.empty((rows,) + tuple(c.shape[1:]), dtype=c.dtype, device=device)
                for c in cols] for _ in range(nslots)]
        return pinned, dev

    def _work(self):
        import numpy as np

        try:
            torch.cuda.set_device(self.device)
            for j in self.batches:
                sl = self.free_q.get()
                if sl is None or self.stop:
                    return
                n = len(j)
                self.ready[sl].synchronize()  # the slot's previous copy has left its pinned buffer (long since)
                jn = j.numpy()
                for c, pin in zip(self.cols, self.pinned[sl]):
                    if c is None:
                        continue
                    try:
                        # one thread at memcpy speed, GIL released (2.5 ms for 256 x 50176 f32).  Not torch.index_select: its
                        # OpenMP team is sized by the HOST's core count (128 on the GPU box) whatever the job's CPU share is
                        # (16 there) -- measured 64 ms for the same gather, and the spinning team starves the launching thread
                        np.take(c.numpy(), jn, axis=0, out=pin[:n].numpy(), mode="clip")
                    except (TypeError, RuntimeError):  # a dtype numpy does not have (bf16)
                        torch.index_select(c, 0, j, out=pin[:n])
                with torch.cuda.stream(self.stream):
                    if self.freed[sl] is not None:
                        self.stream.wait_event(self.freed[sl])  # the step that read this slot's device buffers
                    for pin, dv in zip(self.pinned[sl], self.dev[sl]):
                        if pin is not None:
                            dv[:n].copy_(pin[:n], non_blocking=True)
                    self.ready[sl].record(self.stream)
                self.ready_q.put((sl, n))
            self.ready_q.put(None)
        except BaseException as e:  # noqa: BLE001 - surfaced by the consumer
            self.error = e
            self.ready_q.put(None)

    def _release(self):
        if self.held is not None:
            ev = torch.cuda.Event()
            ev.record(torch.cuda.current_stream(self.device))
            self.freed[self.held] = ev
            self.free_q.put(self.held)
            self.held = None

    def __iter__(self):
        return self

    def __next__(self):
        self._release()
        item = self.ready_q.get()
        if item is None:
            self.close()
            if self.error is not None:
                raise self.error
            raise StopIteration
        sl, n = item
        torch.cuda.current_stream(self.device).wait_event(self.ready[sl])
        self.held = sl
        return tuple(None if dv is None else dv[:n] for dv in self.dev[sl])

    def close(self):
        self._release()
        self.stop = True
        self.free_q.put(None)
        if self.thread.is_alive() and self.thread is not __import__("threading").current_thread():
            self.thread.join(timeout=5.0)

    def __del__(self):
        try:
            self.close()
        except Exception:  # noqa: BLE001
            pass


class SpecLoader:
    """Batches with the reference's DataLoader semantics that matter to the step (basemodule.py:76-85): shuffle on the
    training split unless debugging, DistributedSampler-style sharding over ranks.  The reference ships `2 * B * L * 4`
    bytes per step from pageable host memory through pinned staging (`pin_memory`, workers); here, once a device is bound
    (`bind(device)`, done by `Trainer.fit / validate`), the batch is produced ON the MI355X:

      placement 'device' -- the tensors the step reads are uploaded ONCE and a batch is a row gather in HBM (an index
                            vector per epoch is all that crosses PCIe): 288 GB of HBM hold 1.4 M spectra of L = 50176.
      placement 'host'   -- the split stays in host memory; a worker thread stages batches through pinned buffers and a
                            copy stream, `prefetch` batches ahead of the step (`_Stager`).
      placement 'auto'   -- 'device' when the needed tensors fit a quarter of the HBM that is free, else 'host'.

    Without a bound device (CPU tests, tools) it yields host tensors by whole-tensor indexing as before.  Tuple items the
    step never reads (`_step_reads`) are None on the device paths."""

    def __init__(self, ds: SpecDataset, batch_size: int, shuffle: bool = False, seed: int = 42, drop_last: bool = False,
                 placement: str = "auto", prefetch: int = 2, ship_error: Optional[bool] = None):
        """`ship_error`: None = only when the dataset's noise level says the training step reads it; True = always (a module
        whose noise level is set apart from the dataset's)."""
        self.ds, self.bs, self.shuffle, self.seed, self.epoch, self.drop_last = ds, batch_size, shuffle, seed, 0, drop_last
        self.ship_error = ship_error
        if placement not in ("auto", "device", "host"):
            raise ValueError(f"placement must be 'auto', 'device' or 'host', got '{placement}'")
        self.placement, self.prefetch = placement, max(1, int(prefetch))
        self.device: Optional[torch.device] = None
        self._resident = None  # device copies of the columns the step reads
        self.resolved: Optional[str] = None
        self.stage_ahead = True  # placement 'host': stage the next epoch's first batches when an epoch ends

    def set_epoch(self, e: int) -> None:
        self.epoch = e

    def close(self) -> None:
        """Stop a staging thread that runs ahead of the next epoch (idempotent; also on garbage collection)."""
        ahead, self._ahead = getattr(self, "_ahead", None), None
        if ahead is not None:
            ahead[1].close()
            self._stage_carry = (ahead[1].ready, ahead[1].freed)

    def __del__(self):
        try:
            self.close()
        except Exception:  # noqa: BLE001
            pass

    def __len__(self) -> int:
        n = len(self._indices())
        return n // self.bs if self.drop_last else -(-n // self.bs)

    def bind(self, device) -> "SpecLoader":
        """Produce batches on `device` from now on (no-op for a CPU device)."""
        device = torch.device(device)
        if device.type != "cuda":
            return self
        if self.device != device:
            self.device, self._resident, self.resolved = device, None, None
        return self

    def _columns(self):
        ds = self.ds
        cols = (ds.noisy, ds.flux, ds.error, ds.labels) if ds.noisy is not None else (ds.flux, ds.error, ds.labels)
        reads = list(_step_reads(ds))
        if self.ship_error is not None:
            reads[-2] = bool(self.ship_error)
        return tuple(c if need else None for c, need in zip(cols, reads))

    def _indices(self, epoch: Optional[int] = None) -> torch.Tensor:
        import torch.distributed as dist

        from .ddp import shard_indices

        rank = dist.get_rank() if dist.is_initialized() else 0
        world = dist.get_world_size() if dist.is_initialized() else 1
        return shard_indices(len(self.ds), rank, world, self.epoch if epoch is None else epoch, self.shuffle, self.seed)

    def _batches(self, epoch: Optional[int] = None):
        idx = self._indices(epoch)
        out = []
        for i in range(0, len(idx), self.bs):
            j = idx[i:i + self.bs]
            if self.drop_last and len(j) < self.bs:
                break
            out.append(j)
        return out

    def _resolve(self, cols) -> str:
        if self.resolved is None:
            mode = self.placement
            if mode == "auto":
                need = sum(c.numel() * c.element_size() for c in cols if c is not None)
                free, _total = torch.cuda.mem_get_info(self.device)
                mode = "device" if need <= free // 4 else "host"
            self.resolved = mode
        return self.resolved

    def __iter__(self) -> Iterator[Tuple[Optional[torch.Tensor], ...]]:
        if self.device is None:
            for j in self._batches():
                yield self.ds[j]
            return
        cols = self._columns()
        if self._resolve(cols) == "device":
            if self._resident is None:
                self._resident = tuple(None if c is None else c.to(self.device) for c in cols)
            batches = self._batches()
            # the epoch's order crosses PCIe once, from pinned memory and asynchronously: a pageable .to(device) is a
            # synchronous copy on the stream and drains the queue of launched-ahead steps at every epoch start
            order = torch.cat(batches).pin_memory().to(self.device, non_blocking=True) if batches else None
            pos = 0
            for j in batches:
                jd = order[pos:pos + len(j)]
                pos += len(j)
                yield tuple(None if c is None else c.index_select(0, jd) for c in self._resident)
            return
        yield from self._iter_staged(cols)

    def _iter_staged(self, cols):
        """placement 'host'.  The staging buffers and the slots' events live across epochs; when an epoch has been consumed
        to its end, the FIRST batches of the next one (epoch + 1: what `Trainer.fit` asks for next) are staged right away, so
        that an epoch starts with its batches on the device instead of with a 2.5 ms gather + 0.9 ms copy in the open (a short
        epoch -- 12 steps -- measured 0.971 x the pre-staged rate without this).  A next iteration that asks for anything
        else (another epoch number, the same epoch again) discards what was staged."""
        want = (self.epoch if self.shuffle else None, self.bs, self.drop_last, tuple(c is None for c in cols), len(self.ds))
        st, self._ahead = None, getattr(self, "_ahead", None)
        if self._ahead is not None:
            have, cand = self._ahead
            self._ahead = None
            if have == want:
                st = cand
            else:
                cand.close()
                self._stage_carry = (cand.ready, cand.freed)
        if st is None:
            st = self._make_stager(cols, self.epoch)
        done = False
        try:
            yield from st
            done = True
        finally:
            st.close()  # releases the batch still held (an event on the consumer's stream) and stops the worker
            self._stage_carry = (st.ready, st.freed)
        if done and self.stage_ahead:
            nxt = self.epoch + 1
            key = (nxt if self.shuffle else None,) + want[1:]
            self._ahead = (key, self._make_stager(cols, nxt))

    def _make_stager(self, cols, epoch):
        batches = self._batches(epoch)
        rows = max((len(j) for j in batches), default=0)
        key = (rows, self.prefetch + 1, self.device, tuple(c is None for c in cols))
        if getattr(self, "_stage_key", None) != key:
            self._stage_bufs, self._stage_key = _Stager.make_buffers(cols, rows, self.device, self.prefetch + 1), key
            self._stage_carry = None
        return _Stager(cols, batches, self.device, depth=self.prefetch, buffers=self._stage_bufs, carry=self._stage_carry)


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
        self.placement = str(data.get("placement", "auto"))  # 'auto' | 'device' | 'host' (SpecLoader)
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
        return SpecLoader(self.train, self.batch_size, shuffle=not self.debug, placement=self.placement)

    def val_dataloader(self):
        if self.val is None or len(self.val) == 0:
            print("[WARNING] Validation dataset is None or empty - validation will be skipped")  # basemodule.py:88-94
            return None
        return SpecLoader(self.val, min(max(self.batch_size, 1), len(self.val)), placement=self.placement)

    def test_dataloader(self) -> "SpecLoader":
        return SpecLoader(self.test, self.batch_size, placement=self.placement)
