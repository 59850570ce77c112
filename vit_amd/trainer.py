"""The training loop of the path, with the policy the reference gets from `BaseTrainer(L.Trainer)` / `SpecTrainer`
(src/basemodule.py:203-251, src/vit.py:349-435, 455-465) -- Lightning itself is not part of this image.

  fit:   for epoch: [unfreeze the input preprocessor when its warm-up ends, prepca/callbacks.py:31-46]
           for batch: zero_grad -> training_step (forward) -> backward (when devices > 1 the flat gradient buffer is
                      mean-all-reduced bucket by bucket while backward runs: hardware_utils.py:95 'ddp')
                      -> clip the global gradient norm to `train.grad_clip` or 0.5 (basemodule.py:244) -> optimizer step
           validate every epoch (basemodule.py:249)
           scheduler: ReduceLROnPlateau on `val_{monitor}` / per-epoch / per-step (opt/optimizer.py:150-172)
           ModelCheckpoint(save_top_k=1, monitor=val_{mae|acc}, save_last=True) when `train.save` (vit.py:386-414)
           EarlyStopping(monitor, patience 500 | 100 in a sweep, mode) (vit.py:365, 417-424)
         `fast_dev_run` (one batch of everything, nothing saved) when `train.debug` (basemodule.py:245)
         `fit(..., ckpt_path=)` resumes weights, optimizer state, scheduler state, epoch and step (vit.py:464)
  test:  evaluation only, optionally from a checkpoint path or 'best' / 'last' (scripts/test.py:26-48)

Logged scalars follow Lightning's `self.log(value, on_epoch=True)` reduction: the epoch value is the batch-size-weighted
mean of the per-batch values.  That makes `val_mae` / `val_mse` the exact epoch MAE / MSE, and `val_r2` the weighted mean of
the per-batch R^2 (what the reference logs: it passes the metric's batch VALUE to `self.log`, vit.py:113-121).  The metric
objects' own epoch totals (`compute()` over everything seen) are kept in `Trainer.metric_totals`.  Under DDP both kinds are
reduced from summed STATE (weighted sums and weights; metric sums and counts) so every rank monitors the same number --
the reference logs without `sync_dist`, i.e. rank-local values, which would let ranks disagree about early stopping.

'train.precision': '32' (default, as in the reference) -> fp32-class kernels (split-bf16 x3 GEMMs, fp32 attention);
'bf16-mixed' -> bf16 MFMA operands with fp32 master weights / residual stream / statistics (the throughput path).
"""
from __future__ import annotations

import os
import time
from typing import Any, Dict, Iterable, Optional

import torch

from . import ddp as ddp_mod
from .optimizer import FusedAdamW

__all__ = ["Trainer", "Checkpointer", "FreezeSchedule", "select_accelerator_and_devices", "get_training_strategy", "seed_everything"]


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
    """Batches of a bound SpecLoader are already on the device (no-op); anything else (a plain iterable of host tensors) is
    copied here -- from pageable memory that copy is synchronous, which is why the loaders stage through pinned buffers."""
    return tuple(t.to(device, non_blocking=True) if torch.is_tensor(t) else t for t in batch)


def _bind(loader, device):
    if loader is not None and hasattr(loader, "bind"):
        loader.bind(device)
    return loader


def _batch_size(batch) -> int:
    for t in batch:
        if torch.is_tensor(t) and t.dim() > 0:
            return int(t.shape[0])
    return 1


class Checkpointer:
    """ModelCheckpoint(save_top_k=1, monitor, mode, save_last=True): after each validation, `last.ckpt` is rewritten and,
    when the monitored value improved, `epoch={e}-{monitor}={v:.4f}.ckpt` replaces the previous best file."""

    def __init__(self, dirpath: str, monitor: str, mode: str):
        self.dirpath, self.monitor, self.mode = dirpath, monitor, mode
        self.best_score: Optional[float] = None
        self.best_path: Optional[str] = None
        self.last_path: Optional[str] = None

    def better(self, v: float) -> bool:
        if self.best_score is None:
            return True
        return v > self.best_score if self.mode == "max" else v < self.best_score

    def after_validation(self, trainer: "Trainer", module, logs: Dict[str, float]):
        opt = getattr(trainer, "optimizer", None)
        if hasattr(opt, "gather_sharded_state"):
            opt.gather_sharded_state()  # 'zero1': a collective, so before the rank check
        if trainer.rank != 0:
            return
        os.makedirs(self.dirpath, exist_ok=True)
        ckpt = trainer.make_checkpoint(module)
        v = logs.get(self.monitor)
        if v is not None and self.better(v):
            path = os.path.join(self.dirpath, f"epoch={trainer.current_epoch}-{self.monitor}={v:.4f}.ckpt")
            ckpt["callbacks"]["checkpoint"] = {"best_model_score": float(v), "best_model_path": path, "monitor": self.monitor}
            torch.save(ckpt, path)
            if self.best_path and self.best_path != path and os.path.exists(self.best_path):
                os.remove(self.best_path)
            self.best_score, self.best_path = float(v), path
        ckpt["callbacks"]["checkpoint"] = {"best_model_score": self.best_score, "best_model_path": self.best_path or "",
                                           "monitor": self.monitor}
        self.last_path = os.path.join(self.dirpath, "last.ckpt")
        torch.save(ckpt, self.last_path)

    def resolve(self, which: str) -> str:
        path = {"best": self.best_path, "last": self.last_path}.get(which, which)
        if not path and which in ("best", "last"):
            cand = os.path.join(self.dirpath, "last.ckpt")
            if which == "last" and os.path.exists(cand):
                return cand
            raise FileNotFoundError(f"no '{which}' checkpoint has been written under {self.dirpath}")
        return path


def load_checkpoint_file(path: str) -> Dict[str, Any]:
    """Checkpoints are read with a loader that executes nothing from the file."""
    ckpt = torch.load(path, map_location="cpu", weights_only=True)
    if "state_dict" not in ckpt:  # a bare state_dict (e.g. one written by round 1's --save)
        ckpt = {"state_dict": ckpt}
    return ckpt


def model_state_from_checkpoint(ckpt: Dict[str, Any]) -> Dict[str, torch.Tensor]:
    """Lightning stores the LightningModule's state_dict: the network's tensors sit under the `model.` prefix."""
    sd = ckpt["state_dict"]
    return {(k[len("model."):] if k.startswith("model.") else k): v for k, v in sd.items()}


class FreezeSchedule:
    """When the input preprocessor is trainable (reference: PreprocessorFreezeCallback, src/prepca/callbacks.py; pinned on
    it by tests/golden/freeze.json): `freeze_epochs` > 0 = frozen at the start of training and released once, at the first
    epoch >= freeze_epochs; -1 = frozen for good; 0 = never touched."""

    def __init__(self, freeze_epochs: int = 0):
        self.freeze_epochs = int(freeze_epochs or 0)
        self.released = False

    def on_train_start(self, model) -> None:
        if self.freeze_epochs != 0 and hasattr(model, "set_preprocessor_trainable"):
            model.set_preprocessor_trainable(False)

    def on_epoch_start(self, model, epoch: int) -> bool:
        """True when this call released the preprocessor."""
        due = self.freeze_epochs > 0 and epoch >= self.freeze_epochs and not self.released
        if due and hasattr(model, "set_preprocessor_trainable"):
            model.set_preprocessor_trainable(True)
            self.released = True
            return True
        return False


class Trainer:
    def __init__(self, config: Dict[str, Any], device: Optional[torch.device] = None, verbose: bool = True,
                 sweep: bool = False):
        """`config` is the `train` section (as the reference passes it: vit.py:359)."""
        self.max_epochs = config.get("ep", 10)
        self.gradient_clip_val = config.get("grad_clip", 0.5)
        self.fast_dev_run = bool(config.get("debug", False))
        self.precision = str(config.get("precision", "32"))
        self.patience = int(config.get("patience", 100 if sweep else 500))
        self.save_enabled = bool(config.get("save", False))
        self.rank, self.local_rank, self.world = ddp_mod.init_distributed()
        self.acc, self.device0 = select_accelerator_and_devices(config.get("gpus"))
        self.strategy = get_training_strategy(self.world)
        self.device = device or torch.device("cuda", self.local_rank)
        self.exchanging = ddp_mod.exchange_active()  # world > 1, or the single-rank RCCL rehearsal (VIT_DIST_SINGLE)
        self.backend = torch.distributed.get_backend() if self.exchanging else None
        self.exchange = str(config.get("ddp_exchange", os.environ.get("VIT_DDP_EXCHANGE", "allreduce")))
        # 'fp32' (the reference's DDP) | 'bf16' (half the bytes on the links, the sum rounded in bf16: an option, see ddp.py)
        self.grad_dtype = str(config.get("ddp_grad_dtype", os.environ.get("VIT_DDP_GRAD_DTYPE", "fp32")))
        self.max_bucket_elems = int(config.get("ddp_max_bucket_elems", os.environ.get("VIT_DDP_MAX_BUCKET_ELEMS", 64 << 20)))
        # train.ddp_reserve_cus: CUs the persistent one-workgroup-per-CU kernels leave to the collective (an int), or 'auto':
        # fit() times a few steps on its first batch at 0 / 8 / 16 / 32 and keeps the fastest (autotune_reserve_cus)
        self.reserve_cus_cfg = config.get("ddp_reserve_cus", os.environ.get("VIT_DDP_RESERVE_CUS", 0))
        self.reserve_cus = 0
        # train.hip_graph: replay the optimisation step as one captured hipGraph (vit_amd/graph.py; single GPU only)
        self.use_graph = bool(config.get("hip_graph", False))
        self._graphed = None
        self.verbose = verbose and self.rank == 0
        self.logged: Dict[str, float] = {}
        self.metric_totals: Dict[str, float] = {}
        self._acc: Dict[str, list] = {}
        self._cur_bs = 1
        self.global_step = 0
        self.current_epoch = 0
        self.should_stop = False
        self.history = []
        self.checkpointer: Optional[Checkpointer] = None
        self.optimizer = None
        self.sched_cfg = None
        self.reducer = None
        self._ready_for = None

    # ------------------------------------------------------------------ logging (LightningModule.log lands here)
    def _log(self, name, value, on_step=None, on_epoch=None):
        v = value.detach().to(torch.float64).reshape(()) if torch.is_tensor(value) else torch.tensor(float(value), dtype=torch.float64)
        slot = self._acc.setdefault(name, [None, 0.0])
        w = float(self._cur_bs)
        slot[0] = v * w if slot[0] is None else slot[0] + v.to(slot[0].device) * w
        slot[1] += w

    def _flush_epoch_logs(self) -> Dict[str, float]:
        names = sorted(self._acc)
        if not names:
            return {}
        sums = torch.stack([self._acc[n][0].to(self.device if self.device.type == "cuda" and torch.cuda.is_available() else "cpu")
                            for n in names])
        wts = torch.tensor([self._acc[n][1] for n in names], dtype=torch.float64, device=sums.device)
        if self.world > 1:  # every rank logs the same names in the same order (same step functions)
            both = torch.stack([sums, wts])
            torch.distributed.all_reduce(both)
            sums, wts = both[0], both[1]
        self._acc = {}
        return {n: float(s / w) for n, s, w in zip(names, sums, wts)}

    # ------------------------------------------------------------------ setup
    def _setup(self, module, for_training: bool = True):
        module.trainer = self
        module.to(self.device)
        if hasattr(module.model, "set_precision"):
            module.model.set_precision(self.precision)  # basemodule.py:233: precision=str(train.precision or '32')
        if not for_training or self._ready_for is module:
            return
        conf = module.configure_optimizers()
        if isinstance(conf, dict):
            self.optimizer = conf["optimizer"]
            self.sched_cfg = conf.get("lr_scheduler")
        else:
            self.optimizer, self.sched_cfg = conf, None
        self.optimizers = [self.optimizer]
        eng = module.model.engine
        self.reducer = None
        if self.exchanging:
            eng._ensure_device_state()
            ddp_mod.broadcast_parameters(eng.flat)
            eng._shadow_version = -1
            self.reducer = ddp_mod.make_reducer(self.exchange, eng, grad_dtype=self.grad_dtype,
                                                max_bucket_elems=self.max_bucket_elems)
            eng.grad_ready_cb = self.reducer.bucket_ready
        if isinstance(self.optimizer, FusedAdamW):
            self.optimizer.set_grad_clip(self.gradient_clip_val)
            self.optimizer.attach_reducer(self.reducer)
        # freeze schedule of the input preprocessor (src/prepca/callbacks.py): warmup.freeze_epochs > 0 freezes it for
        # that many epochs, -1 for good, 0 never.  (As in the reference, the optimizer was built from the parameters that
        # existed at configure time: a preprocessor unfrozen later becomes trainable for autograd but is only stepped if it
        # already was among the optimizer's parameters.)
        warm = (getattr(module, "config", {}) or {}).get("warmup") or {}
        self.freeze = FreezeSchedule(warm.get("freeze_epochs", 0))
        self.freeze.on_train_start(module.model)
        mon = getattr(module, "monitor_metric", "loss")
        self.monitor = f"val_{mon}"
        self.monitor_mode = "max" if mon == "acc" else "min"
        if self.save_enabled and not self.fast_dev_run:
            self.checkpointer = Checkpointer(os.environ.get("CKPT_DIR", "./checkpoints"), self.monitor, self.monitor_mode)
        self._ready_for = module

    # ------------------------------------------------------------------ room for the collective
    def _snapshot(self, module):
        """Everything a training_step changes: parameters (the engine's flat buffer + a trainable preprocessor's), optimizer
        state and step count, scheduler, global_step, the dropout stream's position, the epoch's running log sums and
        torch's CPU generator (the noise injection draws its per-step seed from it)."""
        import copy

        eng, opt = module.model.engine, self.optimizer
        eng._ensure_device_state()
        snap = {"flat": eng.flat.detach().clone(), "step_counter": eng.step_counter, "global_step": self.global_step,
                "acc": {k: [None if v[0] is None else v[0].clone(), v[1]] for k, v in self._acc.items()},
                "rng": torch.random.get_rng_state(),
                "sched": copy.deepcopy(self.sched_cfg["scheduler"].state_dict()) if self.sched_cfg else None,
                # every hyper-parameter of every group: a one-cycle scheduler also cycles Adam's beta1 / SGD's momentum per step
                "groups": [copy.deepcopy({k: v for k, v in g.items() if k != "params"}) for g in opt.param_groups]}
        if isinstance(opt, FusedAdamW):
            snap["fused"] = (opt._step, None if opt._m is None else opt._m.clone(), None if opt._v is None else opt._v.clone(),
                             {k: (a.clone(), b.clone()) for k, (a, b) in opt._extra_state.items()},
                             [p.detach().clone() for p in opt._extras])
        else:
            snap["opt"] = copy.deepcopy(opt.state_dict())
            snap["extra_params"] = [p.detach().clone() for p in module.parameters()]
        return snap

    def _restore(self, module, snap):
        eng, opt = module.model.engine, self.optimizer
        torch.cuda.synchronize(self.device)
        with torch.no_grad():
            eng.flat.copy_(snap["flat"])
        eng._shadow_version = -1  # the bf16 shadow is re-cast from the restored master weights
        eng.step_counter, self.global_step = snap["step_counter"], snap["global_step"]
        self._acc = snap["acc"]
        torch.random.set_rng_state(snap["rng"])
        if self.sched_cfg:
            self.sched_cfg["scheduler"].load_state_dict(snap["sched"])
        if "fused" in snap:
            step, m, v, extra_state, extras = snap["fused"]
            opt._step = step
            if m is None:
                opt._m = opt._v = None
            else:
                opt._m.copy_(m); opt._v.copy_(v)
            opt._extra_state = extra_state
            with torch.no_grad():
                for p, q in zip(opt._extras, extras):
                    p.copy_(q)
        else:
            opt.load_state_dict(snap["opt"])
            with torch.no_grad():
                for p, q in zip(module.parameters(), snap["extra_params"]):
                    p.copy_(q)
        for g, saved in zip(opt.param_groups, snap["groups"]):
            g.update(saved)
        opt.zero_grad(set_to_none=True)

    def set_reserve_cus(self, module, n: int):
        """Launch geometry of THIS module's engine (its own vit_handle): nothing process-wide changes."""
        self.reserve_cus = int(n)
        module.model.engine.set_reserve_cus(self.reserve_cus)

    def autotune_reserve_cus(self, module, batch, candidates=(0, 8, 16, 32), steps=3, restore: bool = True):
        """Data-parallel runs only: pick how many CUs the one-workgroup-per-CU kernels (ping-pong GEMMs, pair-pipelined attention
        backward) leave free for the collective's kernels that overlap the backward (vit_handle_set_option "reserve_cus"), by
        timing `steps` optimisation steps on `batch` at each candidate (median of per-step hipEvent times, MAX over ranks) and
        keeping the fastest; every rank takes the same decision.  The timed steps are real optimisation steps, so the state
        they change is snapshotted before and put back afterwards (`restore`; ADVICE r4: with lr 1e-3 and no warm-up ONE step
        moves the C3 loss 0.185 -> 266, and the reference's DDP does nothing of the kind before epoch 0,
        src/basemodule.py:226-251): a run with 'auto' continues bit for bit like a run with the chosen value fixed.  bench.py
        calls it ahead of its warm-up with restore=False (those steps ARE warm-up).  Returns {candidate: ms}."""
        if not self.exchanging:
            return {}
        dist = torch.distributed
        snap = self._snapshot(module) if restore else None
        # a per-step scheduler (one-cycle) has a fixed number of steps to give: the probe steps do not take from it
        sched_cfg, self.sched_cfg = self.sched_cfg, (None if restore else self.sched_cfg)
        out = {}
        for c in candidates:
            self.set_reserve_cus(module, int(c))
            self.training_step(module, batch, 0)  # the grids' first launch at this size
            ev = [torch.cuda.Event(enable_timing=True) for _ in range(steps + 1)]
            torch.cuda.synchronize()
            dist.barrier()
            ev[0].record()
            for i in range(steps):
                self.training_step(module, batch, i)
                ev[i + 1].record()
            torch.cuda.synchronize()
            t = torch.tensor([ev[i].elapsed_time(ev[i + 1]) for i in range(steps)], dtype=torch.float64, device=self.device)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            out[int(c)] = float(t.median())
        best = min(out, key=lambda k: (out[k], k))
        self.set_reserve_cus(module, best)
        self.sched_cfg = sched_cfg
        if snap is not None:
            self._restore(module, snap)
        return out

    # ------------------------------------------------------------------ one optimisation step
    def training_step(self, module, batch, batch_idx):
        """zero_grad -> forward -> backward (+ overlapped gradient exchange) -> clip -> optimizer step."""
        self._cur_bs = _batch_size(batch)
        if self.use_graph and not self.exchanging:
            return self._graph_step(module, batch)
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

    def _graph_step(self, module, batch):
        """One captured graph per (batch shape, precision), at most two kept (the full batch and an epoch's partial last
        batch; each holds its own activation arena).  What the capture cannot take -- another optimizer, a trainable input
        preprocessor, on-the-fly noise -- falls back to eager launches for the rest of the run, with one warning."""
        from .graph import GraphedTrainStep

        key = (tuple(batch[0].shape), module.model.engine.precision)
        if self._graphed is None:
            self._graphed = {}
        g = self._graphed.get(key)
        if g is None:
            try:
                g = GraphedTrainStep(module, self.optimizer, batch)  # capture (runs warm-up steps on this batch)
            except (TypeError, ValueError) as e:
                import warnings

                warnings.warn(f"train.hip_graph: {e}; continuing with eager launches")
                self.use_graph = False
                return self.training_step(module, batch, 0)
            while len(self._graphed) >= 2:
                self._graphed.pop(next(iter(self._graphed)))
            self._graphed[key] = g
        loss = g.step(batch)
        module.log(f"{module.loss_name}_loss", loss, on_step=True, on_epoch=True, prog_bar=True)
        if self.sched_cfg and self.sched_cfg.get("interval") == "step":
            self.sched_cfg["scheduler"].step()
        self.global_step += 1
        return loss

    # ------------------------------------------------------------------ evaluation
    @torch.no_grad()
    def validate(self, module, loader, prefix="val"):
        was_training = module.training
        module.eval()
        metrics = module.epoch_metrics() if hasattr(module, "epoch_metrics") else {}
        for m in metrics.values():
            m.reset()
        stage = "validation" if prefix == "val" else "test"
        _bind(loader, self.device)
        hook = getattr(module, f"on_{stage}_start", None)
        if hook:
            hook()
        step = module.validation_step if prefix == "val" else module.test_step
        for i, batch in enumerate(loader):
            batch = _to_device(batch, self.device)
            self._cur_bs = _batch_size(batch)
            step(batch, i)
            if self.fast_dev_run:
                break
        self._cur_bs = 1
        hook = getattr(module, f"on_{stage}_epoch_end", None)
        if hook:
            hook()
        logs = self._flush_epoch_logs()
        for name, m in metrics.items():
            if m.n:
                m.sync()  # DDP: sum the metric STATE over ranks, then compute
                self.metric_totals[f"{prefix}_{name}"] = float(m.compute())
        self.logged.update(logs)
        if was_training:
            module.train()
        return logs

    def test(self, module, loader, ckpt_path: Optional[str] = None):
        """Evaluation only (scripts/test.py:26-48): no optimizer is built, no parameter changes."""
        if ckpt_path not in (None, "", "none", "None"):
            path = self.checkpointer.resolve(ckpt_path) if self.checkpointer else ckpt_path
            module.model.load_state_dict(model_state_from_checkpoint(load_checkpoint_file(path)))
        self._setup(module, for_training=False)
        return self.validate(module, loader, "test")

    # ------------------------------------------------------------------ checkpoints
    def make_checkpoint(self, module) -> Dict[str, Any]:
        """Lightning's checkpoint layout (the keys the reference's `trainer.fit(ckpt_path=)` / `test(ckpt_path=)` read)."""
        sd = {f"model.{k}": v.detach().cpu().clone() for k, v in module.model.state_dict().items()}
        sch = self.sched_cfg["scheduler"].state_dict() if self.sched_cfg else None
        return {
            "epoch": int(self.current_epoch), "global_step": int(self.global_step), "state_dict": sd,
            # Lightning's loader reads this key to decide on checkpoint migrations; the reference pins lightning 2.5.4 (requirements.txt:23)
            "pytorch-lightning_version": "2.5.4",
            "optimizer_states": [self.optimizer.state_dict()] if self.optimizer is not None else [],
            "lr_schedulers": [_plain(sch)] if sch is not None else [],
            "callbacks": {"early_stopping": {"best_score": self._es_best, "wait_count": int(self._es_bad)}},
            "hyper_parameters": {"config": _plain(getattr(module, "config", {}))},
            # position of the dropout stream (seed, steps drawn): a resumed run continues the same mask sequence
            "vit_amd": {"version": 2, "dropout_base_seed": int(module.model.engine.base_seed),
                        "dropout_step": int(module.model.engine.step_counter)},
        }

    def _resume(self, module, ckpt_path: str) -> int:
        ckpt = load_checkpoint_file(ckpt_path)
        module.model.load_state_dict(model_state_from_checkpoint(ckpt))
        if ckpt.get("optimizer_states"):
            self.optimizer.load_state_dict(ckpt["optimizer_states"][0])
        if ckpt.get("lr_schedulers") and self.sched_cfg:
            self.sched_cfg["scheduler"].load_state_dict(ckpt["lr_schedulers"][0])
        es = (ckpt.get("callbacks") or {}).get("early_stopping") or {}
        self._es_best, self._es_bad = es.get("best_score"), int(es.get("wait_count", 0))
        ck = (ckpt.get("callbacks") or {}).get("checkpoint") or {}
        if self.checkpointer and ck.get("best_model_score") is not None:
            self.checkpointer.best_score, self.checkpointer.best_path = ck["best_model_score"], ck.get("best_model_path") or None
        own = ckpt.get("vit_amd") or {}
        if "dropout_step" in own:
            module.model.engine.base_seed = int(own["dropout_base_seed"])
            module.model.engine.step_counter = int(own["dropout_step"])
        self.global_step = int(ckpt.get("global_step", 0))
        return int(ckpt["epoch"]) + 1 if "epoch" in ckpt else 0  # saved at the END of that epoch

    # ------------------------------------------------------------------ fit
    @staticmethod
    def freeze_heap():
        """Run one full collection, then move everything that survives into the cyclic GC's permanent generation
        (`gc.freeze()`).  The step loop launches ~600 kernels per step from Python and allocates a few thousand tracked
        objects doing so; without this, a generation-2 collection every few dozen steps walks the ~10^6 objects that importing
        torch left behind -- a 50-100 ms pause of the launching thread during which the GPU drains (measured in bench.py's
        per-step list: one 105 ms step among 35 ms ones).  With the start-up heap frozen, collections only scan what the steps
        themselves allocated.  Idempotent; the reference gets the same effect from nothing (its step is a handful of Python
        calls into a C++ autograd engine).  `fit()` undoes it (gc.unfreeze()) when it returns."""
        import gc

        gc.collect()
        gc.freeze()

    def fit(self, module, train_loader: Iterable, val_loader: Optional[Iterable] = None, ckpt_path: Optional[str] = None):
        self._es_best, self._es_bad = None, 0
        self._setup(module)
        _bind(train_loader, self.device)
        first_epoch = self._resume(module, ckpt_path) if ckpt_path else 0
        peeked = None  # (first batch, the rest) of a ONE-SHOT iterable whose first batch the autotune looked at
        if self.exchanging and str(self.reserve_cus_cfg) not in ("0", "", "None"):
            if str(self.reserve_cus_cfg).lower() == "auto":
                # times a few steps on the first batch and puts every piece of training state back (autotune_reserve_cus):
                # the batch is only PEEKED -- a re-iterable loader starts again from its first batch, a one-shot iterator
                # gets the batch chained back in front
                module.train()
                it = iter(train_loader)
                first = next(it, None)
                if it is train_loader:
                    peeked = (first, it)
                if first is not None:
                    tuned = self.autotune_reserve_cus(module, _to_device(first, self.device))
                    if self.verbose:
                        print(f"[trainer] reserve_cus autotune (ms per step) {tuned} -> {self.reserve_cus}")
                if hasattr(it, "close") and it is not train_loader:
                    it.close()
            else:
                self.set_reserve_cus(module, int(self.reserve_cus_cfg))
        import gc
        import itertools

        self.freeze_heap()
        try:
            epochs = 1 if self.fast_dev_run else self.max_epochs
            for epoch in range(first_epoch, epochs):
                self.current_epoch = module.current_epoch = epoch
                if self.freeze.on_epoch_start(module.model, epoch) and self.verbose:
                    print(f"[trainer] epoch {epoch}: input preprocessor unfrozen")
                if hasattr(train_loader, "set_epoch"):
                    train_loader.set_epoch(epoch)
                module.train()
                t0 = time.time()
                batches = train_loader
                if peeked is not None:
                    batches = itertools.chain([] if peeked[0] is None else [peeked[0]], peeked[1])
                    peeked = None
                for i, batch in enumerate(batches):
                    self.training_step(module, _to_device(batch, self.device), i)
                    if self.fast_dev_run:
                        break
                logs = self._flush_epoch_logs()
                if val_loader is not None:
                    logs.update(self.validate(module, val_loader, "val"))
                self._step_epoch_scheduler(logs)
                logs["lr"] = self.optimizer.param_groups[0]["lr"]
                logs["epoch_time_s"] = time.time() - t0
                self.history.append(logs)
                self.logged.update(logs)
                if self.verbose:
                    print(f"[epoch {epoch}] " + " ".join(f"{k}={v:.5g}" for k, v in sorted(logs.items())))
                if self.checkpointer is not None and val_loader is not None:
                    self.checkpointer.after_validation(self, module, logs)
                if self._early_stop(logs):
                    self.should_stop = True
                    break
        finally:
            # everything alive at freeze time (models, engines, optimizers of EARLIER fits in this process) sat in
            # the permanent generation for good; hand it back so that cyclic garbage among them can be collected again
            gc.unfreeze()
        return self.history

    def _step_epoch_scheduler(self, logs):
        if not self.sched_cfg or self.sched_cfg.get("interval", "epoch") == "step":
            return
        sch = self.sched_cfg["scheduler"]
        if self.sched_cfg.get("reduce_on_plateau"):
            if self.monitor in logs:  # strict=False: a missing metric skips the scheduler silently
                sch.step(logs[self.monitor])
        else:
            sch.step()

    def _early_stop(self, logs) -> bool:
        """EarlyStopping(monitor, patience, mode, strict=False): stop after `patience` validations without improvement."""
        if self.monitor not in logs:
            return False
        v = logs[self.monitor]
        improved = self._es_best is None or (v > self._es_best if self.monitor_mode == "max" else v < self._es_best)
        if improved:
            self._es_best, self._es_bad = float(v), 0
            return False
        self._es_bad += 1
        return self._es_bad >= self.patience


def _plain(obj):
    """Containers of tensors / numbers / strings only, so the file loads under torch.load(weights_only=True)."""
    if isinstance(obj, dict):
        return {str(k) if not isinstance(k, (int, str)) else k: _plain(v) for k, v in obj.items()}
    if isinstance(obj, (list, tuple)):
        return [_plain(v) for v in obj]
    if torch.is_tensor(obj) or isinstance(obj, (int, float, str, bool)) or obj is None:
        return obj
    return str(obj)
