"""Optimizer factory of the hot path: mirrors `OptModule` (reference src/opt/optimizer.py:1-173).

Same config keys, same scheduler table, same Lightning-style return value ({optimizer, lr_scheduler{scheduler, monitor,
...}}).  For a `MyViT` on an MI355X, 'adam'/'adamw' resolve to `FusedAdamW`: one HIP kernel over the flat parameter
buffer that applies the global-norm clip coefficient, the AdamW update and the bf16 shadow refresh in a single pass
(torch.optim.AdamW semantics, verified against it in tests/).  Any other optimizer type falls through to torch.optim
exactly as in the reference (it then works on the `.grad` views the model's backward produces).
"""
from __future__ import annotations

from typing import Optional

import torch

from . import functional as vf

__all__ = ["OptModule", "FusedAdamW"]


class FusedAdamW(torch.optim.Optimizer):
    """AdamW over MyViT's flat buffers.  `step()` == torch.optim.AdamW.step(): parameters without a gradient (frozen ones,
    the never-used pooler, everything when no backward ran) are left alone together with their moments, exactly as torch
    skips them; `state_dict()` / `load_state_dict()` speak torch.optim.AdamW's format (per-parameter `step`, `exp_avg`,
    `exp_avg_sq`, indexed in `model.parameters()` order), so optimizer state moves between this class and the reference's
    torch optimizer through a Lightning checkpoint.

    Gradient clipping: either the caller clips `.grad` beforehand (Lightning's gradient_clip_val path; the grads are
    views of the flat buffer, so that is seen here), or `set_grad_clip(max_norm)` fuses clip_grad_norm_(max_norm) into the
    step (norm kernel + coefficient applied on the fly).

    Under data parallelism (`attach_reducer`) the flat gradient buffer holds the rank-averaged gradient when `step()` runs;
    a `.grad` that is NOT the flat-buffer view (a hook or an accumulation replaced it) would silently overwrite that
    average with a local gradient, so it is an error there.  With a sharding reducer ('zero1') each rank updates only its
    shard of every sharded bucket and the updated parameters are all-gathered."""

    def __init__(self, model, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0, adam_l2: bool = False):
        self.model = model
        params = [p for p in model.parameters()]
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay))
        # parameters that are not slices of the engine's flat buffer (a trainable input preprocessor): same AdamW kernel,
        # own moment buffers, and their gradients join the global clipping norm
        own = {id(p) for p in getattr(model, "_param_list", [])}
        self._extras = [p for p in params if id(p) not in own]
        self._extra_state = {}
        self.adam_l2 = adam_l2
        self._m = None
        self._v = None
        self._step = 0
        self._clip: Optional[float] = None
        self._sq = None
        self._reducer = None
        self.last_grad_norm: Optional[torch.Tensor] = None

    def set_grad_clip(self, max_norm: Optional[float]):
        self._clip = max_norm

    def attach_reducer(self, reducer):
        self._reducer = reducer

    def _ensure_state(self):
        eng = self.model.engine
        eng._ensure_device_state()
        if self._m is None or self._m.device != eng.flat.device:
            m = torch.zeros_like(eng.flat)
            v = torch.zeros_like(eng.flat)
            if self._m is not None:  # state loaded before the model moved to the GPU
                m.copy_(self._m)
                v.copy_(self._v)
            self._m, self._v = m, v
        if self._sq is None or self._sq.device != eng.flat.device:
            self._sq = torch.zeros(1, dtype=torch.float32, device=eng.flat.device)

    def _active_ranges(self):
        """Contiguous [lo, hi) runs of the flat buffer whose parameters have a gradient this step; also folds a replaced
        `.grad` back into the flat gradient buffer (single process only)."""
        eng, lay = self.model.engine, self.model.engine.layout
        spans = []
        for name, p in zip(self.model._param_names, self.model._param_list):
            off, _ = lay.entries[name]
            if off >= lay.n_trainable or p.grad is None:
                continue
            gv = eng.g(name)
            if p.grad.data_ptr() != gv.data_ptr():
                if self._reducer is not None:
                    raise RuntimeError(f"{name}.grad is not the flat gradient buffer's view (a hook or an accumulation into "
                                       f"an existing .grad replaced it): under data parallelism that would overwrite the "
                                       f"rank-averaged gradient with a local one")
                gv.copy_(p.grad)
            # the alignment pad behind an entry belongs to it (zeros in every buffer, never read back)
            spans.append((off, off + (lay.numel(name) + 7) // 8 * 8))
        spans.sort()
        runs = []
        for a, b in spans:
            if runs and runs[-1][1] == a:
                runs[-1][1] = b
            else:
                runs.append([a, b])
        return [(a, b) for a, b in runs]

    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        eng = self.model.engine
        self._ensure_state()
        g = self.param_groups[0]
        runs = self._active_ranges()
        extras = [p for p in self._extras if p.grad is not None]
        if not runs and not extras:
            return loss  # nothing has a gradient: torch.optim would not touch anything either
        self._step += 1
        red = self._reducer if (self._reducer is not None and getattr(self._reducer, "mode", "") == "zero1") else None
        if red is not None:
            runs = self._shard_runs(runs, red)
        sq = None
        if self._clip is not None:
            sq = self._sq
            first = True
            for a, b in (runs if red is None else self._norm_runs_local):
                vf.grad_sqnorm(eng.grads[a:b], out=sq, accumulate=not first)
                first = False
            if first:
                sq.zero_()
            if red is not None:  # shard sums -> global; the all-reduced buckets are added once (identical on every rank)
                torch.distributed.all_reduce(sq)
                for a, b in self._norm_runs_shared:
                    vf.grad_sqnorm(eng.grads[a:b], out=sq, accumulate=True)
            for p in extras:
                vf.grad_sqnorm(p.grad.contiguous().view(-1), out=sq, accumulate=True)
            self.last_grad_norm = sq
        shadow = eng.shadow if eng.precision == "bf16" else None  # f32 mode has no bf16 copy to refresh
        kw = dict(lr=float(g["lr"]), beta1=g["betas"][0], beta2=g["betas"][1], eps=g["eps"], weight_decay=g["weight_decay"],
                  step=self._step, sqnorm=sq, max_norm=float(self._clip or 0.0))
        for a, b in runs:
            vf.adamw_step(eng.flat[a:b], eng.grads[a:b], self._m[a:b], self._v[a:b], None if shadow is None else shadow[a:b],
                          **kw)
        if red is not None:
            red.all_gather_params(eng.flat)
            if shadow is not None:
                vf.cast_f32_bf16(eng.flat, shadow)
        eng.mark_shadow_fresh()  # the kernel rewrote flat AND shadow through raw pointers
        for p in extras:
            st = self._extra_state.get(id(p))
            if st is None or st[0].device != p.device:
                st = self._extra_state[id(p)] = (torch.zeros_like(p.data).view(-1), torch.zeros_like(p.data).view(-1))
            vf.adamw_step(p.data.view(-1), p.grad.contiguous().view(-1), st[0], st[1], None, **kw)
        return loss

    def _shard_runs(self, runs, red):
        """Intersect the active runs with what this rank updates under 'zero1': its shard of every sharded bucket, the whole
        of every all-reduced bucket.  Also records which pieces enter the norm locally (shards) / as shared (the rest)."""
        own, local, shared = [], [], []
        for (lo, hi), sh in zip(red.buckets, red.sharded):
            a, b = red.shard(lo, hi) if sh else (lo, hi)
            for ra, rb in runs:
                x, y = max(a, ra), min(b, rb)
                if x < y:
                    own.append((x, y))
                    (local if sh else shared).append((x, y))
        self._norm_runs_local, self._norm_runs_shared = local, shared
        return own

    def zero_grad(self, set_to_none: bool = True):
        super().zero_grad(set_to_none=True)

    def gather_sharded_state(self):
        """Under the 'zero1' exchange a rank's moment buffers are current only inside its own shard of every sharded bucket.
        A checkpoint is written by rank 0 and read back by EVERY rank, so before `state_dict()` the shards are all-gathered
        into every rank's buffers (a collective: all ranks must call this; a no-op for the all-reduce exchange and for a
        single process).  Without it the other ranks resumed with zero moments (first update ~3x lr)."""
        red = self._reducer
        if red is None or getattr(red, "mode", "") != "zero1" or self._m is None or self._step == 0:
            return
        red.all_gather_params(self._m)
        red.all_gather_params(self._v)

    # ------------------------------------------------------------------ torch.optim.AdamW-format state
    def state_dict(self):
        lay = self.model.engine.layout
        index = {id(p): i for i, p in enumerate(self.param_groups[0]["params"])}
        state = {}
        if self._m is not None and self._step > 0:
            for name, p in zip(self.model._param_names, self.model._param_list):
                off, shape = lay.entries[name]
                if off >= lay.n_trainable:
                    continue
                n = lay.numel(name)
                state[index[id(p)]] = {"step": torch.tensor(float(self._step)),
                                       "exp_avg": self._m[off:off + n].detach().cpu().clone().view(shape),
                                       "exp_avg_sq": self._v[off:off + n].detach().cpu().clone().view(shape)}
            for p in self._extras:
                st = self._extra_state.get(id(p))
                if st is not None:
                    state[index[id(p)]] = {"step": torch.tensor(float(self._step)),
                                           "exp_avg": st[0].detach().cpu().clone().view(p.shape),
                                           "exp_avg_sq": st[1].detach().cpu().clone().view(p.shape)}
        groups = []
        for grp in self.param_groups:
            d = {k: (list(v) if isinstance(v, tuple) else v) for k, v in grp.items() if k != "params"}
            d["params"] = list(range(len(grp["params"])))
            groups.append(d)
        return {"state": state, "param_groups": groups}

    def load_state_dict(self, sd):
        if "state" not in sd:  # round-1 layout: flat moment buffers
            self._step = int(sd["step"])
            if sd.get("m") is not None:
                self._m, self._v = sd["m"].detach().clone().float(), sd["v"].detach().clone().float()
            for grp, src in zip(self.param_groups, sd.get("param_groups", [])):
                grp.update({k: v for k, v in src.items() if k != "params"})
            return
        lay = self.model.engine.layout
        flat = self.model.engine.flat
        if self._m is None or self._m.device != flat.device:
            self._m, self._v = torch.zeros_like(flat), torch.zeros_like(flat)
        params = self.param_groups[0]["params"]
        names = {id(p): n for n, p in zip(self.model._param_names, self.model._param_list)}
        step = 0
        for idx, st in sd["state"].items():
            p = params[int(idx)]
            step = max(step, int(float(st["step"])))
            if id(p) in names:
                off, _ = lay.entries[names[id(p)]]
                n = lay.numel(names[id(p)])
                self._m[off:off + n].copy_(st["exp_avg"].reshape(-1))
                self._v[off:off + n].copy_(st["exp_avg_sq"].reshape(-1))
            else:
                self._extra_state[id(p)] = (st["exp_avg"].detach().clone().to(p.device).float().view(-1),
                                            st["exp_avg_sq"].detach().clone().to(p.device).float().view(-1))
        self._step = step
        for grp, src in zip(self.param_groups, sd.get("param_groups", [])):
            for k, v in src.items():
                if k != "params":
                    grp[k] = tuple(v) if k == "betas" else v


# ---------------------------------------------------------------------------------------------------------------- factory
_OPTIMIZERS = {
    "adam": "Adam", "adamw": "AdamW", "sgd": "SGD", "rmsprop": "RMSprop", "adadelta": "Adadelta", "adagrad": "Adagrad",
    "adamax": "Adamax", "asgd": "ASGD", "lbfgs": "LBFGS", "rprop": "Rprop", "sparseadam": "SparseAdam",
}
_SCHEDULERS = {
    "cosine": "CosineAnnealingLR", "cosineannealing": "CosineAnnealingLR", "cosineannealinglr": "CosineAnnealingLR",
    "onecycle": "OneCycleLR", "constant": "ConstantLR", "constantlr": "ConstantLR", "plateau": "ReduceLROnPlateau",
}


def _family(name: str) -> str:
    """Which argument rules a scheduler name falls under (substring match, first hit wins, as the reference resolves it)."""
    for fam in ("cosine", "onecycle", "constant", "plateau"):
        if fam in name:
            return fam
    return ""


# per family: (required arguments as {kwarg: how to get it}, optional keys copied only when present in the config)
def _scheduler_kwargs(fam: str, cfg: dict, lr: float) -> dict:
    if fam == "cosine":
        kw = {"T_max": cfg.get("T_max", cfg.get("ep", 100))}
        optional = ("eta_min",)
    elif fam == "onecycle":  # the run length (steps_per_epoch, epochs) is injected by configure_optimizers
        kw = {"max_lr": lr}
        optional = ("steps_per_epoch", "epochs", "pct_start", "div_factor", "final_div_factor")
    elif fam == "constant":
        kw = {"factor": cfg.get("factor", 1.0), "total_iters": cfg.get("total_iters", 1)}
        optional = ()
    elif fam == "plateau":
        kw = {"factor": cfg.get("factor", 0.1), "patience": cfg.get("patience", 10)}
        optional = ("mode",)
    else:
        return {}
    kw.update({k: cfg[k] for k in optional if k in cfg})
    return kw


class OptModule:
    """The optimizer / scheduler factory behind `configure_optimizers` (reference: src/opt/optimizer.py:1-173).  Same config
    keys (`lr`, `type`, `weight_decay`, `lr_sch`, `warmup.{ratio,epochs}` / `warmup_ratio` / `warmup_epochs`, the scheduler's
    own keys), same result shape (an optimizer, or Lightning's {"optimizer", "lr_scheduler": {scheduler, monitor, ...}});
    behaviour pinned case by case on the reference's class in tests/golden/opt.json."""

    def __init__(self, lr, monitor_metric="loss", opt_type="adam", weight_decay=0.0, lr_scheduler_name=None,
                 warmup_ratio=0.0, warmup_epochs=None, **kwargs) -> None:
        self.lr = float(lr)
        self.monitor_metric = monitor_metric
        self.opt_type = opt_type
        self.weight_decay = weight_decay
        self.lr_scheduler_name = lr_scheduler_name
        self.warmup_ratio = warmup_ratio
        self.warmup_epochs = warmup_epochs
        self.kwargs = kwargs
        self.opt_fns = {k: getattr(torch.optim, v) for k, v in _OPTIMIZERS.items()}
        self.lr_schedulers = {k: getattr(torch.optim.lr_scheduler, v) for k, v in _SCHEDULERS.items()}

    @classmethod
    def from_config(cls, config):
        lr = config.get("lr", 1e-3)
        warm = config.get("warmup", {})
        common = dict(
            lr=lr, monitor_metric=config.get("monitor_metric", "loss"), opt_type=config.get("type", "adam").lower(),
            weight_decay=config.get("weight_decay", 0),
            warmup_ratio=warm.get("ratio", config.get("warmup_ratio", 0.0)),
            warmup_epochs=warm.get("epochs", config.get("warmup_epochs", None)),
        )
        if "lr_sch" not in config:
            return cls(**common)
        name = config["lr_sch"].lower()
        return cls(lr_scheduler_name=name, **common, **_scheduler_kwargs(_family(name), config, lr))

    def _make_optimizer(self, model):
        from .specvit import MyViT

        fused_ok = self.opt_type == "adamw" or (self.opt_type == "adam" and not self.weight_decay)
        if isinstance(model, MyViT) and fused_ok:
            # torch.optim.Adam's weight_decay is L2-in-the-gradient; with the reference's default weight_decay=0
            # (optimizer.py:51) Adam == AdamW, and only then does 'adam' take the fused kernel
            return FusedAdamW(model, lr=self.lr, weight_decay=self.weight_decay)
        return self.opt_fns[self.opt_type](model.parameters(), lr=self.lr, weight_decay=self.weight_decay)

    def _warmup_length(self) -> Optional[int]:
        """Epochs of linear warm-up in front of the scheduler, or None.  One-cycle brings its own."""
        wanted = self.warmup_ratio > 0 or self.warmup_epochs is not None
        if not wanted or "onecycle" in self.lr_scheduler_name:
            return None
        if self.warmup_epochs is not None:
            return self.warmup_epochs
        horizon = self.kwargs.get("T_max", self.kwargs.get("epochs", 100))
        return max(1, int(horizon * self.warmup_ratio))

    def __call__(self, model):
        optimizer = self._make_optimizer(model)
        name = self.lr_scheduler_name
        if name is None:
            return optimizer
        if name not in self.lr_schedulers:
            raise ValueError(f"Unknown scheduler: {name}")
        sched_mod = torch.optim.lr_scheduler
        n_warm = self._warmup_length()
        if n_warm is None:
            scheduler = self.lr_schedulers[name](optimizer, **self.kwargs)
        else:  # LinearLR from 10 % of the target rate, then the scheduler proper (constructed in this order: each
            # constructor records / touches the optimizer's learning rate)
            ramp = sched_mod.LinearLR(optimizer, start_factor=0.1, total_iters=n_warm)
            main = self.lr_schedulers[name](optimizer, **self.kwargs)
            scheduler = sched_mod.SequentialLR(optimizer, schedulers=[ramp, main], milestones=[n_warm])
        entry = {"scheduler": scheduler, "monitor": f"val_{self.monitor_metric}"}
        fam = _family(name)
        if fam == "plateau":
            entry.update(reduce_on_plateau=True, strict=False)  # Lightning: step after validation, tolerate a missing metric
        else:
            entry.update(interval="step" if fam == "onecycle" else "epoch", frequency=1)
        return {"optimizer": optimizer, "lr_scheduler": entry}
