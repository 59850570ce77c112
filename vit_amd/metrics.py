"""Epoch metrics of the eval path (the reference uses torchmetrics: src/vit.py:66-73, 104-125 -- `MeanAbsoluteError`,
`MeanSquaredError`, `R2Score`, `Accuracy`).  torchmetrics is not part of this image, so these are written from the
definitions, with torchmetrics' calling convention: `metric(preds, target)` updates the running state AND returns the
value on that batch; `compute()` is the value over everything seen since `reset()`; under DDP the STATE (sums and counts)
is summed across ranks before `compute()` (`sync`), never the per-rank values.

State lives in float64 on the device of the first update (no host sync per step).
"""
from __future__ import annotations

from typing import Dict, Optional

import torch

__all__ = ["MeanAbsoluteError", "MeanSquaredError", "R2Score", "Accuracy"]


class _SumMetric:
    """A metric that is a function of a few running sums."""

    fields = ()

    def __init__(self):
        self.reset()

    def reset(self):
        self._state: Optional[torch.Tensor] = None  # float64 [len(fields)], on the data's device
        self._updates = 0

    @property
    def n(self) -> int:
        """Number of updates since reset (0 = nothing to compute)."""
        return self._updates

    def _sums(self, preds: torch.Tensor, target: torch.Tensor) -> torch.Tensor:
        raise NotImplementedError

    def _value(self, s: torch.Tensor) -> torch.Tensor:
        raise NotImplementedError

    def update(self, preds: torch.Tensor, target: torch.Tensor) -> torch.Tensor:
        s = self._sums(preds.detach(), target.detach()).to(torch.float64)
        self._state = s if self._state is None else self._state + s
        self._updates += 1
        return s

    def __call__(self, preds, target) -> torch.Tensor:
        return self._value(self.update(preds, target)).float()

    def compute(self) -> torch.Tensor:
        if self._state is None:
            raise RuntimeError(f"{type(self).__name__}.compute() before any update")
        return self._value(self._state).float()

    def sync(self, group=None):
        """DDP: sum the running state over ranks (what torchmetrics' dist_reduce_fx='sum' does)."""
        import torch.distributed as dist

        if self._state is not None and dist.is_initialized() and dist.get_world_size(group) > 1:
            dist.all_reduce(self._state, group=group)

    def state(self) -> Dict[str, float]:
        return {} if self._state is None else {k: float(v) for k, v in zip(self.fields, self._state)}


def _flat(preds, target):
    return preds.reshape(-1).to(torch.float64), target.reshape(-1).to(torch.float64)


class MeanAbsoluteError(_SumMetric):
    fields = ("abs_err", "count")

    def _sums(self, preds, target):
        p, t = _flat(preds, target)
        return torch.stack([(p - t).abs().sum(), p.new_tensor(float(p.numel()))])

    def _value(self, s):
        return s[0] / s[1]


class MeanSquaredError(_SumMetric):
    fields = ("sq_err", "count")

    def _sums(self, preds, target):
        p, t = _flat(preds, target)
        return torch.stack([((p - t) ** 2).sum(), p.new_tensor(float(p.numel()))])

    def _value(self, s):
        return s[0] / s[1]


class R2Score(_SumMetric):
    """1 - SS_res / SS_tot over all elements (single-output form, as the reference uses it on squeezed predictions)."""

    fields = ("sum_t", "sum_tt", "ss_res", "count")

    def _sums(self, preds, target):
        p, t = _flat(preds, target)
        return torch.stack([t.sum(), (t * t).sum(), ((t - p) ** 2).sum(), p.new_tensor(float(p.numel()))])

    def _value(self, s):
        ss_tot = s[1] - s[0] * s[0] / s[3]
        return 1.0 - s[2] / ss_tot


class Accuracy(_SumMetric):
    """Multiclass top-1 accuracy from logits [B, C] (or class ids [B]) against int labels [B]."""

    fields = ("correct", "count")

    def _sums(self, preds, target):
        ids = preds.argmax(-1) if preds.dim() > target.dim() else preds
        hit = (ids.reshape(-1) == target.reshape(-1)).to(torch.float64)
        return torch.stack([hit.sum(), hit.new_tensor(float(hit.numel()))])

    def _value(self, s):
        return s[0] / s[1]
