"""`get_model(config)`: the model factory of the path (reference: src/models/builder.py:136-197).

Plain ViT, or a ViT behind an input preprocessor chosen by `warmup.preprocessor`:
  'zca' / 'pca'  -- a fixed-at-start linear map built from covariance statistics on disk (`warmup.cov_path`),
  'attention'    -- the prefilled-attention module in the 2-D form the ViT path uses (its query projection).
Contract kept from the reference: config keys and defaults, the ValueErrors (missing cov_path, dimension mismatch,
unknown type), `config['model']['image_size']` rewritten to the preprocessor's output width, and the run-name prefix
(`ZCA{r}_fz{N|perm}[_s{10*shrinkage}][_nobias]`, `PCA{r}_fz..[_nobias]`, `Attn{r|Full}[_scaled]_fz..`), because
checkpoints and logs are named after it.
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import Callable, Dict, Optional

import torch

from .config import get_vit_config
from .preprocessor import LinearPreprocessor, PrefilledAttention, compute_pca_matrix, compute_zca_matrix, load_cov_stats
from .specvit import MyViT

__all__ = ["get_model", "get_vit_config"]


@dataclass
class _Front:
    """An input preprocessor ready to sit in front of the ViT."""

    module: torch.nn.Module
    out_dim: int
    tag: str  # run-name prefix


def _freeze_tag(warm: dict) -> str:
    epochs = warm.get("freeze_epochs", 0)
    return "_fzperm" if epochs == -1 else f"_fz{epochs}"


def _rank_tag(kind: str, r: Optional[int]) -> str:
    return kind if r is None else f"{kind}{r}"


def _linear_front(P: torch.Tensor, warm: dict, stats: dict, tag: str) -> _Front:
    """y = (x - mean) P^T written as x P^T + bias with bias = -mean P^T; `warmup.bias: false` drops the centering."""
    centred = bool(warm.get("bias", True))
    mean = stats.get("mean")
    bias = -(mean @ P.t()) if (centred and mean is not None) else None
    frozen_at_start = warm.get("freeze_epochs", 0) != 0
    return _Front(LinearPreprocessor(P, bias=bias, freeze=frozen_at_start), int(P.shape[0]),
                  tag + ("" if centred else "_nobias"))


def _zca_front(warm: dict, stats: dict) -> _Front:
    shrink = warm.get("shrinkage", 0.0)
    P = compute_zca_matrix(stats["eigvecs"], stats["eigvals"], eps=warm.get("eps", 1e-5), r=warm.get("r"), shrinkage=shrink)
    tag = _rank_tag("ZCA", warm.get("r")) + _freeze_tag(warm) + (f"_s{int(shrink * 10)}" if shrink > 0 else "")
    return _linear_front(P, warm, stats, tag)


def _pca_front(warm: dict, stats: dict) -> _Front:
    P = compute_pca_matrix(stats["eigvecs"], r=warm.get("r"))
    return _linear_front(P, warm, stats, _rank_tag("PCA", warm.get("r")) + _freeze_tag(warm))


def _attention_front(warm: dict, stats: dict) -> _Front:
    vecs, vals, r = stats["eigvecs"], stats.get("eigvals"), warm.get("r")
    scaled = bool(warm.get("scale_by_eigvals", True))
    width = int(vecs.shape[0])
    module = PrefilledAttention(input_dim=width, eigvecs=vecs, eigvals=vals, r=r, scale_by_eigvals=scaled,
                                eps=warm.get("eps", 1e-5))
    tag = f"Attn{r if r else 'Full'}" + ("_scaled" if scaled and vals is not None else "") + _freeze_tag(warm)
    return _Front(module, int(r) if r is not None else width, tag)


_FRONTS: Dict[str, Callable[[dict, dict], _Front]] = {"zca": _zca_front, "pca": _pca_front, "attention": _attention_front}


def _build_preprocessor(preproc_type: str, warmup_cfg: dict, stats: dict, initial_freeze: Optional[bool] = None):
    """(module, output width, run-name prefix) -- kept as a function for callers / tests that build fronts directly."""
    try:
        make = _FRONTS[preproc_type]
    except KeyError:
        raise ValueError(f"Unknown preprocessor type: '{preproc_type}'") from None  # builder.py:131
    front = make(warmup_cfg, stats)
    return front.module, front.out_dim, front.tag


def get_model(config):
    warm = config.get("warmup") or {}
    loss_name = (config.get("loss") or {}).get("name")
    kind = warm.get("preprocessor")
    if kind is None or str(kind).lower() in ("none", "null"):
        return MyViT(get_vit_config(config), loss_name=loss_name, model_name="ViT", full_config=config)

    kind = str(kind).lower()
    if warm.get("cov_path") is None:
        raise ValueError(f"preprocessor='{kind}' requires 'cov_path' in warmup config")  # builder.py:155
    stats = load_cov_stats(warm["cov_path"])
    width_in = int(stats["eigvecs"].shape[0])
    declared = config["model"]["image_size"]
    if width_in != declared:
        raise ValueError(f"Mismatch: eigvecs dimension {width_in} != image_size {declared}")  # builder.py:163-166
    module, width_out, tag = _build_preprocessor(kind, warm, stats)
    # the ViT is sized for what the preprocessor emits; the reference writes that back into the config too (:176-178)
    config["model"]["image_size"] = width_out
    return MyViT(get_vit_config(config), loss_name=loss_name, model_name=f"{tag}_ViT", preprocessor=module,
                 full_config=config)
