"""Config mapping of the hot path: mirrors `get_vit_config` (reference src/models/builder.py:200-258).

The reference returns a HuggingFace `ViTConfig`; the fields below are the ones the path reads, with the values the
reference hard-codes (num_channels=1, intermediate=4*hidden, erf-GELU, dropout 0.1/0.1, layer_norm_eps=1e-12,
qkv_bias=True).  Attribute names match ViTConfig so code written against the reference's config object keeps working.
"""
from __future__ import annotations

import math
from dataclasses import dataclass
from typing import Optional

__all__ = ["ViTConfig", "get_vit_config"]


@dataclass
class ViTConfig:
    task_type: str
    image_size: int
    patch_size: int
    hidden_size: int
    num_hidden_layers: int
    num_attention_heads: int
    proj_fn: str = "SW"
    stride_ratio: float = 1
    stride_size: Optional[int] = None
    num_labels: int = 1
    num_channels: int = 1
    hidden_act: str = "gelu"
    hidden_dropout_prob: float = 0.1
    attention_probs_dropout_prob: float = 0.1
    initializer_range: float = 0.02
    layer_norm_eps: float = 1e-12
    qkv_bias: bool = True
    pos_encoding_type: Optional[str] = None
    max_position_embeddings: int = 512
    rope_base: float = 10000.0
    use_return_dict: bool = True

    @property
    def intermediate_size(self) -> int:  # builder.py:243
        return 4 * self.hidden_size

    @property
    def stride(self) -> int:  # embedding.py:25-26
        s = self.stride_size
        return int(s) if s and s > 0 else int(self.stride_ratio * self.patch_size)

    @property
    def num_patches(self) -> int:
        L, P, S = self.image_size, self.patch_size, self.stride
        if self.proj_fn == "SW":  # tokenization.py:40
            return math.ceil((L - P) / S) + 1
        if self.proj_fn in ("C1D", "CNN"):  # tokenization.py:65
            return (L - P) // S + 1
        raise ValueError(f"Unsupported proj_fn '{self.proj_fn}'")  # embedding.py:44

    @property
    def seq_len(self) -> int:
        return self.num_patches + 1

    @property
    def head_dim(self) -> int:
        return self.hidden_size // self.num_attention_heads


def _targets_in(param) -> int:
    """How many regression targets `data.param` names: a comma-separated string or a list; anything else (None, '', []) = 1."""
    if isinstance(param, str):
        names = [tok for tok in (piece.strip() for piece in param.split(",")) if tok]
        return len(names) or 1
    if isinstance(param, (list, tuple)):
        return len(param) or 1
    return 1


def get_vit_config(config) -> ViTConfig:
    """The model config from the YAML dict (reference: src/models/builder.py:200-258, pinned on it by
    tests/golden/config.json).  Regression: `num_labels` ALWAYS follows `data.param` -- a conflicting `model.num_labels` is
    overridden with a warning -- and is written back into config['model']; classification takes `model.num_labels`."""
    import warnings

    model = config["model"]
    task = str(model.get("task_type") or model.get("task") or "cls").lower()
    if task in ("reg", "regression"):
        num_labels = _targets_in((config.get("data") or {}).get("param"))
        stated = model.get("num_labels")
        if stated is not None and int(stated) != num_labels:
            warnings.warn(f"model.num_labels={stated} disagrees with data.param, which names {num_labels} target(s); "
                          f"using {num_labels}", stacklevel=2)
        model["num_labels"] = num_labels
    else:
        num_labels = int(model.get("num_labels", 1) or 1)
    optional = {k: model[k] for k in ("stride_ratio", "stride_size", "pos_encoding_type", "max_position_embeddings", "rope_base")
                if k in model}
    return ViTConfig(
        task_type=model["task_type"], image_size=model["image_size"], patch_size=model["patch_size"],
        hidden_size=model["hidden_size"], num_hidden_layers=model["num_hidden_layers"],
        num_attention_heads=model["num_attention_heads"], proj_fn=model["proj_fn"], num_labels=num_labels, **optional)
