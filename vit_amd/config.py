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


def get_vit_config(config) -> ViTConfig:
    """Build the model config from the YAML dict; same rules as the reference (builder.py:200-258): for regression
    `num_labels` is ALWAYS derived from data.param (and written back into config['model']), with the same warning."""
    m = config["model"]
    d = config.get("data", {}) or {}
    task = (m.get("task_type") or m.get("task") or "cls").lower()
    if task in ("reg", "regression"):
        p = d.get("param", None)
        num_labels = 1
        if isinstance(p, str) and len(p) > 0:
            plist = [x.strip() for x in p.split(",") if x.strip()]
            if len(plist) >= 1:
                num_labels = len(plist)
        elif isinstance(p, (list, tuple)) and len(p) > 0:
            num_labels = len(p)
        config_num_labels = m.get("num_labels")
        if config_num_labels is not None and int(config_num_labels) != num_labels:
            print(f"Warning: model.num_labels={config_num_labels} conflicts with data.param (which implies "
                  f"{num_labels} labels). Using {num_labels} from data.param.")
        m["num_labels"] = num_labels
    else:
        num_labels = int(m.get("num_labels", 1) or 1)
    return ViTConfig(
        task_type=m["task_type"],
        image_size=m["image_size"],
        patch_size=m["patch_size"],
        hidden_size=m["hidden_size"],
        num_hidden_layers=m["num_hidden_layers"],
        num_attention_heads=m["num_attention_heads"],
        stride_ratio=m.get("stride_ratio", 1),
        stride_size=m.get("stride_size", None),
        proj_fn=m["proj_fn"],
        num_labels=num_labels,
        pos_encoding_type=m.get("pos_encoding_type", None),
        max_position_embeddings=m.get("max_position_embeddings", 512),
        rope_base=m.get("rope_base", 10000.0),
    )
