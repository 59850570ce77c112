"""Config plumbing of the path: `load_config` with the reference's semantics (src/utils.py:311-359): YAML + expansion
of ${ENV} / ~ in every string + transparent flattening of W&B-exported configs."""
from __future__ import annotations

import os

import yaml

__all__ = ["load_config"]


def _expand(item):
    if isinstance(item, str):
        return os.path.expanduser(os.path.expandvars(item))
    if isinstance(item, dict):
        return {k: _expand(v) for k, v in item.items()}
    if isinstance(item, list):
        return [_expand(v) for v in item]
    return item


def _maybe_flatten_wandb_cfg(cfg):
    if not isinstance(cfg, dict):
        return cfg
    inner = cfg.get("config")
    if isinstance(inner, dict) and isinstance(inner.get("value"), dict):
        return inner["value"]
    flattened, saw = {}, False
    for k, v in cfg.items():
        if isinstance(v, dict) and "value" in v and isinstance(v["value"], (dict, list, str, int, float, bool, type(None))):
            flattened[k] = v["value"]
            saw = True
        elif k != "_wandb":
            flattened[k] = v
    return flattened if saw else cfg


def load_config(config_path):
    with open(config_path, "r") as f:
        raw = yaml.safe_load(f)
    return _expand(_maybe_flatten_wandb_cfg(raw))
