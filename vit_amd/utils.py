"""Config file loading for the path (the reference's `load_config`, src/utils.py:311-359, minus its W&B-export unwrapping:
W&B is outside the hot path): YAML -> plain dict with `$VAR` / `${VAR}` / `~` expanded in every
string value, however deeply nested."""
from __future__ import annotations

import os
from typing import Any

import yaml

__all__ = ["load_config", "expand_strings"]


def expand_strings(node: Any) -> Any:
    """A copy of a YAML tree (dicts / lists / scalars) with environment variables and the home directory expanded in
    its strings.  Keys are left alone; non-string scalars pass through."""
    if isinstance(node, str):
        return os.path.expanduser(os.path.expandvars(node))
    if isinstance(node, dict):
        return {key: expand_strings(value) for key, value in node.items()}
    if isinstance(node, (list, tuple)):
        return [expand_strings(value) for value in node]
    return node


def load_config(config_path: str) -> dict:
    with open(config_path, "r") as fh:
        tree = yaml.safe_load(fh)
    if tree is None:
        return {}
    if not isinstance(tree, dict):
        raise ValueError(f"{config_path}: expected a mapping at the top level, got {type(tree).__name__}")
    if "_wandb" in tree or isinstance((tree.get("config") or {}).get("value") if isinstance(tree.get("config"), dict) else None, dict):
        raise ValueError(f"{config_path} looks like a W&B-exported config ('_wandb' / 'config.value' wrappers); "
                         f"W&B is outside this path -- pass the plain experiment YAML")
    return expand_strings(tree)
