"""Config file loading for the path (the reference's `load_config`, src/utils.py:311-359): YAML -> plain dict with `$VAR` /
`${VAR}` / `~` expanded in every string value, however deeply nested, and the two W&B-export shapes the reference accepts
unwrapped first (r04; pinned on the reference's own function: tests/golden/wandbcfg.json):
  * the experiment config nested under `config: {value: {...}}` -> that inner mapping, everything beside it dropped;
  * every top-level key wrapped as `{value: ...}` (W&B's per-key export) -> the bare values; keys without a wrapper pass
    through, except a bare `_wandb` metadata section, which is dropped (a WRAPPED `_wandb` is unwrapped and kept, as the
    reference's code does whatever its docstring says).
W&B itself (logging, sweeps) stays outside the hot path."""
from __future__ import annotations

import os
from typing import Any

import yaml

__all__ = ["load_config", "expand_strings", "unwrap_wandb_export"]


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


_SCALARS = (dict, list, str, int, float, bool, type(None))


def unwrap_wandb_export(tree: Any) -> Any:
    """The two W&B-export shapes of src/utils.py:330-355, else the tree itself."""
    if not isinstance(tree, dict):
        return tree
    nested = tree.get("config")
    if isinstance(nested, dict) and isinstance(nested.get("value"), dict):
        return nested["value"]
    wrapped = {k for k, v in tree.items() if isinstance(v, dict) and "value" in v and isinstance(v["value"], _SCALARS)}
    if not wrapped:
        return tree
    return {k: (v["value"] if k in wrapped else v) for k, v in tree.items() if k in wrapped or k != "_wandb"}


def load_config(config_path: str) -> dict:
    with open(config_path, "r") as fh:
        tree = yaml.safe_load(fh)
    if tree is None:
        return {}
    if not isinstance(tree, dict):
        raise ValueError(f"{config_path}: expected a mapping at the top level, got {type(tree).__name__}")
    return expand_strings(unwrap_wandb_export(tree))
