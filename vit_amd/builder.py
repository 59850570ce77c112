"""`get_model(config)`: mirrors the reference's model factory (src/models/builder.py:136-197) for the vanilla-ViT path.

The preprocessor variants (ZCA / PCA / prefilled attention in front of the ViT) need covariance files that are not part
of the hot path; they raise NotImplementedError here, loudly.
"""
from __future__ import annotations

from .config import get_vit_config
from .specvit import MyViT

__all__ = ["get_model", "get_vit_config"]


def get_model(config):
    warmup_cfg = config.get("warmup", {}) or {}
    loss_name = (config.get("loss", {}) or {}).get("name", None)
    preproc_type = warmup_cfg.get("preprocessor", None)
    if preproc_type is None or str(preproc_type).lower() in ("none", "null"):
        vit_config = get_vit_config(config)
        model = MyViT(vit_config, loss_name=loss_name, model_name="ViT", full_config=config)
        print("[builder] Created vanilla ViT model")
        return model
    if warmup_cfg.get("cov_path", None) is None:
        raise ValueError(f"preprocessor='{preproc_type}' requires 'cov_path' in warmup config")  # builder.py:155
    raise NotImplementedError(
        f"preprocessor='{preproc_type}' (ZCA/PCA/attention front-end) is outside the MI355X hot path built so far "
        "(SURVEY.md section 8f item 4)")
