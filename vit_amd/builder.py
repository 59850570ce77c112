"""`get_model(config)`: mirrors the reference's model factory (src/models/builder.py:136-197): the vanilla ViT, or a ViT behind
a linear input preprocessor (ZCA whitening / PCA projection built from covariance statistics on disk; `warmup:` section of
the config), or behind the prefilled-attention preprocessor in the 2-D form the ViT path uses (its query projection).
"""
from __future__ import annotations

from .config import get_vit_config
from .preprocessor import LinearPreprocessor, PrefilledAttention, compute_pca_matrix, compute_zca_matrix, load_cov_stats
from .specvit import MyViT

__all__ = ["get_model", "get_vit_config"]


def _freeze_suffix(freeze_epochs: int) -> str:  # builder.py:14-27
    return "perm" if freeze_epochs == -1 else str(freeze_epochs)


def _build_preprocessor(preproc_type: str, warmup_cfg: dict, stats: dict, initial_freeze: bool):
    """(preprocessor, output_dim, name_prefix) for 'zca' / 'pca' (builder.py:45-133): same matrices, same centering bias
    `-mean @ P^T`, same run-name prefixes."""
    eigvecs = stats["eigvecs"]
    mean = stats.get("mean", None)
    r = warmup_cfg.get("r", None)
    fz = _freeze_suffix(warmup_cfg.get("freeze_epochs", 0))
    use_bias = warmup_cfg.get("bias", True)
    if preproc_type == "zca":
        eps = warmup_cfg.get("eps", 1e-5)
        shrinkage = warmup_cfg.get("shrinkage", 0.0)
        P = compute_zca_matrix(eigvecs, stats["eigvals"], eps=eps, r=r, shrinkage=shrinkage)
        rank = f"ZCA{r}" if r is not None else "ZCA"
        prefix = f"{rank}_fz{fz}" + (f"_s{int(shrinkage * 10)}" if shrinkage > 0 else "") + ("" if use_bias else "_nobias")
    elif preproc_type == "pca":
        P = compute_pca_matrix(eigvecs, r=r)
        rank = f"PCA{r}" if r is not None else "PCA"
        prefix = f"{rank}_fz{fz}" + ("" if use_bias else "_nobias")
    elif preproc_type == "attention":  # builder.py:110-129
        eigvals = stats.get("eigvals", None)
        scale = warmup_cfg.get("scale_by_eigvals", True)
        pre = PrefilledAttention(input_dim=int(eigvecs.shape[0]), eigvecs=eigvecs, eigvals=eigvals, r=r,
                                 scale_by_eigvals=scale, eps=warmup_cfg.get("eps", 1e-5))
        prefix = f"Attn{r if r else 'Full'}" + ("_scaled" if scale and eigvals is not None else "") + f"_fz{fz}"
        return pre, (r if r is not None else int(eigvecs.shape[0])), prefix
    else:
        raise ValueError(f"Unknown preprocessor type: '{preproc_type}'")  # builder.py:131
    bias = (-mean @ P.t()) if (use_bias and mean is not None) else None
    return LinearPreprocessor(P, bias=bias, freeze=initial_freeze), int(P.shape[0]), prefix


def get_model(config):
    warmup_cfg = config.get("warmup", {}) or {}
    loss_name = (config.get("loss", {}) or {}).get("name", None)
    preproc_type = warmup_cfg.get("preprocessor", None)
    if preproc_type is None or str(preproc_type).lower() in ("none", "null"):
        vit_config = get_vit_config(config)
        model = MyViT(vit_config, loss_name=loss_name, model_name="ViT", full_config=config)
        print("[builder] Created vanilla ViT model")
        return model
    cov_path = warmup_cfg.get("cov_path", None)
    if cov_path is None:
        raise ValueError(f"preprocessor='{preproc_type}' requires 'cov_path' in warmup config")  # builder.py:155
    stats = load_cov_stats(cov_path)
    input_dim = stats["eigvecs"].shape[0]
    original = config["model"]["image_size"]
    if input_dim != original:
        raise ValueError(f"Mismatch: eigvecs dimension {input_dim} != image_size {original}")  # builder.py:163-166
    freeze_epochs = warmup_cfg.get("freeze_epochs", 0)
    pre, out_dim, prefix = _build_preprocessor(str(preproc_type).lower(), warmup_cfg, stats, freeze_epochs != 0)
    if out_dim != original:
        print(f"[builder] Auto-adjusting image_size: {original} -> {out_dim}")
        config["model"]["image_size"] = out_dim  # builder.py:176-178: the ViT sees the preprocessor's output width
    vit_config = get_vit_config(config)
    model = MyViT(vit_config, loss_name=loss_name, model_name=f"{prefix}_ViT", preprocessor=pre, full_config=config)
    print(f"[builder] Created ViT model behind a {preproc_type} preprocessor ({original} -> {out_dim})")
    return model
