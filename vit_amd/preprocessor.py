"""Linear input preprocessors in front of the ViT: ZCA whitening / PCA projection.

Mirrors the reference's surface -- `compute_zca_matrix`, `compute_pca_matrix`, `LinearPreprocessor` with `freeze()`
(src/models/preprocessor.py:12-111, src/models/layers.py:11-63) -- with the arithmetic of the layer itself on the MI355X
path: forward `x @ P^T + bias` is one vit_gemm, backward (when the layer is trainable) `dP = dy^T x` is the split-K dW GEMM
and `dbias` the column-sum kernel.  The two matrix builders are set-up time linear algebra on the host (they run once,
from covariance statistics on disk), like weight initialisation.

The whitening matrix follows the reference's definitions:
  full rank:  P = V diag(1 / sqrt(lam_hat + eps)) V^T,  lam_hat = (1 - s) lam + s mean(lam)      (shrinkage s)
  rank r:     P = V_r diag(1 / sqrt(lam_hat_r + eps)) V_r^T + s_perp (I - V_r V_r^T),
              s_perp = 1 / sqrt(max(median(lam_hat[r:]), 1e-3 mean(lam_hat[:r])) + eps)
  PCA:        P = V[:, :r]^T  (all columns when r is None)
"""
from __future__ import annotations

from typing import Optional

import torch
import torch.nn as nn

__all__ = ["LinearPreprocessor", "PrefilledAttention", "compute_zca_matrix", "compute_pca_matrix", "load_cov_stats"]


def _shrunk(eigvals: torch.Tensor, shrinkage: float) -> torch.Tensor:
    if shrinkage > 0.0:
        return (1.0 - shrinkage) * eigvals + shrinkage * eigvals.mean()
    return eigvals


def compute_zca_matrix(eigvecs: torch.Tensor, eigvals: torch.Tensor, eps: float = 1e-5, r: Optional[int] = None,
                       shrinkage: float = 0.1) -> torch.Tensor:
    """ZCA whitening matrix [D, D] from eigenvectors (columns, eigenvalues descending); preprocessor.py:12-77."""
    lam = _shrunk(eigvals, shrinkage)
    if r is None:
        return eigvecs @ torch.diag(1.0 / torch.sqrt(lam + eps)) @ eigvecs.t()
    vr = eigvecs[:, :r]
    inv_sqrt_r = torch.rsqrt(lam[:r] + eps)
    tail = lam[r:]
    lam0 = tail.median() if tail.numel() > 0 else lam[r - 1]
    lam0 = torch.clamp(lam0, min=1e-3 * lam[:r].mean())
    s_perp = 1.0 / torch.sqrt(lam0 + eps)
    dim = eigvecs.shape[0]
    eye = torch.eye(dim, dtype=eigvecs.dtype, device=eigvecs.device)
    return (vr * inv_sqrt_r) @ vr.t() + s_perp * (eye - vr @ vr.t())


def compute_pca_matrix(eigvecs: torch.Tensor, r: Optional[int] = None) -> torch.Tensor:
    """PCA projection [r, D] (or [D, D]); preprocessor.py:80-93."""
    return eigvecs.t() if r is None else eigvecs[:, :r].t()


_COV_CACHE: dict = {}


def load_cov_stats(cov_path) -> dict:
    """Covariance statistics file -> dict with at least 'eigvecs' (and 'eigvals', 'mean' where used); cached per path like
    the reference's loader (src/utils.py:17-...).  Loaded with weights_only=True (tensors only) or as .npz."""
    import os

    path = os.path.abspath(str(cov_path))
    if path in _COV_CACHE:
        return _COV_CACHE[path]
    if not os.path.exists(path):
        raise FileNotFoundError(f"Covariance file not found: {path}")
    if path.endswith(".npz"):
        import numpy as np

        raw = np.load(path)
        stats = {k: torch.from_numpy(raw[k]) for k in raw.files}
    else:
        stats = torch.load(path, map_location="cpu", weights_only=True)
    if "eigvecs" not in stats:
        raise ValueError(f"{path}: covariance statistics need an 'eigvecs' entry")
    _COV_CACHE[path] = stats
    return stats


class _LinearFn(torch.autograd.Function):
    """y = x P^T + b through the C ABI; backward produces dP and db only (x is the network input)."""

    @staticmethod
    def forward(ctx, x, weight, bias, op_dtype):
        from . import functional as vf

        B, L = x.shape
        r = weight.shape[0]
        x, weight = x.contiguous(), weight.contiguous()
        if op_dtype == torch.bfloat16:  # the operand casts go through the library too (vit_cast_f32_bf16)
            xo, wo = vf.cast_f32_bf16(x), vf.cast_f32_bf16(weight)
        else:
            xo, wo = x, weight
        y = vf.gemm(xo, wo, M=B, N=r, K=L, bias=bias, out_dtype=torch.float32)
        ctx.save_for_backward(xo)
        ctx.shape = (B, L, r)
        ctx.has_bias = bias is not None
        ctx.need_w = bool(ctx.needs_input_grad[1])
        return y

    @staticmethod
    def backward(ctx, dy):
        from . import functional as vf

        (xo,) = ctx.saved_tensors
        B, L, r = ctx.shape
        dw = db = None
        if ctx.need_w:
            dyo = dy.contiguous()
            if xo.dtype == torch.bfloat16:
                dyo = vf.cast_f32_bf16(dyo.float())
            dw = vf.gemm(dyo, xo, M=r, N=L, K=B, a_trans=True, b_trans=True, out_dtype=torch.float32, split_k=-1)
            if ctx.has_bias:
                db = vf.colsum(dy.contiguous().float())
        return None, dw, db, None


class LinearPreprocessor(nn.Module):
    """x -> x P^T + bias (ZCA: P [D, D]; PCA: P [r, D]); frozen = buffers, trainable = Parameters, `freeze()` converts
    between the two exactly as the reference's PrefilledLinear does (layers.py:17-60), so `parameters()` and
    `state_dict()` look the same from outside (keys `linear.weight`, `linear.bias`)."""

    class _Holder(nn.Module):
        pass

    def __init__(self, matrix: torch.Tensor, bias: Optional[torch.Tensor] = None, freeze: bool = True) -> None:
        super().__init__()
        self.linear = LinearPreprocessor._Holder()
        self._is_frozen = None
        self._set(matrix.to(torch.float32), None if bias is None else bias.to(torch.float32), freeze)
        self.op_dtype = torch.float32  # 'bf16-mixed': torch.bfloat16 (what F.linear computes under autocast)

    def _set(self, weight, bias, freeze):
        lin = self.linear
        for name in ("weight", "bias"):
            if name in lin._parameters:
                del lin._parameters[name]
            if name in lin._buffers:
                del lin._buffers[name]
        if freeze:
            lin.register_buffer("weight", weight)
            lin.register_buffer("bias", bias)
        else:
            lin.weight = nn.Parameter(weight)
            if bias is not None:
                lin.bias = nn.Parameter(bias)
            else:
                lin.register_buffer("bias", None)
        self._is_frozen = bool(freeze)

    @property
    def out_features(self) -> int:
        return int(self.linear.weight.shape[0])

    def freeze(self, freeze: bool = True) -> None:
        if bool(freeze) == self._is_frozen:
            return
        w = self.linear.weight.detach().clone()
        b = None if self.linear.bias is None else self.linear.bias.detach().clone()
        self._set(w, b, freeze)

    def set_precision(self, precision) -> None:
        self.op_dtype = torch.bfloat16 if str(precision).lower() in ("bf16-mixed", "bf16", "16-mixed") else torch.float32

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        if not x.is_cuda:
            from ._cabi import VitError

            raise VitError("LinearPreprocessor: the input must live on the GPU (there is no CPU path)")
        return _LinearFn.apply(x.to(torch.float32), self.linear.weight, self.linear.bias, self.op_dtype)


class PrefilledAttention(nn.Module):
    """`warmup.preprocessor: attention` (src/models/attention.py:12-121): query / key projections prefilled with the
    (optionally 1/sqrt(eigenvalue)-scaled) leading eigenvectors.  The ViT hands it 2-D spectra, for which the reference's
    forward IS the query projection (attention.py:81-84: `if x.dim() == 2: return self.q_lin(x)`) -- one vit_gemm here;
    the key / value projections exist only as parameters (same state_dict keys: q_lin / k_lin / v_lin .weight), and the 3-D
    self-attention branch, which the ViT path never reaches, is not built."""

    class _Lin(nn.Module):
        def __init__(self, weight):
            super().__init__()
            self.weight = nn.Parameter(weight)

    def __init__(self, input_dim: int, eigvecs: torch.Tensor, eigvals: Optional[torch.Tensor] = None, r: Optional[int] = None,
                 low_rank: Optional[bool] = None, scale_by_eigvals: bool = True, eps: float = 1e-5) -> None:
        super().__init__()
        import math

        self.input_dim = input_dim
        self.r = r if r is not None else eigvecs.shape[1]
        self.low_rank = low_rank if low_rank is not None else (self.r < input_dim)
        self.scale_by_eigvals = scale_by_eigvals and eigvals is not None
        basis = eigvecs[:, : self.r].t().contiguous().to(torch.float32)            # [r, D]
        if self.scale_by_eigvals:
            basis = basis * torch.rsqrt(eigvals[: self.r].to(torch.float32) + eps).unsqueeze(1)
        if self.low_rank:
            wq = basis.clone()
        else:
            wq = torch.zeros(input_dim, input_dim)
            wq[: basis.shape[0], :] = basis
        self.q_lin = PrefilledAttention._Lin(wq)
        self.k_lin = PrefilledAttention._Lin(wq.clone())
        wv = torch.empty(input_dim, input_dim)
        nn.init.kaiming_uniform_(wv, a=math.sqrt(5))
        self.v_lin = PrefilledAttention._Lin(wv)
        self.op_dtype = torch.float32

    @property
    def out_features(self) -> int:
        return int(self.q_lin.weight.shape[0])

    def set_qk_trainable(self, trainable: bool = True) -> None:  # attention.py:98-103
        self.q_lin.weight.requires_grad = trainable
        self.k_lin.weight.requires_grad = trainable

    def set_precision(self, precision) -> None:
        self.op_dtype = torch.bfloat16 if str(precision).lower() in ("bf16-mixed", "bf16", "16-mixed") else torch.float32

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        if x.dim() != 2:
            raise NotImplementedError("PrefilledAttention: only the 2-D (spectra) form the ViT path uses is built")
        if not x.is_cuda:
            from ._cabi import VitError

            raise VitError("PrefilledAttention: the input must live on the GPU (there is no CPU path)")
        return _LinearFn.apply(x.to(torch.float32), self.q_lin.weight, None, self.op_dtype)
