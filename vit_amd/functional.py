"""Tensor-level wrappers over the C ABI (include/vit_amd.h).  torch supplies device memory and the current HIP stream;
every computation happens in libvit_amd.so.  No function here has a PyTorch / CPU fallback."""
from __future__ import annotations

import ctypes as C
from typing import Optional, Tuple

import torch

from . import _cabi
from ._cabi import ACT_DGELU, ACT_GELU, ACT_GELU_GRAD, ACT_MUL_AUX, ACT_NONE, LOSS_CE, LOSS_L1, LOSS_MSE, VIT_BF16, VIT_F32, GemmDesc, check

_DT = {torch.float32: VIT_F32, torch.bfloat16: VIT_BF16}


def _stream(t: torch.Tensor) -> int:
    return torch.cuda.current_stream(t.device).cuda_stream


_HANDLE_OVERRIDE: Optional[_cabi.Handle] = None


def _h(t: torch.Tensor) -> _cabi.Handle:
    if not t.is_cuda:
        raise _cabi.VitError("vit_amd kernels need tensors on an MI355X (cuda/hip device); there is no CPU path")
    return _HANDLE_OVERRIDE if _HANDLE_OVERRIDE is not None else _cabi.handle_for(t.device)


class use_handle:
    """Route the calls inside the block through another vit_handle (its own workspace): what runs concurrently on a second
    HIP stream must not share split-K slabs / reduction partials with the main stream's kernels."""

    def __init__(self, handle: _cabi.Handle):
        self.handle = handle

    def __enter__(self):
        global _HANDLE_OVERRIDE
        self.prev, _HANDLE_OVERRIDE = _HANDLE_OVERRIDE, self.handle
        return self.handle

    def __exit__(self, *exc):
        global _HANDLE_OVERRIDE
        _HANDLE_OVERRIDE = self.prev
        return False


def _ptr(t: Optional[torch.Tensor]) -> Optional[int]:
    return None if t is None else t.data_ptr()


def _chk(t: torch.Tensor, dtype, name: str):
    if t.dtype != dtype:
        raise _cabi.VitError(f"{name}: expected {dtype}, got {t.dtype}")
    if not t.is_contiguous():
        raise _cabi.VitError(f"{name}: tensor must be contiguous")


Dropout = Tuple[float, int, int]  # (p, seed, site)
NO_DROP: Dropout = (0.0, 0, 0)


# ------------------------------------------------------------------------------------------------ GEMM
def gemm(a: torch.Tensor, b: torch.Tensor, *, M: int, N: int, K: int, a_trans: bool = False, b_trans: bool = False,
         lda: Optional[int] = None, ldb: Optional[int] = None, out: Optional[torch.Tensor] = None,
         out_dtype=torch.bfloat16, ldc: Optional[int] = None, alpha: float = 1.0, bias: Optional[torch.Tensor] = None,
         act: int = ACT_NONE, aux_out: Optional[torch.Tensor] = None, aux_in: Optional[torch.Tensor] = None,
         dropout: Dropout = NO_DROP, residual: Optional[torch.Tensor] = None, row_map: Tuple[int, int, int] = (0, 0, 0),
         out_rows: Optional[int] = None, split_k: int = 0, accumulate: bool = False,
         colsum_out: Optional[torch.Tensor] = None,
         rope: Optional[Tuple[torch.Tensor, torch.Tensor, int, int, int]] = None) -> torch.Tensor:
    """`rope = (cos, sin, T, head_dim, cols)`: rotary embedding of columns [0, cols) of the output (the q / k thirds of a fused QKV
    projection; cos / sin: f32 [T, head_dim / 2], rows = the reference's LINEAR zero-based positions t * theta_i -- the rotating
    epilogue steps angles by a recurrence over table row 8; other tables: `rope_qk`) -- in the GEMM's epilogue where the kernel can, by a vit_rope_qk pass behind it
    otherwise (the library decides; same result contract)."""
    h = _h(a)
    if a.dtype != b.dtype or a.dtype not in _DT:
        raise _cabi.VitError(f"gemm: operands must both be bf16 or both f32 (got {a.dtype}, {b.dtype})")
    if out is None:
        out = torch.empty((out_rows if out_rows is not None else M, N), dtype=out_dtype, device=a.device)
    d = GemmDesc()
    d.M, d.N, d.K = M, N, K
    d.a_trans, d.b_trans = int(a_trans), int(b_trans)
    d.ab_dtype = _DT[a.dtype]  # f32 operands: split-bf16 "x3" kernel (fp32-class results)
    d.A, d.lda = a.data_ptr(), lda if lda is not None else (M if a_trans else K)
    d.B, d.ldb = b.data_ptr(), ldb if ldb is not None else (N if b_trans else K)
    d.C, d.ldc, d.c_dtype = out.data_ptr(), ldc if ldc is not None else N, _DT[out.dtype]
    d.alpha = alpha
    d.bias = _ptr(bias)
    d.act = act
    d.aux_out, d.aux_in, d.ldaux = _ptr(aux_out), _ptr(aux_in), N
    d.dropout_p, d.seed, d.site = dropout
    d.residual, d.ldres = _ptr(residual), N
    d.rows_per_batch, d.out_batch_rows, d.out_row_offset = row_map
    d.split_k = split_k
    d.accumulate = int(accumulate)
    if colsum_out is not None:  # column sums of C (a bias gradient): fused into the epilogue or a vit_colsum pass
        _chk(colsum_out, torch.float32, "gemm colsum_out")
        if colsum_out.numel() != N:
            raise _cabi.VitError(f"gemm: colsum_out must have N={N} elements")
        d.colsum_out = colsum_out.data_ptr()
        h.ensure_workspace(max(2 * (-(-M // 256)), 2048) * N * 4)
    if rope is not None:
        cos, sin, rT, rdh, rcols = rope
        _chk(cos, torch.float32, "gemm rope cos")
        _chk(sin, torch.float32, "gemm rope sin")
        if cos.numel() != rT * (rdh // 2) or sin.numel() != cos.numel():
            raise _cabi.VitError(f"gemm: rope tables must be [T={rT}, head_dim/2={rdh // 2}]")
        d.rope_cos, d.rope_sin, d.rope_T, d.rope_dh, d.rope_cols = cos.data_ptr(), sin.data_ptr(), int(rT), int(rdh), int(rcols)
    if split_k != 0 and split_k != 1:
        if split_k < 0:  # upper bound over the automatic choices of both GEMM cores (gemm.hip / gemm2.hip)
            tiles = -(-M // 128) * -(-N // 128)
            ktiles = -(-K // 64)
            split_k = min(max(1, 512 // tiles), max(1, ktiles // 4)) if tiles < 256 else 1
            t2 = max(1, (M // 256) * max(1, N // 256))
            split_k = max(split_k, min(max(1, 256 // t2), max(1, ktiles // 8)) + 1)
        h.ensure_workspace(split_k * M * N * 4)
        split_k = d.split_k
    check(h.lib.vit_gemm(h.h, C.byref(d), _stream(a)), "vit_gemm")
    return out


def linear_fwd(x, W, bias=None, *, out_dtype=torch.bfloat16, act=ACT_NONE, aux_out=None, dropout: Dropout = NO_DROP,
               residual=None, out=None):
    M, K = x.shape
    N = W.shape[0]
    return gemm(x, W, M=M, N=N, K=K, out=out, out_dtype=out_dtype, bias=bias, act=act, aux_out=aux_out, dropout=dropout,
                residual=residual)


def linear_bwd_dx(dy, W, *, out_dtype=torch.bfloat16, dgelu_aux=None, out=None):
    M, N = dy.shape
    K = W.shape[1]
    return gemm(dy, W, M=M, N=K, K=N, b_trans=True, out=out, out_dtype=out_dtype,
                act=ACT_DGELU if dgelu_aux is not None else ACT_NONE, aux_in=dgelu_aux)


def linear_bwd_dw(dy, x, *, out=None, accumulate=False):
    M, N = dy.shape
    K = x.shape[1]
    return gemm(dy, x, M=N, N=K, K=M, a_trans=True, b_trans=True, out=out, out_dtype=torch.float32, split_k=-1,
                accumulate=accumulate)


# ------------------------------------------------------------------------------------------------ LayerNorm
def layernorm_fwd(x, gamma, beta, eps: float, out_dtype=torch.bfloat16, out=None, mean=None, rstd=None):
    _chk(x, torch.float32, "layernorm_fwd x")
    h = _h(x)
    D = x.shape[-1]
    rows = x.numel() // D
    y = out if out is not None else torch.empty(x.shape, dtype=out_dtype, device=x.device)
    mean = mean if mean is not None else torch.empty(rows, dtype=torch.float32, device=x.device)
    rstd = rstd if rstd is not None else torch.empty(rows, dtype=torch.float32, device=x.device)
    check(h.lib.vit_layernorm_fwd(h.h, x.data_ptr(), gamma.data_ptr(), beta.data_ptr(), y.data_ptr(), _DT[y.dtype],
                                  _ptr(mean), _ptr(rstd), rows, D, eps, _stream(x)), "vit_layernorm_fwd")
    return y, mean, rstd


def layernorm_fwd_residual(x, delta, xsum, gamma, beta, eps: float, out_dtype=torch.bfloat16, out=None, mean=None,
                           rstd=None):
    """xsum = x + delta (f32 stream + bf16/f32 projection output); y = LayerNorm(xsum)."""
    _chk(x, torch.float32, "layernorm_fwd_residual x")
    _chk(xsum, torch.float32, "layernorm_fwd_residual xsum")
    if delta.dtype not in _DT or not delta.is_contiguous() or delta.numel() != x.numel() or xsum.numel() != x.numel():
        raise _cabi.VitError("layernorm_fwd_residual: delta must be a contiguous bf16/f32 tensor of x's shape")
    h = _h(x)
    D = x.shape[-1]
    rows = x.numel() // D
    y = out if out is not None else torch.empty(x.shape, dtype=out_dtype, device=x.device)
    mean = mean if mean is not None else torch.empty(rows, dtype=torch.float32, device=x.device)
    rstd = rstd if rstd is not None else torch.empty(rows, dtype=torch.float32, device=x.device)
    check(h.lib.vit_layernorm_fwd_residual(h.h, x.data_ptr(), delta.data_ptr(), _DT[delta.dtype], xsum.data_ptr(),
                                           gamma.data_ptr(), beta.data_ptr(), y.data_ptr(), _DT[y.dtype], _ptr(mean),
                                           _ptr(rstd), rows, D, eps, _stream(x)), "vit_layernorm_fwd_residual")
    return y, mean, rstd


def layernorm_bwd(dy, x, gamma, mean, rstd, dres=None, dx=None, dgamma=None, dbeta=None):
    _chk(x, torch.float32, "layernorm_bwd x")
    h = _h(x)
    D = x.shape[-1]
    rows = x.numel() // D
    dx = dx if dx is not None else torch.empty_like(x)
    dgamma = dgamma if dgamma is not None else torch.empty(D, dtype=torch.float32, device=x.device)
    dbeta = dbeta if dbeta is not None else torch.empty(D, dtype=torch.float32, device=x.device)
    check(h.lib.vit_layernorm_bwd(h.h, dy.data_ptr(), _DT[dy.dtype], x.data_ptr(), gamma.data_ptr(), mean.data_ptr(),
                                  rstd.data_ptr(), _ptr(dres), dx.data_ptr(), dgamma.data_ptr(), dbeta.data_ptr(),
                                  rows, D, _stream(x)), "vit_layernorm_bwd")
    return dx, dgamma, dbeta


def layernorm_bwd_fused(dy, x, gamma, mean, rstd, dres, dx, dgamma, dbeta, dyn, dbias, dropout: Dropout = NO_DROP):
    """LayerNorm backward that also emits dyn = mask * dx (bf16) and dbias = colsum(dyn)."""
    _chk(x, torch.float32, "layernorm_bwd_fused x")
    h = _h(x)
    D = x.shape[-1]
    rows = x.numel() // D
    p, seed, site = dropout
    check(h.lib.vit_layernorm_bwd_fused(h.h, dy.data_ptr(), _DT[dy.dtype], x.data_ptr(), gamma.data_ptr(),
                                        mean.data_ptr(), rstd.data_ptr(), _ptr(dres), dx.data_ptr(), dgamma.data_ptr(),
                                        dbeta.data_ptr(), rows, D, dyn.data_ptr(), _DT[dyn.dtype], dbias.data_ptr(), p, seed,
                                        site, _stream(x)), "vit_layernorm_bwd_fused")
    return dx, dgamma, dbeta, dyn, dbias


# ------------------------------------------------------------------------------------------------ attention
def attention_fwd(qkv, B: int, H: int, T: int, dh: int, scale: float, dropout: Dropout = NO_DROP, ctx=None, lse=None,
                  ctx_lo=None):
    """ctx_lo (bf16, like ctx): also store the rounding residual of the context for the backward's delta (vit_amd.h)."""
    h = _h(qkv)
    ctx = ctx if ctx is not None else torch.empty((B * T, H * dh), dtype=qkv.dtype, device=qkv.device)
    lse = lse if lse is not None else torch.empty((B * H, T), dtype=torch.float32, device=qkv.device)
    p, seed, site = dropout
    if ctx_lo is not None:
        _chk(ctx_lo, ctx.dtype, "attention_fwd ctx_lo")
    check(h.lib.vit_attention_fwd_lo(h.h, qkv.data_ptr(), ctx.data_ptr(), _ptr(ctx_lo), lse.data_ptr(), _DT[qkv.dtype], B, H,
                                     T, dh, scale, p, seed, site, _stream(qkv)), "vit_attention_fwd")
    return ctx, lse


def attention_bwd(qkv, ctx, dctx, lse, B: int, H: int, T: int, dh: int, scale: float, dropout: Dropout = NO_DROP,
                  dqkv=None, delta=None, colsum_out=None, ctx_lo=None):
    """colsum_out (f32 [3*H*dh]): also the column sums of dqkv (the QKV projection's bias gradient).
    ctx_lo: the residual attention_fwd(ctx_lo=...) stored."""
    _chk(dctx, qkv.dtype, "attention_bwd dctx")
    h = _h(qkv)
    dqkv = dqkv if dqkv is not None else torch.empty_like(qkv)
    delta = delta if delta is not None else torch.empty((B * H, T), dtype=torch.float32, device=qkv.device)
    p, seed, site = dropout
    if colsum_out is not None:
        _chk(colsum_out, torch.float32, "attention_bwd colsum_out")
        h.ensure_workspace(B * 16 * 3 * H * dh * 4)
    check(h.lib.vit_attention_bwd_lo(h.h, qkv.data_ptr(), ctx.data_ptr(), _ptr(ctx_lo), dctx.data_ptr(), lse.data_ptr(),
                                     delta.data_ptr(), dqkv.data_ptr(), _DT[qkv.dtype], B, H, T, dh, scale, p, seed, site,
                                     _ptr(colsum_out), _stream(qkv)), "vit_attention_bwd")
    return dqkv


def attention_probs(qkv, B: int, H: int, T: int, dh: int, scale: float):
    h = _h(qkv)
    probs = torch.empty((B, H, T, T), dtype=torch.float32, device=qkv.device)
    check(h.lib.vit_attention_probs(h.h, qkv.data_ptr(), probs.data_ptr(), _DT[qkv.dtype], B, H, T, dh, scale,
                                    _stream(qkv)), "vit_attention_probs")
    return probs


# ------------------------------------------------------------------------------------------------ embedding side
def fold_add(dpatches, B: int, L: int, P: int, S: int, N: int, out=None):
    """dx[B, L] = overlap-add of dpatches [B*N, P] (f32): the backward of unfold_cast wrt the signal."""
    _chk(dpatches, torch.float32, "fold_add dpatches")
    if dpatches.numel() != B * N * P:
        raise _cabi.VitError("fold_add: dpatches must hold B*N*P elements")
    out = out if out is not None else torch.empty((B, L), dtype=torch.float32, device=dpatches.device)
    h = _h(dpatches)
    check(h.lib.vit_fold_add(h.h, dpatches.data_ptr(), out.data_ptr(), B, L, P, S, N, _stream(dpatches)), "vit_fold_add")
    return out


def add_noise(flux, error, noise_level: float, seed: int, out=None):
    """flux + N(0,1) * error * noise_level (vit.py:86-88), counter-based normals keyed on (seed, element index)."""
    _chk(flux, torch.float32, "add_noise flux")
    _chk(error, torch.float32, "add_noise error")
    if error.shape != flux.shape or flux.numel() % 4:
        raise _cabi.VitError("add_noise: flux/error must have the same shape and a multiple of 4 elements")
    out = out if out is not None else torch.empty_like(flux)
    h = _h(flux)
    check(h.lib.vit_add_noise(h.h, flux.data_ptr(), error.data_ptr(), out.data_ptr(), flux.numel(), float(noise_level),
                              int(seed) & 0xFFFFFFFFFFFFFFFF, _stream(flux)), "vit_add_noise")
    return out


def rope_qk(qkv, cos_half, sin_half, T: int, H: int, dh: int, inverse: bool = False):
    """In-place rotary embedding of the q and k thirds of qkv [B*T, 3*H*dh] (bf16 or f32); cos/sin f32 [>=T, dh/2]."""
    if qkv.dtype not in _DT or not qkv.is_contiguous() or qkv.dim() != 2 or qkv.shape[1] != 3 * H * dh:
        raise _cabi.VitError(f"rope_qk: qkv must be a contiguous [rows, {3 * H * dh}] bf16/f32 tensor")
    _chk(cos_half, torch.float32, "rope_qk cos")
    _chk(sin_half, torch.float32, "rope_qk sin")
    if cos_half.shape[-1] != dh // 2 or cos_half.shape[0] < T or sin_half.shape != cos_half.shape:
        raise _cabi.VitError("rope_qk: tables must be [>= T, dh/2]")
    h = _h(qkv)
    check(h.lib.vit_rope_qk(h.h, qkv.data_ptr(), _DT[qkv.dtype], cos_half.data_ptr(), sin_half.data_ptr(), qkv.shape[0], T,
                            H, dh, qkv.shape[1], 1 if inverse else 0, _stream(qkv)), "vit_rope_qk")
    return qkv


def unfold_cast(x, P: int, S: int, N: int, out=None, out_dtype=torch.bfloat16):
    _chk(x, torch.float32, "unfold_cast x")
    h = _h(x)
    B, L = x.shape
    out = out if out is not None else torch.empty((B * N, P), dtype=out_dtype, device=x.device)
    check(h.lib.vit_unfold_cast(h.h, x.data_ptr(), out.data_ptr(), _DT[out.dtype], B, L, P, S, N, _stream(x)),
          "vit_unfold_cast")
    return out


def embed_finish(tokens, cls, pos=None, dropout: Dropout = NO_DROP):
    _chk(tokens, torch.float32, "embed_finish tokens")
    h = _h(tokens)
    B, T, D = tokens.shape
    p, seed, site = dropout
    check(h.lib.vit_embed_finish(h.h, tokens.data_ptr(), cls.data_ptr(), _ptr(pos), B, T, D, p, seed, site,
                                 _stream(tokens)), "vit_embed_finish")
    return tokens


def embed_finish_bwd(dtokens, dcls, dpos=None, dropout: Dropout = NO_DROP, dpatch=None, out_dtype=torch.bfloat16):
    _chk(dtokens, torch.float32, "embed_finish_bwd dtokens")
    h = _h(dtokens)
    B, T, D = dtokens.shape
    dpatch = dpatch if dpatch is not None else torch.empty((B * (T - 1), D), dtype=out_dtype, device=dtokens.device)
    p, seed, site = dropout
    check(h.lib.vit_embed_finish_bwd(h.h, dtokens.data_ptr(), dpatch.data_ptr(), _DT[dpatch.dtype], dcls.data_ptr(),
                                     _ptr(dpos), B, T, D, p, seed, site, 0, _stream(dtokens)), "vit_embed_finish_bwd")
    return dpatch


# ------------------------------------------------------------------------------------------------ elementwise
def dropout_bwd_cast(dx, dropout: Dropout = NO_DROP, out=None, out_dtype=torch.bfloat16):
    _chk(dx, torch.float32, "dropout_bwd_cast dx")
    h = _h(dx)
    cols = dx.shape[-1]
    rows = dx.numel() // cols
    out = out if out is not None else torch.empty(dx.shape, dtype=out_dtype, device=dx.device)
    p, seed, site = dropout
    check(h.lib.vit_dropout_bwd_cast(h.h, dx.data_ptr(), out.data_ptr(), _DT[out.dtype], rows, cols, p, seed, site,
                                     _stream(dx)), "vit_dropout_bwd_cast")
    return out


def colsum(a, out=None, accumulate: bool = False):
    h = _h(a)
    rows, cols = a.shape
    out = out if out is not None else torch.empty(cols, dtype=torch.float32, device=a.device)
    check(h.lib.vit_colsum(h.h, a.data_ptr(), _DT[a.dtype], a.stride(0), out.data_ptr(), rows, cols, int(accumulate),
                           _stream(a)), "vit_colsum")
    return out


def cast_f32_bf16(src, out=None):
    _chk(src, torch.float32, "cast_f32_bf16 src")
    h = _h(src)
    out = out if out is not None else torch.empty(src.shape, dtype=torch.bfloat16, device=src.device)
    check(h.lib.vit_cast_f32_bf16(h.h, src.data_ptr(), out.data_ptr(), src.numel(), _stream(src)), "vit_cast_f32_bf16")
    return out


def cast_bf16_f32(src, out, scale: float = 1.0):
    """out (f32) = scale * src (bf16)."""
    _chk(src, torch.bfloat16, "cast_bf16_f32 src")
    _chk(out, torch.float32, "cast_bf16_f32 out")
    h = _h(src)
    check(h.lib.vit_cast_bf16_f32(h.h, src.data_ptr(), out.data_ptr(), src.numel(), float(scale), _stream(src)), "vit_cast_bf16_f32")
    return out


# ------------------------------------------------------------------------------------------------ head + loss
def head_loss_fwd(last_hidden, W, b, labels, loss_kind: int):
    _chk(last_hidden, torch.float32, "head_loss_fwd last_hidden")
    h = _h(last_hidden)
    B, T, D = last_hidden.shape
    Cn = W.shape[0]
    logits = torch.empty((B, Cn), dtype=torch.float32, device=last_hidden.device)
    loss = torch.zeros((), dtype=torch.float32, device=last_hidden.device) if labels is not None else None
    check(h.lib.vit_head_loss_fwd(h.h, last_hidden.data_ptr(), W.data_ptr(), b.data_ptr(), _ptr(labels),
                                  logits.data_ptr(), _ptr(loss), B, T, D, Cn, loss_kind, _stream(last_hidden)),
          "vit_head_loss_fwd")
    return logits, loss


def head_loss_bwd(last_hidden, W, logits, labels, dloss, loss_kind: int, dlast=None, dW=None, db=None):
    h = _h(last_hidden)
    B, T, D = last_hidden.shape
    Cn = W.shape[0]
    dlast = dlast if dlast is not None else torch.empty_like(last_hidden)
    dW = dW if dW is not None else torch.empty_like(W)
    db = db if db is not None else torch.empty(Cn, dtype=torch.float32, device=W.device)
    check(h.lib.vit_head_loss_bwd(h.h, last_hidden.data_ptr(), W.data_ptr(), logits.data_ptr(), labels.data_ptr(),
                                  dloss.data_ptr(), dlast.data_ptr(), dW.data_ptr(), db.data_ptr(), B, T, D, Cn,
                                  loss_kind, 0, _stream(last_hidden)), "vit_head_loss_bwd")
    return dlast, dW, db


# ------------------------------------------------------------------------------------------------ optimizer
def grad_sqnorm(g, out=None, accumulate: bool = False):
    """out[0] = sum g^2 (accumulate: += ), deterministic two-stage reduction."""
    _chk(g, torch.float32, "grad_sqnorm g")
    h = _h(g)
    out = out if out is not None else torch.empty(1, dtype=torch.float32, device=g.device)
    fn = h.lib.vit_grad_sqnorm_acc if accumulate else h.lib.vit_grad_sqnorm
    check(fn(h.h, g.data_ptr(), g.numel(), out.data_ptr(), _stream(g)), "vit_grad_sqnorm")
    return out


def adamw_step(p, g, m, v, p_bf16, *, lr, beta1=0.9, beta2=0.999, eps=1e-8, weight_decay=0.0, step=1, sqnorm=None,
               max_norm=0.0, n: Optional[int] = None):
    h = _h(p)
    n = p.numel() if n is None else n
    check(h.lib.vit_adamw_step(h.h, p.data_ptr(), g.data_ptr(), m.data_ptr(), v.data_ptr(), _ptr(p_bf16), n, lr, beta1,
                               beta2, eps, weight_decay, step, _ptr(sqnorm), max_norm, _stream(p)), "vit_adamw_step")
