"""ORACLE (test infrastructure, NOT product code).

CPU restatement, in plain fp32 PyTorch ops, of the ViT training hot path of
ViskaWei/VIT.  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline
leg may import this file; the product path (vit_amd/) never does.

Parity status: PINNED against outputs of the reference's own modules run in the
build container -- oracle/make_golden.py composes the reference's
`get_vit_config` (src/models/builder.py:200-258), `SpectraEmbeddings`
(src/models/embedding.py:14-100) and `SlidingWindowTokenizer`
(src/models/tokenization.py:31-50) with the installed HuggingFace `ViTModel`
exactly as `MyViT` does (src/models/specvit.py:32-35, 68-94) and asserts this
restatement against it before writing tests/golden/*.npz.  The encoder
arithmetic itself lives in the third-party `transformers` package (pinned
4.56.0 in the reference's requirements.txt:57, 5.15.0 installed here, same
math); the reference has no tests of its own, so there are
no upstream golden vectors beyond what make_golden.py records.

Parameter names follow the reference's pin (transformers 4.56): see
`param_shapes()`.
"""
from __future__ import annotations

import math
from dataclasses import dataclass, field
from typing import Dict, List, Optional

import numpy as np
import torch
import torch.nn.functional as F


# --------------------------------------------------------------------------- config
@dataclass
class RefConfig:
    """The subset of ViTConfig the path uses; defaults are the values
    get_vit_config hard-codes (src/models/builder.py:234-258)."""

    image_size: int
    patch_size: int
    hidden_size: int
    num_hidden_layers: int
    num_attention_heads: int
    stride_size: Optional[int] = None
    stride_ratio: float = 1
    proj_fn: str = "SW"
    task_type: str = "reg"
    num_labels: int = 1
    pos_encoding_type: Optional[str] = None
    max_position_embeddings: int = 512  # builder.py:231, 256
    rope_base: float = 10000.0          # builder.py:232, 257
    hidden_dropout_prob: float = 0.1
    attention_probs_dropout_prob: float = 0.1
    layer_norm_eps: float = 1e-12
    loss_name: str = ""

    @property
    def intermediate_size(self) -> int:  # builder.py:243
        return 4 * self.hidden_size

    @property
    def stride(self) -> int:  # embedding.py:25-26, tokenization.py:36
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

    @property
    def head_name(self) -> str:  # specvit.py:46-50
        return "classifier" if self.task_type == "cls" else "regressor"


def config_from_dict(config: dict) -> RefConfig:
    """Restates get_vit_config (src/models/builder.py:200-258) + the loss-name plumbing of
    get_model (builder.py:142)."""
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
    else:
        num_labels = int(m.get("num_labels", 1) or 1)
    return RefConfig(
        image_size=m["image_size"],
        patch_size=m["patch_size"],
        hidden_size=m["hidden_size"],
        num_hidden_layers=m["num_hidden_layers"],
        num_attention_heads=m["num_attention_heads"],
        stride_size=m.get("stride_size", None),
        stride_ratio=m.get("stride_ratio", 1),
        proj_fn=m["proj_fn"],
        task_type=m["task_type"],
        num_labels=num_labels,
        pos_encoding_type=m.get("pos_encoding_type", None),
        max_position_embeddings=m.get("max_position_embeddings", 512),
        rope_base=m.get("rope_base", 10000.0),
        loss_name=(config.get("loss", {}) or {}).get("name", None) or "",
    )


# named configurations of SURVEY.md section 8 / BASELINE.md section 2
def named_config(name: str) -> RefConfig:
    table = {
        # configs/exp/att_clp/baseline.yaml:9-27
        "C1": dict(image_size=4096, patch_size=32, hidden_size=32, num_hidden_layers=3,
                   num_attention_heads=2, stride_size=32, loss_name="mae"),
        "C2": dict(image_size=1024, patch_size=256, hidden_size=192, num_hidden_layers=12,
                   num_attention_heads=3, stride_size=256, loss_name="mae"),
        "C3": dict(image_size=50176, patch_size=256, hidden_size=768, num_hidden_layers=12,
                   num_attention_heads=12, stride_size=256, loss_name="mae"),
        "C5": dict(image_size=147456, patch_size=256, hidden_size=1024, num_hidden_layers=24,
                   num_attention_heads=16, stride_size=256, loss_name="mae"),
        # four corners of the reference's own sweep space (configs/sweep.yaml:10-21: patch 8..256 x stride 1..32 x hidden
        # {32, 128} x heads {2, 4, 8} x layers {3, 4, 6} x {SW, CNN} at image_size 4096)
        "S1": dict(image_size=4096, patch_size=8, hidden_size=32, num_hidden_layers=3,      # T 4090, head_dim 4
                   num_attention_heads=8, stride_size=1, loss_name="mae"),
        "S2": dict(image_size=4096, patch_size=64, hidden_size=128, num_hidden_layers=3,    # T 4034, head_dim 64
                   num_attention_heads=2, stride_size=1, loss_name="mae"),
        "S3": dict(image_size=4096, patch_size=16, hidden_size=128, num_hidden_layers=4,    # T 2042, head_dim 16, Conv1D
                   num_attention_heads=8, stride_size=2, proj_fn="CNN", loss_name="mae"),
        "S4": dict(image_size=4096, patch_size=256, hidden_size=32, num_hidden_layers=6,    # T 122, head_dim 8
                   num_attention_heads=4, stride_size=32, loss_name="mae"),
    }
    return RefConfig(**table[name])


# --------------------------------------------------------------------------- parameters
def param_shapes(cfg: RefConfig) -> Dict[str, tuple]:
    """state_dict layout of MyViT under transformers 4.56 names (specvit.py:32-50; names corroborated by
    specvit.py:64-66, vit_with_rope.py:54-56, cka_callback.py:99, viz_callback.py:233)."""
    D, P, Fd = cfg.hidden_size, cfg.patch_size, cfg.intermediate_size
    s: Dict[str, tuple] = {}
    s["vit.embeddings.cls_token"] = (1, 1, D)
    if cfg.pos_encoding_type == "learned":
        s["vit.embeddings.position_embeddings"] = (1, cfg.num_patches + 1, D)
    if cfg.proj_fn == "SW":
        s["vit.embeddings.patch_embeddings.projection.weight"] = (D, P)
    else:
        s["vit.embeddings.patch_embeddings.projection.weight"] = (D, 1, P)
    s["vit.embeddings.patch_embeddings.projection.bias"] = (D,)
    for i in range(cfg.num_hidden_layers):
        pre = f"vit.encoder.layer.{i}."
        for n in ("query", "key", "value"):
            s[pre + f"attention.attention.{n}.weight"] = (D, D)
            s[pre + f"attention.attention.{n}.bias"] = (D,)
        s[pre + "attention.output.dense.weight"] = (D, D)
        s[pre + "attention.output.dense.bias"] = (D,)
        s[pre + "intermediate.dense.weight"] = (Fd, D)
        s[pre + "intermediate.dense.bias"] = (Fd,)
        s[pre + "output.dense.weight"] = (D, Fd)
        s[pre + "output.dense.bias"] = (D,)
        s[pre + "layernorm_before.weight"] = (D,)
        s[pre + "layernorm_before.bias"] = (D,)
        s[pre + "layernorm_after.weight"] = (D,)
        s[pre + "layernorm_after.bias"] = (D,)
    s["vit.layernorm.weight"] = (D,)
    s["vit.layernorm.bias"] = (D,)
    s["vit.pooler.dense.weight"] = (D, D)  # add_pooling_layer default True (specvit.py:32); output never used (:78)
    s["vit.pooler.dense.bias"] = (D,)
    s[cfg.head_name + ".weight"] = (cfg.num_labels, D)
    s[cfg.head_name + ".bias"] = (cfg.num_labels,)
    return s


def make_state_dict(cfg: RefConfig, seed: int) -> Dict[str, torch.Tensor]:
    """Deterministic synthetic parameters (numpy PCG64, independent of torch's generator) so that fixtures need not
    store multi-MB weights.  Distribution mimics the reference's init (HF _init_weights: Linear ~ N(0, 0.02), bias 0,
    LayerNorm 1/0; cls_token stays torch.randn, embedding.py:47) but with non-trivial biases / LN affine so that every
    term of the arithmetic is exercised."""
    rng = np.random.Generator(np.random.PCG64(seed))
    sd: Dict[str, torch.Tensor] = {}
    for name, shape in param_shapes(cfg).items():
        if name.endswith("cls_token") or name.endswith("position_embeddings"):
            a = rng.standard_normal(shape)
        elif "layernorm" in name and name.endswith("weight"):
            a = 1.0 + 0.1 * rng.standard_normal(shape)
        elif "layernorm" in name and name.endswith("bias"):
            a = 0.05 * rng.standard_normal(shape)
        elif name.endswith("bias"):
            a = 0.02 * rng.standard_normal(shape)
        elif "patch_embeddings" in name:
            a = rng.standard_normal(shape) / math.sqrt(cfg.patch_size)
        elif name.startswith(cfg.head_name):
            a = rng.standard_normal(shape) / math.sqrt(cfg.hidden_size)
        else:  # encoder Linear weights: fan-in scaling keeps activations O(1) through all layers
            a = rng.standard_normal(shape) / math.sqrt(shape[-1])
        sd[name] = torch.from_numpy(a.astype(np.float32))
    return sd


def make_inputs(cfg: RefConfig, batch: int, seed: int):
    """Synthetic batch with the reference's contract (src/dataloader/spec_datasets.py:28-34): flux, error f32 [B,L];
    labels f32 [B] in [0,1) (reg, min-max normalised) or int64 [B] (cls)."""
    rng = np.random.Generator(np.random.PCG64(seed))
    flux = torch.from_numpy(rng.standard_normal((batch, cfg.image_size)).astype(np.float32))
    error = torch.from_numpy((0.1 * np.abs(rng.standard_normal((batch, cfg.image_size)))).astype(np.float32))
    if cfg.task_type == "cls":
        labels = torch.from_numpy(rng.integers(0, cfg.num_labels, size=(batch,)).astype(np.int64))
    elif cfg.num_labels == 1:
        labels = torch.from_numpy(rng.random((batch,)).astype(np.float32))
    else:
        labels = torch.from_numpy(rng.random((batch, cfg.num_labels)).astype(np.float32))
    return flux, error, labels


# --------------------------------------------------------------------------- rotary position embedding
def rope_tables(dim: int, seq_len: int, base: float = 10000.0):
    """RotaryPositionEmbedding.__init__/_precompute_freqs (src/models/rope.py:36-56): cos/sin of
    outer(arange(seq_len), 1 / base**(arange(0, dim, 2)/dim)), the half-width table repeated twice along the last dim."""
    inv_freq = 1.0 / (base ** (torch.arange(0, dim, 2).float() / dim))
    t = torch.arange(seq_len).type_as(inv_freq)
    freqs = torch.outer(t, inv_freq)
    emb = torch.cat([freqs, freqs], dim=-1)
    return emb.cos(), emb.sin()


def rotate_half(x: torch.Tensor) -> torch.Tensor:
    """rope.py:58-64: [x1, x2] -> [-x2, x1] over the two halves of the last dim."""
    x1, x2 = x.chunk(2, dim=-1)
    return torch.cat([-x2, x1], dim=-1)


def apply_rope(x: torch.Tensor, cos: torch.Tensor, sin: torch.Tensor) -> torch.Tensor:
    """rope.py:66-98 for the 4-D (batch, heads, seq, head_dim) case used by the attention (vit_with_rope.py:58-60)."""
    T = x.shape[2]
    c, s_ = cos[:T, :].unsqueeze(0).unsqueeze(0), sin[:T, :].unsqueeze(0).unsqueeze(0)
    return (x * c) + (rotate_half(x) * s_)


# --------------------------------------------------------------------------- forward
@dataclass
class RefOutput:
    loss: Optional[torch.Tensor]
    logits: torch.Tensor
    hidden_states: Optional[List[torch.Tensor]] = None
    attentions: Optional[List[torch.Tensor]] = None
    last_hidden_state: Optional[torch.Tensor] = None
    tokens: Optional[torch.Tensor] = None


def tokenize(cfg: RefConfig, sd: Dict[str, torch.Tensor], x: torch.Tensor) -> torch.Tensor:
    """SlidingWindowTokenizer.forward (tokenization.py:43-50) / Conv1DPatchTokenizer.forward (:66-69)."""
    w = sd["vit.embeddings.patch_embeddings.projection.weight"]
    b = sd["vit.embeddings.patch_embeddings.projection.bias"]
    P, S, N = cfg.patch_size, cfg.stride, cfg.num_patches
    if cfg.proj_fn == "SW":
        patches = x.unfold(1, P, S)
        if patches.size(1) < N:  # zero-pad the ragged tail patch (tokenization.py:46-48)
            pad = torch.zeros(x.size(0), N - patches.size(1), P, dtype=x.dtype)
            patches = torch.cat([patches, pad], dim=1)
        patches = patches.contiguous().reshape(x.size(0), N, P)
        return F.linear(patches, w, b)
    y = F.conv1d(x.reshape(-1, 1, cfg.image_size), w, b, stride=S)
    return y.transpose(1, 2)


def forward(
    cfg: RefConfig,
    sd: Dict[str, torch.Tensor],
    x: torch.Tensor,
    labels: Optional[torch.Tensor] = None,
    *,
    training: bool = False,
    p_hidden: Optional[float] = None,
    p_attn: Optional[float] = None,
    output_hidden_states: bool = False,
    output_attentions: bool = False,
) -> RefOutput:
    """MyViT.forward (specvit.py:68-94) over HF ViTModel.forward semantics.

    Dropout uses torch's CPU generator when `training`; parity tests run it with p=0 or training=False because dropout
    masks are implementation-defined (hard parts)."""
    ph = cfg.hidden_dropout_prob if p_hidden is None else p_hidden
    pa = cfg.attention_probs_dropout_prob if p_attn is None else p_attn
    B = x.size(0)
    D, H, dh = cfg.hidden_size, cfg.num_attention_heads, cfg.head_dim

    def drop(t, p):
        return F.dropout(t, p=p, training=training) if (training and p > 0) else t

    # --- embeddings (embedding.py:79-100)
    tok = tokenize(cfg, sd, x)
    h = torch.cat((sd["vit.embeddings.cls_token"].expand(B, -1, -1), tok), dim=1)
    if cfg.pos_encoding_type == "learned":
        h = h + sd["vit.embeddings.position_embeddings"]
    elif cfg.pos_encoding_type not in (None, "none", "rope"):
        raise ValueError(f"Unsupported pos_encoding_type '{cfg.pos_encoding_type}'")
    h = drop(h, ph)

    hs = [h] if output_hidden_states else None
    atts = [] if output_attentions else None
    T = h.size(1)
    rope = None
    if cfg.pos_encoding_type == "rope":
        # ViTSelfAttentionWithRoPE.__init__ (vit_with_rope.py:26-39): per-head tables of max_position_embeddings rows,
        # re-computed by the same formula when the sequence is longer (rope.py:110-112)
        rope = rope_tables(dh, max(T, cfg.max_position_embeddings), cfg.rope_base)
    for i in range(cfg.num_hidden_layers):
        pre = f"vit.encoder.layer.{i}."
        res = h
        y = F.layer_norm(h, (D,), sd[pre + "layernorm_before.weight"], sd[pre + "layernorm_before.bias"], cfg.layer_norm_eps)
        q = F.linear(y, sd[pre + "attention.attention.query.weight"], sd[pre + "attention.attention.query.bias"])
        k = F.linear(y, sd[pre + "attention.attention.key.weight"], sd[pre + "attention.attention.key.bias"])
        v = F.linear(y, sd[pre + "attention.attention.value.weight"], sd[pre + "attention.attention.value.bias"])
        q = q.view(B, T, H, dh).transpose(1, 2)
        k = k.view(B, T, H, dh).transpose(1, 2)
        v = v.view(B, T, H, dh).transpose(1, 2)
        if rope is not None:  # vit_with_rope.py:58-60 -> rope.py:116-131
            q, k = apply_rope(q, *rope), apply_rope(k, *rope)
        # vit_with_rope.py:63-71 (the in-repo statement of the eager arithmetic); there is no logit clamp
        scores = torch.matmul(q, k.transpose(-1, -2)) / math.sqrt(dh)
        probs = F.softmax(scores, dim=-1)
        if output_attentions:
            atts.append(probs)
        ctx = torch.matmul(drop(probs, pa), v)
        ctx = ctx.transpose(1, 2).contiguous().view(B, T, D)
        a = F.linear(ctx, sd[pre + "attention.output.dense.weight"], sd[pre + "attention.output.dense.bias"])
        h = drop(a, ph) + res
        res = h
        y = F.layer_norm(h, (D,), sd[pre + "layernorm_after.weight"], sd[pre + "layernorm_after.bias"], cfg.layer_norm_eps)
        y = F.gelu(F.linear(y, sd[pre + "intermediate.dense.weight"], sd[pre + "intermediate.dense.bias"]))  # erf GELU
        y = F.linear(y, sd[pre + "output.dense.weight"], sd[pre + "output.dense.bias"])
        h = drop(y, ph) + res
        if output_hidden_states:
            hs.append(h)
    last = F.layer_norm(h, (D,), sd["vit.layernorm.weight"], sd["vit.layernorm.bias"], cfg.layer_norm_eps)
    cls = last[:, 0, :]  # specvit.py:78 (pooler output is computed by HF and never used)
    logits = F.linear(cls, sd[cfg.head_name + ".weight"], sd[cfg.head_name + ".bias"])
    loss = None
    if labels is not None:
        loss = loss_fn(cfg, logits, labels)
    return RefOutput(loss=loss, logits=logits, hidden_states=hs, attentions=atts, last_hidden_state=last, tokens=tok)


def resolved_loss(cfg: RefConfig) -> str:
    """specvit.py:45-53: 'ce' for cls; for reg L1 iff 'l1' in loss_name.lower() else MSE (so baseline.yaml's
    loss.name 'mae' resolves to MSELoss)."""
    if cfg.task_type == "cls":
        return "ce"
    if cfg.task_type == "reg":
        return "l1" if "l1" in (cfg.loss_name or "l2").lower() else "mse"
    raise ValueError(f"Unsupported task_type '{cfg.task_type}'")


def loss_fn(cfg: RefConfig, logits: torch.Tensor, labels: torch.Tensor) -> torch.Tensor:
    kind = resolved_loss(cfg)
    if kind == "ce":
        return F.cross_entropy(logits.view(-1, cfg.num_labels), labels.view(-1))
    if kind == "l1":
        return F.l1_loss(logits.view(-1), labels.view(-1).float())
    return F.mse_loss(logits.view(-1), labels.view(-1).float())


# --------------------------------------------------------------------------- training step
class RefTrainer:
    """One optimisation step as the reference's Trainer performs it: zero_grad -> fwd -> bwd ->
    clip global grad-norm (basemodule.py:244, default 0.5) -> AdamW(lr, weight_decay=0) (opt/optimizer.py:49-51,108)."""

    def __init__(self, cfg: RefConfig, sd: Dict[str, torch.Tensor], lr: float = 1e-3, weight_decay: float = 0.0,
                 grad_clip: float = 0.5, training: bool = True, p_hidden: Optional[float] = None,
                 p_attn: Optional[float] = None):
        self.cfg = cfg
        self.params = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
        self.opt = torch.optim.AdamW(list(self.params.values()), lr=lr, weight_decay=weight_decay)
        self.grad_clip = grad_clip
        self.training = training
        self.p_hidden, self.p_attn = p_hidden, p_attn
        self.last_grad_norm = None

    def step(self, flux: torch.Tensor, labels: torch.Tensor) -> float:
        self.opt.zero_grad(set_to_none=True)
        out = forward(self.cfg, self.params, flux, labels, training=self.training, p_hidden=self.p_hidden,
                      p_attn=self.p_attn)
        out.loss.backward()
        with_grad = [p for p in self.params.values() if p.grad is not None]  # the pooler never gets one
        self.last_grad_norm = float(torch.nn.utils.clip_grad_norm_(with_grad, self.grad_clip))
        self.opt.step()
        return float(out.loss.detach())

    def grads(self) -> Dict[str, Optional[torch.Tensor]]:
        return {k: (None if p.grad is None else p.grad.detach().clone()) for k, p in self.params.items()}

    def state_dict(self) -> Dict[str, torch.Tensor]:
        return {k: p.detach().clone() for k, p in self.params.items()}
