"""Host-side engine of the ViT step: owns the flat parameter / gradient buffers and the activation arena in HBM and
sequences the HIP kernels of libvit_amd.so (through vit_amd.functional -> ctypes -> C ABI) for forward and backward.

Data layout in HBM (one MI355X, 288 GB: nothing is recomputed or re-materialised to save memory)
  params   f32 [n_total]   flat; every tensor of the reference's state_dict (transformers-4.56 names) is a view.
                           query/key/value weights (and biases) of a layer are adjacent, so the fused QKV projection
                           reads them as one [3D, D] matrix.  Pooler last (never receives a gradient: specvit.py:78).
  shadow   bf16 [n_total]  same offsets; what the MFMA GEMMs read.  Refreshed by the fused AdamW kernel, or by one cast
                           pass whenever the f32 buffer was modified behind our back (load_state_dict, a torch optimizer).
  grads    f32 [n_total]   same offsets; every kernel that produces a parameter gradient writes its slice exactly once.
  arena    per layer: x_in f32 [M,D] (residual stream), h1/h2 bf16 [M,D] (LN outputs), qkv bf16 [M,3D], ctx bf16 [M,D],
           lse f32 [B*H,T], x1 f32 [M,D], u/g bf16 [M,4D] (pre/post GELU), LN statistics.  M = B*T token rows.
Dropout masks are regenerated from (seed, site) in backward; no mask is stored.
"""
from __future__ import annotations

import math
from typing import Callable, Dict, List, Optional, Tuple

import torch

from . import functional as vf
from ._cabi import ACT_GELU, LOSS_CE, LOSS_L1, LOSS_MSE, VitError
from .config import ViTConfig

ALIGN = 8  # elements; keeps every view 32-byte (f32) / 16-byte (bf16) aligned


def resolved_loss(task_type: str, loss_name: str) -> Tuple[str, int]:
    """specvit.py:45-55: cls -> CrossEntropy ('ce'); reg -> L1 iff 'l1' in loss_name.lower() else MSE."""
    if task_type == "cls":
        return "ce", LOSS_CE
    if task_type == "reg":
        ln = loss_name or "l2"
        return ln, (LOSS_L1 if "l1" in ln.lower() else LOSS_MSE)
    raise ValueError(f"Unsupported task_type '{task_type}'")


class ParamLayout:
    """name -> (offset, shape) inside the flat buffers; `state_order` is the reference's state_dict order."""

    def __init__(self, cfg: ViTConfig):
        D, P, Fd, T = cfg.hidden_size, cfg.patch_size, cfg.intermediate_size, cfg.seq_len
        self.entries: Dict[str, Tuple[int, Tuple[int, ...]]] = {}
        self.layer_ranges: List[Tuple[int, int]] = []
        off = 0

        def add(name, shape):
            nonlocal off
            n = 1
            for s in shape:
                n *= s
            self.entries[name] = (off, tuple(shape))
            off += (n + ALIGN - 1) // ALIGN * ALIGN

        e = "vit.embeddings."
        self.embed_start = off
        add(e + "cls_token", (1, 1, D))
        if cfg.pos_encoding_type == "learned":
            add(e + "position_embeddings", (1, T, D))
        add(e + "patch_embeddings.projection.weight", (D, P) if cfg.proj_fn == "SW" else (D, 1, P))
        add(e + "patch_embeddings.projection.bias", (D,))
        self.embed_end = off
        for i in range(cfg.num_hidden_layers):
            p = f"vit.encoder.layer.{i}."
            start = off
            for n in ("query", "key", "value"):
                add(p + f"attention.attention.{n}.weight", (D, D))
            for n in ("query", "key", "value"):
                add(p + f"attention.attention.{n}.bias", (D,))
            add(p + "attention.output.dense.weight", (D, D))
            add(p + "attention.output.dense.bias", (D,))
            add(p + "intermediate.dense.weight", (Fd, D))
            add(p + "intermediate.dense.bias", (Fd,))
            add(p + "output.dense.weight", (D, Fd))
            add(p + "output.dense.bias", (D,))
            add(p + "layernorm_before.weight", (D,))
            add(p + "layernorm_before.bias", (D,))
            add(p + "layernorm_after.weight", (D,))
            add(p + "layernorm_after.bias", (D,))
            self.layer_ranges.append((start, off))
        self.tail_start = off
        add("vit.layernorm.weight", (D,))
        add("vit.layernorm.bias", (D,))
        head = "classifier" if cfg.task_type == "cls" else "regressor"
        self.head = head
        add(head + ".weight", (cfg.num_labels, D))
        add(head + ".bias", (cfg.num_labels,))
        self.n_trainable = off
        add("vit.pooler.dense.weight", (D, D))
        add("vit.pooler.dense.bias", (D,))
        self.n_total = off
        # the order MyViT's state_dict has in the reference (HF module registration order)
        order = [e + "cls_token"]
        if cfg.pos_encoding_type == "learned":
            order.append(e + "position_embeddings")
        order += [e + "patch_embeddings.projection.weight", e + "patch_embeddings.projection.bias"]
        for i in range(cfg.num_hidden_layers):
            p = f"vit.encoder.layer.{i}."
            for n in ("query", "key", "value"):
                order += [p + f"attention.attention.{n}.weight", p + f"attention.attention.{n}.bias"]
            for n in ("attention.output.dense", "intermediate.dense", "output.dense", "layernorm_before", "layernorm_after"):
                order += [p + n + ".weight", p + n + ".bias"]
        order += ["vit.layernorm.weight", "vit.layernorm.bias", "vit.pooler.dense.weight", "vit.pooler.dense.bias",
                  head + ".weight", head + ".bias"]
        assert sorted(order) == sorted(self.entries)
        self.state_order = order

    def view(self, flat: torch.Tensor, name: str) -> torch.Tensor:
        off, shape = self.entries[name]
        n = 1
        for s in shape:
            n *= s
        return flat[off:off + n].view(shape)

    def numel(self, name: str) -> int:
        n = 1
        for s in self.entries[name][1]:
            n *= s
        return n

    def buckets(self) -> List[Tuple[int, int]]:
        """Gradient buckets in the order backward completes them: tail (final LN + head), layers L-1..0, embeddings."""
        return [(self.tail_start, self.n_trainable)] + list(reversed(self.layer_ranges)) + \
               [(self.embed_start, self.embed_end)]


class ViTEngine:
    def __init__(self, cfg: ViTConfig, loss_name: str = ""):
        if cfg.pos_encoding_type == "rope" and (cfg.head_dim % 8) != 0:
            raise ValueError(f"pos_encoding_type='rope' needs head_dim % 8 == 0 here (got {cfg.head_dim}); "
                             f"the reference needs it even (rope.py:28-29)")
        if cfg.pos_encoding_type not in (None, "none", "learned", "rope"):
            raise ValueError(f"Unsupported pos_encoding_type '{cfg.pos_encoding_type}'. "
                             f"Choose from: 'rope', 'learned', 'none', or None")  # embedding.py:74
        if cfg.proj_fn not in ("SW", "C1D", "CNN"):
            raise ValueError(f"Unsupported proj_fn '{cfg.proj_fn}'")  # embedding.py:44
        # what the kernels take, said at construction instead of as a VIT_ERR_UNSUPPORTED in the middle of a step.  The
        # reference's own sweep (configs/sweep.yaml:10-21) reaches head_dim 4 (hidden 32 / 8 heads) and T = 4090 (patch 8,
        # stride 1): both run in both precisions.
        if cfg.hidden_size % cfg.num_attention_heads:
            raise ValueError(f"The hidden size {cfg.hidden_size} is not a multiple of the number of attention heads "
                             f"{cfg.num_attention_heads}.")  # HF ViTSelfAttention.__init__
        if cfg.head_dim % 4 or cfg.head_dim > 128 or cfg.hidden_size % 8 or cfg.patch_size % 8:
            raise ValueError(f"vit_amd kernels need head_dim % 4 == 0 and <= 128, hidden_size % 8 == 0, patch_size % 8 == 0 "
                             f"(got head_dim {cfg.head_dim}, hidden_size {cfg.hidden_size}, patch_size {cfg.patch_size})")
        self.cfg = cfg
        self.loss_name, self.loss_kind = resolved_loss(cfg.task_type, loss_name)
        self.layout = ParamLayout(cfg)
        self.flat = torch.zeros(self.layout.n_total, dtype=torch.float32)
        self.shadow: Optional[torch.Tensor] = None
        self.grads: Optional[torch.Tensor] = None
        self._shadow_version = -1
        self._arena_key = None
        self._arenas: Dict[bool, tuple] = {}  # train? -> (key, act, tmp, Mp): one arena for training, one for evaluation
        self.act: Dict[str, object] = {}
        self.tmp: Dict[str, torch.Tensor] = {}
        self.precision = "f32"       # 'f32' (reference default precision='32') | 'bf16' (precision='bf16-mixed')
        self.base_seed = int(torch.initial_seed()) & 0x7FFFFFFFFFFFFFFF
        self.step_counter = 0
        self._last = None
        # weight-gradient GEMMs on a second HIP stream (see _dw): None = off
        self.side_stream: Optional[torch.cuda.Stream] = None
        self._side_handle = None
        self._main_handle = None  # this engine's own vit_handle (workspace + launch geometry): see handle()
        self.reserve_cus = -1     # -1 = the process-wide vit_set_option value
        self._side_reads: Dict[str, object] = {}
        # True / False / "auto": the second stream pays when the main stream's GEMMs leave a large part of the chip idle for
        # whole kernels (see _want_side)
        self.overlap_dw = "auto"
        self._gen = 0  # bumped by every forward: the activation arena holds ONE forward, backward checks it is still that one
        self.grad_ready_cb: Optional[Callable[[int, int], None]] = None
        # Parameters are views of `flat` with their OWN version counters (nn.Parameter / .data re-pointing do not share
        # the base's), so "were the f32 weights modified since the bf16 shadow was made?" is answered by a signature
        # over the parameters' versions, supplied by the owning module.
        self.version_fn: Callable[[], int] = lambda: self.flat._version

    # ------------------------------------------------------------------ precision
    def set_precision(self, precision) -> str:
        """Lightning-style precision strings (basemodule.py:233: `precision=str(train.precision or '32')`).
        '32' -> fp32-class arithmetic: f32 activations, split-bf16 ("x3") MFMA GEMMs, fp32 attention / LayerNorm.
        'bf16-mixed' -> bf16 MFMA operands with fp32 accumulation, residual stream and statistics (the fast path)."""
        ps = str(precision).lower()
        if ps in ("32", "32-true", "fp32", "f32", "float32"):
            mode = "f32"
        elif ps in ("bf16-mixed", "bf16", "bf16-true", "bfloat16"):
            mode = "bf16"
        elif ps in ("16-mixed", "16", "16-true", "fp16"):
            raise ValueError(f"precision '{precision}': fp16 arithmetic is not implemented on this path (it would need loss "
                             f"scaling); use '32' or 'bf16-mixed'")
        else:
            raise ValueError(f"Unsupported precision '{precision}'")
        if mode == "f32" and self.cfg.seq_len > 4096:
            raise ValueError(f"precision '{precision}': the fp32 attention kernels keep a score row in the LDS and take at most "
                             f"4096 tokens (this model has {self.cfg.seq_len}); use 'bf16-mixed'")
        if mode != self.precision:
            self.precision = mode
            self._drop_arenas()
        return mode

    @property
    def adt(self):
        return torch.bfloat16 if self.precision == "bf16" else torch.float32

    # ------------------------------------------------------------------ parameters
    @property
    def device(self):
        return self.flat.device

    def rebind(self, new_flat: torch.Tensor):
        if new_flat.dtype != torch.float32:
            raise VitError("vit_amd keeps master parameters in fp32 (precision '32' / 'bf16-mixed' semantics); "
                           "casting the module to another dtype is not supported")
        self.flat = new_flat.contiguous()
        self.shadow = None
        self.grads = None
        self._shadow_version = -1
        self._drop_arenas()

    def p(self, name: str) -> torch.Tensor:
        return self.layout.view(self.flat, name)

    def _ensure_device_state(self):
        if not self.flat.is_cuda:
            raise VitError("vit_amd runs on an MI355X only: move the model to the GPU first (model.to('cuda')); "
                           "there is no CPU fallback path")
        if self.grads is None:
            self.grads = torch.zeros(self.layout.n_total, dtype=torch.float32, device=self.flat.device)
        if self.precision != "bf16":
            return  # the x3 GEMMs read the f32 master weights directly
        if self.shadow is None:
            self.shadow = torch.empty(self.layout.n_total, dtype=torch.bfloat16, device=self.flat.device)
            self._shadow_version = -1
        ver = self.version_fn()
        if self._shadow_version != ver:
            vf.cast_f32_bf16(self.flat, self.shadow)
            self._shadow_version = ver

    def handle(self):
        """This engine's own vit_handle.  Launch geometry (`reserve_cus`) and the split-K / reduction workspace belong to the
        handle, so two engines of one process -- a training and an evaluation model, a sweep's trials -- do not see each
        other's settings (include/vit_amd.h: vit_handle_set_option)."""
        from . import _cabi

        dev = self.flat.device
        if self._main_handle is None or self._main_handle.device_index != (dev.index if dev.index is not None else torch.cuda.current_device()):
            self._main_handle = _cabi.Handle(dev.index if dev.index is not None else torch.cuda.current_device())
            self._main_handle.set_option("reserve_cus", self.reserve_cus)
        return self._main_handle

    def set_reserve_cus(self, n: int):
        """CUs the one-workgroup-per-CU kernels of THIS engine leave to a collective that overlaps them (data-parallel runs);
        -1 = follow the process-wide default."""
        self.reserve_cus = int(n)
        for h in (self._main_handle, self._side_handle):
            if h is not None:
                h.set_option("reserve_cus", self.reserve_cus)

    def mark_shadow_fresh(self):
        self._shadow_version = self.version_fn()

    def w16(self, name: str) -> torch.Tensor:
        """GEMM weight operand: bf16 shadow view, or the f32 master view in f32 mode."""
        return self.layout.view(self.shadow if self.precision == "bf16" else self.flat, name)

    def g(self, name: str) -> torch.Tensor:
        return self.layout.view(self.grads, name)

    def _qkv16(self, i: int):
        off, _ = self.layout.entries[f"vit.encoder.layer.{i}.attention.attention.query.weight"]
        D = self.cfg.hidden_size
        src = self.shadow if self.precision == "bf16" else self.flat
        return src[off:off + 3 * D * D].view(3 * D, D)

    def _qkv_bias(self, i: int, buf: torch.Tensor):
        off, _ = self.layout.entries[f"vit.encoder.layer.{i}.attention.attention.query.bias"]
        D = self.cfg.hidden_size
        return buf[off:off + 3 * D]

    def _qkv_wgrad(self, i: int):
        off, _ = self.layout.entries[f"vit.encoder.layer.{i}.attention.attention.query.weight"]
        D = self.cfg.hidden_size
        return self.grads[off:off + 3 * D * D].view(3 * D, D)

    # ------------------------------------------------------------------ arena
    def _ensure_arena(self, B: int, train: bool):
        """Select (allocating on first use) the activation arena for this batch size.  Training and evaluation forwards keep
        SEPARATE arenas (one slot each): a validation pass between two optimisation steps must not release the training
        arena -- a captured hipGraph (vit_amd/graph.py) replays raw pointers into it, and re-allocating 17 GB per epoch would
        be waste in any case.  A slot is replaced when its batch size or the precision changes."""
        key = (B, train, self.precision)
        if self._arena_key == key:
            return
        slot = self._arenas.get(train)
        if slot is not None and slot[0] == key:
            self._arena_key, self.act, self.tmp, self._Mp = key, slot[1], slot[2], slot[3]
            return
        self.act, self.tmp = {}, {}
        self._arenas.pop(train, None)  # release the slot's old tensors before allocating their replacement
        c = self.cfg
        dev = self.flat.device
        T, D, Fd, H, L, N, P = c.seq_len, c.hidden_size, c.intermediate_size, c.num_attention_heads, \
            c.num_hidden_layers, c.num_patches, c.patch_size
        M = B * T
        # GEMM row count: token rows padded to the 256-row tiles of the ping-pong GEMM core when the widths allow it.
        # Every buffer a GEMM reads or writes is allocated ZEROED with Mp rows and used through its first M rows by
        # everything else (LayerNorm, attention, reducers write real rows only), so pad rows hold zeros or finite
        # bias-only values in the forward tensors and exact zeros in every gradient tensor: dW = dY^T X over Mp rows
        # equals the sum over the M real rows, and bias-gradient column sums likewise.
        Mp = -(-M // 256) * 256 if (D % 256 == 0 and Fd % 256 == 0) else M
        self._Mp = Mp
        nl = L if train else 1
        f32, b16 = torch.float32, self.adt  # "b16" = the activation/operand dtype of the current precision mode

        def E(shape, dt):
            return torch.empty(shape, dtype=dt, device=dev)

        def G(cols, dt):  # a GEMM operand / result with M real rows (+ zeroed pad rows behind the view)
            return torch.zeros((Mp, cols), dtype=dt, device=dev)[:M]

        self.act = dict(
            patches=E((B * N, P), b16),
            x=[E((B, T, D), f32) for _ in range(L + 1)],
            h1=[G(D, b16) for _ in range(nl)], qkv=[G(3 * D, b16) for _ in range(nl)],
            ctx=[G(D, b16) for _ in range(nl)], lse=[E((B * H, T), f32) for _ in range(nl)],
            # bf16 training: the rounding residual of ctx, so the backward's delta sees the context to ~16 bits (vit_amd.h)
            ctx_lo=[E((M, D), b16) if (train and self.precision == "bf16") else None for _ in range(nl)],
            x1=[E((M, D), f32) for _ in range(nl)], h2=[G(D, b16) for _ in range(nl)],
            u=[G(Fd, b16) for _ in range(nl)], g=[G(Fd, b16) for _ in range(nl)],
            mean1=[E((M,), f32) for _ in range(nl)], rstd1=[E((M,), f32) for _ in range(nl)],
            mean2=[E((M,), f32) for _ in range(nl)], rstd2=[E((M,), f32) for _ in range(nl)],
            last=E((B, T, D), f32), meanF=E((M,), f32), rstdF=E((M,), f32),
            y=G(D, b16),  # dropout(Linear(.)) of the attention-output / FC2 projection, until the next LayerNorm adds it
        )
        self.tmp = {}
        if train:
            self.tmp = dict(
                dxa=E((M, D), f32), dxb=E((M, D), f32), dy=G(D, b16), dy2=G(D, b16), dU=G(Fd, b16), dh=G(D, b16),
                dqkv=G(3 * D, b16), dctx=G(D, b16), delta=E((B * H, T), f32), dpatch=E((B * N, D), b16),
                dlast=E((B, T, D), f32),
            )
        self._arena_key = key
        self._arenas[train] = (key, self.act, self.tmp, Mp)

    def _drop_arenas(self):
        self._arena_key = None
        self._arenas = {}
        self.act, self.tmp = {}, {}

    def _rope_tables(self, T: int):
        """Half-width cos / sin tables of RotaryPositionEmbedding (rope.py:36-56), computed on the host by the reference's
        own expression (so the table bits are the reference's) and kept on the device."""
        c = self.cfg
        key = (T, c.head_dim, float(c.rope_base), self.flat.device)
        if getattr(self, "_rope_key", None) != key:
            dim = c.head_dim
            inv_freq = 1.0 / (c.rope_base ** (torch.arange(0, dim, 2).float() / dim))
            freqs = torch.outer(torch.arange(T).type_as(inv_freq), inv_freq)
            self._rope = (freqs.cos().contiguous().to(self.flat.device), freqs.sin().contiguous().to(self.flat.device))
            self._rope_key = key
        return self._rope

    # ------------------------------------------------------------------ forward
    def _site(self, layer: int, which: int) -> int:
        return 1 + 4 * layer + which

    def forward(self, x: torch.Tensor, labels: Optional[torch.Tensor], training: bool, need_grad: bool,
                output_hidden_states: bool = False, output_attentions: bool = False, capture_ctx: Optional[list] = None):
        self._ensure_device_state()
        with vf.use_handle(self.handle()):
            return self._forward(x, labels, training, need_grad, output_hidden_states, output_attentions, capture_ctx)

    def _forward(self, x, labels, training, need_grad, output_hidden_states, output_attentions, capture_ctx):
        c = self.cfg
        if x.dim() != 2 or x.shape[1] != c.image_size:
            raise ValueError(f"pixel_values must be [batch, {c.image_size}], got {tuple(x.shape)}")
        if not x.is_cuda:
            raise VitError("pixel_values must live on the GPU")
        if self.precision == "f32" and c.seq_len > 4096:
            raise ValueError(f"precision '32': the fp32 attention kernels take at most 4096 tokens (this model has "
                             f"{c.seq_len}); use train.precision 'bf16-mixed'")
        x = x.contiguous().to(torch.float32)
        B = x.shape[0]
        self._gen += 1
        T, D, Fd, H, L, N, P, S = c.seq_len, c.hidden_size, c.intermediate_size, c.num_attention_heads, \
            c.num_hidden_layers, c.num_patches, c.patch_size, c.stride
        dh, M = c.head_dim, B * T
        self._ensure_arena(B, need_grad)
        Mp = self._Mp
        a = self.act
        ph = c.hidden_dropout_prob if training else 0.0
        pa = c.attention_probs_dropout_prob if training else 0.0
        if training:
            self.step_counter += 1
        seed = (self.base_seed + 0x9E3779B97F4A7C15 * self.step_counter) & 0xFFFFFFFFFFFFFFFF
        scale = dh ** -0.5
        eps = c.layer_norm_eps
        e = "vit.embeddings."

        # --- embeddings: unfold -> projection (+bias) into token rows 1..N -> CLS / pos-emb / dropout
        vf.unfold_cast(x, P, S, N, out=a["patches"])  # bf16 or f32 patches, per the arena dtype
        wp = self.w16(e + "patch_embeddings.projection.weight").view(D, P)
        x0 = a["x"][0]
        vf.gemm(a["patches"], wp, M=B * N, N=D, K=P, out=x0.view(M, D), bias=self.p(e + "patch_embeddings.projection.bias"),
                row_map=(N, T, 1))
        pos = self.p(e + "position_embeddings").view(T, D) if c.pos_encoding_type == "learned" else None
        vf.embed_finish(x0, self.p(e + "cls_token").view(D), pos, dropout=(ph, seed, 0))

        atts = [] if output_attentions else None
        rope = self._rope_tables(T) if c.pos_encoding_type == "rope" else None
        # The two "dropout(Linear(.)) + residual" sums of a layer (HF ViTLayer / ViTOutput) are formed by the LayerNorm
        # that consumes them: the projection GEMM writes y = dropout(acc + bias) in the activation dtype, the LayerNorm
        # pass reads the f32 stream and y, writes the new stream and its normalised operand (vit_layernorm_fwd_residual).
        y = a["y"]
        for i in range(L):
            j = i if need_grad else 0
            pre = f"vit.encoder.layer.{i}."
            xin = a["x"][i].view(M, D)
            if i == 0:
                self._ln(xin, pre + "layernorm_before", a["h1"][j], a["mean1"][j], a["rstd1"][j])
            else:
                self._ln_res(a["x1"][jprev], y, xin, pre + "layernorm_before", a["h1"][j], a["mean1"][j], a["rstd1"][j])
            # vit_with_rope.py:58-60: q, k rotated per head before the scores -- inside the projection's epilogue where the
            # kernel has one (f32 values, one rounding), by a vit_rope_qk pass behind it otherwise (the library decides)
            vf.gemm(a["h1"][j], self._qkv16(i), M=Mp, N=3 * D, K=D, out=a["qkv"][j], bias=self._qkv_bias(i, self.flat),
                    rope=None if rope is None else (rope[0], rope[1], T, dh, 2 * D))
            vf.attention_fwd(a["qkv"][j], B, H, T, dh, scale, dropout=(pa, seed, self._site(i, 0)), ctx=a["ctx"][j],
                             lse=a["lse"][j], ctx_lo=a["ctx_lo"][j])
            if output_attentions:
                atts.append(vf.attention_probs(a["qkv"][j], B, H, T, dh, scale))
            if capture_ctx is not None:  # per-layer context for forward hooks on the attention modules
                capture_ctx.append(a["ctx"][j].view(B, T, D).clone())
            vf.gemm(a["ctx"][j], self.w16(pre + "attention.output.dense.weight"), M=Mp, N=D, K=D, out=y,
                    bias=self.p(pre + "attention.output.dense.bias"), dropout=(ph, seed, self._site(i, 1)))
            self._ln_res(xin, y, a["x1"][j], pre + "layernorm_after", a["h2"][j], a["mean2"][j], a["rstd2"][j])
            vf.gemm(a["h2"][j], self.w16(pre + "intermediate.dense.weight"), M=Mp, N=Fd, K=D, out=a["g"][j],
                    bias=self.p(pre + "intermediate.dense.bias"), act=vf.ACT_GELU_GRAD if need_grad else ACT_GELU,
                    aux_out=a["u"][j] if need_grad else None)  # a["u"] holds gelu'(pre-activation) for the backward
            vf.gemm(a["g"][j], self.w16(pre + "output.dense.weight"), M=Mp, N=D, K=Fd, out=y,
                    bias=self.p(pre + "output.dense.bias"), dropout=(ph, seed, self._site(i, 2)))
            jprev = j
        if L > 0:
            self._ln_res(a["x1"][jprev], y, a["x"][L].view(M, D), "vit.layernorm", a["last"].view(M, D), a["meanF"],
                         a["rstdF"])
        else:
            self._ln(a["x"][L].view(M, D), "vit.layernorm", a["last"].view(M, D), a["meanF"], a["rstdF"])
        hn = self.layout.head
        if labels is not None:
            labels = labels.to(self.flat.device)
            labels = labels.contiguous().to(torch.int64 if self.loss_kind == LOSS_CE else torch.float32)
            n_expected = B if self.loss_kind == LOSS_CE else B * c.num_labels
            if labels.numel() != n_expected:
                raise ValueError(f"labels has {labels.numel()} elements, expected {n_expected}")
        logits, loss = vf.head_loss_fwd(a["last"], self.p(hn + ".weight"), self.p(hn + ".bias"), labels, self.loss_kind)
        self._last = dict(B=B, seed=seed, ph=ph, pa=pa, labels=labels, logits=logits, gen=self._gen, grad=need_grad)
        hs = [t.clone() for t in a["x"]] if output_hidden_states else None
        return loss, logits, hs, atts

    def saved_attentions(self):
        """Attention probabilities (before dropout) of every layer of the LAST need_grad forward, from its saved qkv."""
        st, c = self._last, self.cfg
        if st is None or not st["grad"] or st["gen"] != self._gen:
            raise VitError("saved_attentions(): the last forward did not keep per-layer activations")
        with vf.use_handle(self.handle()):
            return [vf.attention_probs(self.act["qkv"][i], st["B"], c.num_attention_heads, c.seq_len, c.head_dim,
                                       c.head_dim ** -0.5) for i in range(c.num_hidden_layers)]

    def _ln(self, x, name, out, mean, rstd):
        vf.layernorm_fwd(x, self.p(name + ".weight"), self.p(name + ".bias"), self.cfg.layer_norm_eps, out=out,
                         mean=mean, rstd=rstd)

    def _ln_res(self, x, delta, xsum, name, out, mean, rstd):
        vf.layernorm_fwd_residual(x, delta, xsum, self.p(name + ".weight"), self.p(name + ".bias"),
                                  self.cfg.layer_norm_eps, out=out, mean=mean, rstd=rstd)

    # ------------------------------------------------------------------ weight gradients beside the data path
    def _dw(self, *args, reads=(), **kw):
        """A weight-gradient GEMM (dW = dY^T X, deterministic split-K).  Its only consumer is the optimizer at the end of the
        step, so it is enqueued on a second HIP stream (own vit_handle = own split-K workspace) right after its operands are
        complete and runs beside whatever continues the chain on the main stream.  What that buys: the CUs a kernel of the main
        stream leaves idle (partial rounds of the N = 768 GEMMs, the tail of a LayerNorm pass) get dW workgroups -- a GEMM
        workgroup takes a whole CU (128 KiB of LDS, 2 x 232 of a SIMD's 512 VGPRs), so nothing shares a CU with it; measured
        0.1-0.2 ms per step.
        `reads` names the scratch buffers the GEMM reads; `_before_write(name)` makes the main stream wait for the last such
        reader before a kernel overwrites the buffer (one layer later, so the wait is normally already satisfied)."""
        if self.side_stream is None:
            return vf.gemm(*args, **kw)
        main = torch.cuda.current_stream(self.flat.device)
        self.side_stream.wait_stream(main)
        with torch.cuda.stream(self.side_stream), vf.use_handle(self._side_handle):
            out = vf.gemm(*args, **kw)
            if reads:
                ev = torch.cuda.Event()
                ev.record(self.side_stream)
                for name in reads:
                    self._side_reads[name] = ev
        return out

    def _before_write(self, *names):
        if self.side_stream is None:
            return
        main = torch.cuda.current_stream(self.flat.device)
        for name in names:
            ev = self._side_reads.pop(name, None)
            if ev is not None:
                main.wait_event(ev)

    def _join_side(self):
        if self.side_stream is not None:
            torch.cuda.current_stream(self.flat.device).wait_stream(self.side_stream)
            self._side_reads.clear()

    def _notify(self, lo: int, hi: int):
        """Hand grads[lo:hi] to the gradient exchange.  With the second stream on, the collective is enqueued from THAT stream
        (after it has caught up with the main stream's position): the exchange then waits for the weight-gradient GEMMs
        without the main stream having to join them, and the data path of the next layer keeps running."""
        cb = self.grad_ready_cb
        if cb is None:
            return
        if self.side_stream is None:
            return cb(lo, hi)
        self.side_stream.wait_stream(torch.cuda.current_stream(self.flat.device))
        with torch.cuda.stream(self.side_stream):
            cb(lo, hi)

    def _want_side(self) -> bool:
        """`overlap_dw == "auto"`: second stream iff the balanced grid of this batch's [M, D] x [D, D] products leaves at least
        30 % of the CUs without a workgroup (the dW GEMMs then run on CUs nobody uses).  Measured on MI355X with the half-tile
        tail launch off (r05, tools/stream_rule.sh; the chip is power-limited, so CU-time spent is what counts): ViT-B at B = 64 /
        128 (41 % idle) two streams +3.5 %, ViT-L at B = 32 (43 %) +3.7 %; ViT-B at B = 192 ... 512 (7-23 % idle) within +-0.6 %,
        and at the benchmarked B = 256 one stream wins by 0.5-1 % -- there the hand-off gaps between the streams (0.6 ms per
        step, profiles/r05_a_gaps.txt) cost more than the overlap returns.  Shapes off the 256-aligned core: two streams."""
        if self.overlap_dw != "auto":
            return bool(self.overlap_dw)
        D = self.cfg.hidden_size
        Mp = getattr(self, "_Mp", 0)
        if D % 256 or not Mp or Mp % 256:
            return True
        cus = max(1, torch.cuda.get_device_properties(self.flat.device).multi_processor_count - max(0, self.reserve_cus))
        tiles = (Mp // 256) * (D // 256)
        rounds = -(-tiles // cus)
        return -(-tiles // rounds) <= 0.7 * cus

    def _setup_side(self):
        from . import _cabi

        on = self.precision == "bf16" and self._want_side()
        if on and self.side_stream is None:
            import os

            self.side_stream = torch.cuda.Stream(device=self.flat.device, priority=int(os.environ.get("VIT_SIDE_STREAM_PRIORITY", "0")))
            if self._side_handle is None:
                self._side_handle = _cabi.Handle(self.flat.device.index if self.flat.device.index is not None
                                                 else torch.cuda.current_device())
                self._side_handle.set_option("reserve_cus", self.reserve_cus)
        elif not on:
            self.side_stream = None

    # ------------------------------------------------------------------ backward
    def backward(self, dloss: torch.Tensor, need_dx: bool = False, gen: Optional[int] = None):
        with vf.use_handle(self.handle()):
            return self._backward(dloss, need_dx, gen)

    def _backward(self, dloss: torch.Tensor, need_dx: bool = False, gen: Optional[int] = None):
        """Fill self.grads (every trainable slice exactly once) for the last forward; calls grad_ready_cb(lo, hi) as
        each bucket of the flat gradient buffer is complete (used to overlap the RCCL all-reduce).  `gen`: the forward this
        backward belongs to (ViTEngine._gen at that time); the arena holds one forward's activations, so a later forward
        (a second loss term, an eval pass, a viz hook) makes them unavailable and that is an error, not a wrong gradient."""
        st = self._last
        if st is None or st["labels"] is None:
            raise VitError("backward without a preceding forward(labels=...)")
        if not st["grad"] or (gen is not None and gen != st["gen"]) or st["gen"] != self._gen:
            raise VitError("backward: the activations of that forward are gone -- another forward ran on this model in "
                           "between (the engine keeps the activations of ONE forward; run backward before the next forward)")
        c = self.cfg
        a, t = self.act, self.tmp
        B, seed, ph, pa = st["B"], st["seed"], st["ph"], st["pa"]
        T, D, Fd, H, L, N, P = c.seq_len, c.hidden_size, c.intermediate_size, c.num_attention_heads, \
            c.num_hidden_layers, c.num_patches, c.patch_size
        dh, M = c.head_dim, B * T
        Mp = self._Mp
        scale = dh ** -0.5
        rope = self._rope_tables(T) if c.pos_encoding_type == "rope" else None
        hn = self.layout.head
        dloss = dloss.reshape(1).to(torch.float32).contiguous()

        vf.head_loss_bwd(a["last"], self.p(hn + ".weight"), st["logits"], st["labels"], dloss, self.loss_kind,
                         dlast=t["dlast"], dW=self.g(hn + ".weight"), db=self.g(hn + ".bias"))
        self._setup_side()
        dx, dx_other = t["dxa"], t["dxb"]
        # every LayerNorm backward below also emits dy = dropout_mask * dx (bf16) and its column sums: the gradient of
        # the Linear output underneath the next "dropout(.) + residual" going down, and that Linear's bias gradient
        if L == 0:  # no encoder layer: the final LayerNorm sits directly on the embeddings
            vf.layernorm_bwd(t["dlast"].view(M, D), a["x"][0].view(M, D), self.p("vit.layernorm.weight"), a["meanF"],
                             a["rstdF"], dres=None, dx=dx, dgamma=self.g("vit.layernorm.weight"),
                             dbeta=self.g("vit.layernorm.bias"))
        else:
            last_pre = f"vit.encoder.layer.{L - 1}."
            vf.layernorm_bwd_fused(t["dlast"].view(M, D), a["x"][L].view(M, D), self.p("vit.layernorm.weight"), a["meanF"],
                                   a["rstdF"], None, dx, self.g("vit.layernorm.weight"), self.g("vit.layernorm.bias"),
                                   t["dy"], self.g(last_pre + "output.dense.bias"), (ph, seed, self._site(L - 1, 2)))
        self._notify(self.layout.tail_start, self.layout.n_trainable)
        for i in reversed(range(L)):
            pre = f"vit.encoder.layer.{i}."
            # x2 = dropout(g W2^T + b2) + x1      (t["dy"] = mask * dx and db2 were produced by the LN backward above)
            self._dw(t["dy"], a["g"][i], M=D, N=Fd, K=Mp, a_trans=True, b_trans=True, out=self.g(pre + "output.dense.weight"),
                     split_k=-1, reads=("dy",))
            self._before_write("dU")
            vf.gemm(t["dy"], self.w16(pre + "output.dense.weight"), M=Mp, N=Fd, K=D, b_trans=True, out=t["dU"],
                    act=vf.ACT_MUL_AUX, aux_in=a["u"][i], colsum_out=self.g(pre + "intermediate.dense.bias"))
            self._dw(t["dU"], a["h2"][i], M=Fd, N=D, K=Mp, a_trans=True, b_trans=True,
                     out=self.g(pre + "intermediate.dense.weight"), split_k=-1, reads=("dU",))
            vf.gemm(t["dU"], self.w16(pre + "intermediate.dense.weight"), M=Mp, N=D, K=Fd, b_trans=True, out=t["dh"])
            # x1 = dropout(ctx Wo^T + bo) + x:  LN2 backward -> dx1, and dya = mask * dx1 with dbo
            # this pass writes the OTHER dy buffer: the FC2 weight gradient still reading t["dy"] keeps running beside it
            self._before_write("dy2")
            vf.layernorm_bwd_fused(t["dh"], a["x1"][i], self.p(pre + "layernorm_after.weight"), a["mean2"][i],
                                   a["rstd2"][i], dx, dx_other, self.g(pre + "layernorm_after.weight"),
                                   self.g(pre + "layernorm_after.bias"), t["dy2"],
                                   self.g(pre + "attention.output.dense.bias"), (ph, seed, self._site(i, 1)))
            dx, dx_other = dx_other, dx
            self._dw(t["dy2"], a["ctx"][i], M=D, N=D, K=Mp, a_trans=True, b_trans=True,
                     out=self.g(pre + "attention.output.dense.weight"), split_k=-1, reads=("dy2",))
            vf.gemm(t["dy2"], self.w16(pre + "attention.output.dense.weight"), M=Mp, N=D, K=D, b_trans=True, out=t["dctx"])
            self._before_write("dqkv")
            # the QKV bias gradient = column sums of dqkv: taken by the attention kernels on their way out, unless RoPE
            # sits in between (then after the inverse rotation, by the column-sum kernel)
            vf.attention_bwd(a["qkv"][i], a["ctx"][i], t["dctx"], a["lse"][i], B, H, T, dh, scale,
                             dropout=(pa, seed, self._site(i, 0)), dqkv=t["dqkv"], delta=t["delta"],
                             colsum_out=None if rope is not None else self._qkv_bias(i, self.grads), ctx_lo=a["ctx_lo"][i])
            if rope is not None:  # gradient wrt the un-rotated q, k: the inverse rotation
                vf.rope_qk(t["dqkv"], rope[0], rope[1], T, H, dh, inverse=True)
                vf.colsum(t["dqkv"], out=self._qkv_bias(i, self.grads))
            self._dw(t["dqkv"], a["h1"][i], M=3 * D, N=D, K=Mp, a_trans=True, b_trans=True, out=self._qkv_wgrad(i),
                     split_k=-1, reads=("dqkv",))
            vf.gemm(t["dqkv"], self._qkv16(i), M=Mp, N=D, K=3 * D, b_trans=True, out=t["dh"])
            self._before_write("dy")  # the LayerNorm backward below rewrites t["dy"] (read by this layer's FC2 weight gradient)
            if i > 0:
                # LN1 backward -> dx (input of this layer = output of layer i-1), plus layer i-1's FC2 pieces
                prev = f"vit.encoder.layer.{i - 1}."
                vf.layernorm_bwd_fused(t["dh"], a["x"][i].view(M, D), self.p(pre + "layernorm_before.weight"),
                                       a["mean1"][i], a["rstd1"][i], dx, dx_other,
                                       self.g(pre + "layernorm_before.weight"), self.g(pre + "layernorm_before.bias"),
                                       t["dy"], self.g(prev + "output.dense.bias"), (ph, seed, self._site(i - 1, 2)))
            else:
                vf.layernorm_bwd(t["dh"], a["x"][i].view(M, D), self.p(pre + "layernorm_before.weight"), a["mean1"][i],
                                 a["rstd1"][i], dres=dx, dx=dx_other, dgamma=self.g(pre + "layernorm_before.weight"),
                                 dbeta=self.g(pre + "layernorm_before.bias"))
            dx, dx_other = dx_other, dx
            self._notify(*self.layout.layer_ranges[i])
        e = "vit.embeddings."
        dpos = self.g(e + "position_embeddings").view(T, D) if c.pos_encoding_type == "learned" else None
        vf.embed_finish_bwd(dx.view(B, T, D), self.g(e + "cls_token").view(D), dpos, dropout=(ph, seed, 0),
                            dpatch=t["dpatch"])  # dtype follows the buffer
        vf.colsum(t["dpatch"], out=self.g(e + "patch_embeddings.projection.bias"))
        vf.gemm(t["dpatch"], a["patches"], M=D, N=P, K=B * N, a_trans=True, b_trans=True,
                out=self.g(e + "patch_embeddings.projection.weight").view(D, P), split_k=-1)
        self._join_side()
        self._notify(self.layout.embed_start, self.layout.embed_end)
        if need_dx:
            # gradient wrt the signal itself (a trainable input preprocessor sits in front): dpatches = dpatch Wp, then the
            # overlap-add that undoes the tokenizer's unfold
            S = c.stride
            wp = self.w16(e + "patch_embeddings.projection.weight").view(D, P)
            dpat = vf.gemm(t["dpatch"], wp, M=B * N, N=P, K=D, b_trans=True, out_dtype=torch.float32)
            return vf.fold_add(dpat, B, c.image_size, P, S, N)
        return None
