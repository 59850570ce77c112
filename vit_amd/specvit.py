"""`MyViT`: drop-in for the reference's model class (src/models/specvit.py:16-115) on MI355X.

Same constructor arguments, `forward(pixel_values, labels=None, output_attentions=None, output_hidden_states=None,
return_dict=None)` returning an object with `.loss / .logits / .hidden_states / .attentions`, same `.name`,
`.loss_name`, `compute_loss`, `log_outputs`, `set_preprocessor_trainable`, the same ValueError for a bad task_type, and
a `state_dict()` whose keys are those of the reference's checkpoints (transformers 4.56 module names), so checkpoints
move in both directions.  All arithmetic runs in libvit_amd.so; parameters are views of one flat fp32 buffer.
"""
from __future__ import annotations

from dataclasses import dataclass, fields
from typing import Any, Optional, Tuple

import torch
import torch.nn as nn

from ._cabi import VitError
from .config import ViTConfig
from .engine import ViTEngine

__all__ = ["MyViT", "SequenceClassifierOutput", "build_model_name"]


@dataclass
class SequenceClassifierOutput:
    """Field-compatible with transformers.modeling_outputs.SequenceClassifierOutput (attribute, key and index access)."""

    loss: Optional[torch.Tensor] = None
    logits: Optional[torch.Tensor] = None
    hidden_states: Optional[Tuple[torch.Tensor, ...]] = None
    attentions: Optional[Tuple[torch.Tensor, ...]] = None

    def to_tuple(self):
        return tuple(getattr(self, f.name) for f in fields(self) if getattr(self, f.name) is not None)

    def __getitem__(self, k):
        return getattr(self, k) if isinstance(k, str) else self.to_tuple()[k]

    def get(self, k, default=None):
        v = getattr(self, k, None)
        return default if v is None else v

    def keys(self):
        return [f.name for f in fields(self) if getattr(self, f.name) is not None]


def build_model_name(config, model_prefix: str = "ViT", full_config: dict = None) -> str:
    """Run name `{prefix}_p{patch}_h{hidden}_l{layers}_a{heads}_s{stride}_p{proj_fn}[_nz{noise digits}]` -- the rule of
    src/models/model_utils.py:9-45 (pinned by tests/golden/names.json).  The stride tag is the explicit `stride_size` when
    one is set (truthy), else the `stride_ratio`; a positive noise level is appended with its decimal point removed."""
    explicit = getattr(config, "stride_size", None)
    fields = [
        ("p", config.patch_size), ("h", config.hidden_size), ("l", config.num_hidden_layers),
        ("a", config.num_attention_heads), ("s", int(explicit) if explicit else config.stride_ratio), ("p", config.proj_fn),
    ]
    name = "_".join([model_prefix] + [f"{tag}{value}" for tag, value in fields])
    level = ((full_config or {}).get("noise") or {}).get("noise_level", 0)
    if level and level > 0:
        name += "_nz" + str(level).replace(".", "")
    return name


class _Node(nn.Module):
    """Structural container: only exists so parameters get the reference's dotted names."""


def _trunc_normal_(t: torch.Tensor, std: float):
    nn.init.trunc_normal_(t, mean=0.0, std=std, a=-2.0, b=2.0)


class _ViTFunction(torch.autograd.Function):
    """Whole-model autograd node: forward/backward are the engine's kernel sequences; parameter gradients come back as
    views of the engine's flat gradient buffer (no copies when .grad is None, i.e. zero_grad(set_to_none=True))."""

    @staticmethod
    def forward(ctx, model, x, labels, training, *params):
        eng = model.engine
        loss, logits, _, _ = eng.forward(x, labels, training=training, need_grad=True)
        ctx.model = model
        ctx.gen = eng._gen
        ctx.need_dx = bool(x.requires_grad)  # a trainable input preprocessor in front of the ViT
        ctx.mark_non_differentiable(logits)
        return loss, logits

    @staticmethod
    def backward(ctx, dloss, _dlogits):
        model = ctx.model
        eng = model.engine
        dx = eng.backward(dloss, need_dx=ctx.need_dx, gen=ctx.gen)
        grads = []
        for name, p in zip(model._param_names, model._param_list):
            if not p.requires_grad or name.startswith("vit.pooler."):
                grads.append(None)  # the pooler output is never used (specvit.py:78): no gradient, as in the reference
            else:
                grads.append(eng.g(name))
        return (None, dx, None, None, *grads)


class MyViT(nn.Module):
    """Vision Transformer over 1-D spectra (reference: `MyViT(ViTPreTrainedModel, BaseModel)`)."""

    config_class = ViTConfig

    def __init__(self, config: ViTConfig, loss_name: str = "", model_name: str = "ViT",
                 preprocessor: Optional[nn.Module] = None, full_config: dict = None) -> None:
        super().__init__()
        self.config = config
        self.task_type = config.task_type
        if self.task_type not in ("cls", "reg"):
            raise ValueError(f"Unsupported task_type '{self.task_type}'")  # specvit.py:55
        self.preprocessor = preprocessor  # specvit.py:43, 72-73: applied to pixel_values before the ViT
        self.engine = ViTEngine(config, loss_name=loss_name)
        self._loss_name = self.engine.loss_name
        self._model_name = build_model_name(config, model_name, full_config=full_config)
        self._build_tree()
        self.init_weights()

    # ------------------------------------------------------------------ structure
    def _build_tree(self):
        lay = self.engine.layout
        self._param_names = list(lay.state_order)
        plist = []
        for name in self._param_names:
            parts = name.split(".")
            node = self
            for part in parts[:-1]:
                if part not in node._modules:
                    node.add_module(part, _Node())
                node = node._modules[part]
            prm = nn.Parameter(lay.view(self.engine.flat, name), requires_grad=True)
            node.register_parameter(parts[-1], prm)
            plist.append(prm)
        self._param_list = plist
        self.engine.version_fn = self._version_signature

    def _version_signature(self) -> int:
        # in-place updates by torch optimizers / load_state_dict / p.copy_() bump the parameter's version counter
        return sum(p._version for p in self._param_list) + self.engine.flat._version

    def _rebind_views(self):
        lay = self.engine.layout
        for name, prm in zip(self._param_names, self._param_list):
            prm.data = lay.view(self.engine.flat, name)
            prm.grad = None

    def _apply(self, fn, recurse=True):
        """.to()/.cuda()/.cpu(): move the ONE flat buffer and re-point every parameter at its slice."""
        new_flat = fn(self.engine.flat)
        self.engine.rebind(new_flat)
        self._rebind_views()
        if self.preprocessor is not None:
            self.preprocessor._apply(fn)
        return self

    def load_state_dict(self, state_dict, strict: bool = True, assign: bool = False):
        out = super().load_state_dict(state_dict, strict=strict, assign=False)
        self.engine._shadow_version = -1
        return out

    @torch.no_grad()
    def init_weights(self):
        """HF `_init_weights` as the reference triggers it (specvit.py:57): Linear / Conv weights ~ trunc_normal(0,
        initializer_range), biases 0, LayerNorm 1/0.  cls_token and learned position_embeddings belong to
        SpectraEmbeddings (not an HF ViTEmbeddings), so they keep their torch.randn values (embedding.py:47,62-64)."""
        std = self.config.initializer_range
        for name, p in zip(self._param_names, self._param_list):
            if name.endswith("cls_token") or name.endswith("position_embeddings"):
                p.copy_(torch.randn(p.shape))
            elif "layernorm" in name:
                p.fill_(1.0 if name.endswith("weight") else 0.0)
            elif name.endswith("bias"):
                p.zero_()
            else:
                _trunc_normal_(p, std)

    # ------------------------------------------------------------------ reference surface
    @property
    def name(self):
        return self._model_name

    @property
    def loss_name(self):
        return self._loss_name

    def forward(self, pixel_values, labels=None, output_attentions=None, output_hidden_states=None, return_dict=None):
        eng = self.engine
        training = self.training
        if self.preprocessor is not None:
            pixel_values = self.preprocessor(pixel_values)
        # forward hooks on `vit.encoder.layer.N.attention.attention` (what the reference's viz callback registers to read
        # attention maps, src/viz/viz_callback.py:183-214, 231-235): the kernels have no per-layer Python forward to hook,
        # so when such hooks exist the maps are produced anyway and the hooks are called with HF's (context, probs) output
        hooked = self._hooked_attention_layers()
        if hooked:
            output_attentions_user, output_attentions = output_attentions, True
        want_grad = torch.is_grad_enabled() and labels is not None and any(p.requires_grad for p in self._param_list)
        ctxs = [] if hooked else None
        if want_grad:
            # the loss stays differentiable whatever else is asked for (as in the reference); hidden states / attention maps
            # are read back from the activations that forward kept (maps: the probabilities before dropout)
            loss, logits = _ViTFunction.apply(self, pixel_values, labels, training, *self._param_list)
            hs = [t.clone() for t in eng.act["x"]] if output_hidden_states else None
            atts = eng.saved_attentions() if output_attentions else None
            if hooked:
                B, T, D = pixel_values.shape[0], self.config.seq_len, self.config.hidden_size
                ctxs = [c.view(B, T, D) for c in eng.act["ctx"]]
        else:
            with torch.no_grad():
                loss, logits, hs, atts = eng.forward(pixel_values, labels, training=training, need_grad=False,
                                                     output_hidden_states=bool(output_hidden_states),
                                                     output_attentions=bool(output_attentions), capture_ctx=ctxs)
        if hooked:
            for i in hooked:
                node = self._attention_node(i)
                for hook in list(node._forward_hooks.values()):
                    hook(node, (), (ctxs[i].detach(), atts[i]))
            if not output_attentions_user:
                atts = None
        out = SequenceClassifierOutput(loss=loss, logits=logits,
                                       hidden_states=tuple(hs) if hs is not None else None,
                                       attentions=tuple(atts) if atts is not None else None)
        if return_dict is False:
            return out.to_tuple()
        return out

    def _attention_node(self, i: int):
        return self._modules["vit"]._modules["encoder"]._modules["layer"]._modules[str(i)]._modules["attention"]._modules["attention"]

    def _hooked_attention_layers(self):
        return [i for i in range(self.config.num_hidden_layers) if self._attention_node(i)._forward_hooks]

    def compute_loss(self, *args, **kwargs):
        return self.forward(*args, **kwargs).loss

    def log_outputs(self, outputs, log_fn=print, stage: str = ""):
        loss = outputs.get("loss") if isinstance(outputs, dict) else getattr(outputs, "loss", None)
        if loss is not None:
            log_fn({f"{self.loss_name}_loss": loss})

    def set_preprocessor_trainable(self, trainable: bool) -> None:
        """specvit.py:118-130: freeze / unfreeze the input preprocessor."""
        if self.preprocessor is None:
            return
        if hasattr(self.preprocessor, "set_qk_trainable"):
            self.preprocessor.set_qk_trainable(trainable)
        elif hasattr(self.preprocessor, "freeze"):
            self.preprocessor.freeze(not trainable)
        else:
            for param in self.preprocessor.parameters():
                param.requires_grad = trainable

    def set_precision(self, precision) -> str:
        """'32' (reference default; fp32-class kernels) or 'bf16-mixed' (bf16 MFMA operands): see ViTEngine."""
        if self.preprocessor is not None and hasattr(self.preprocessor, "set_precision"):
            self.preprocessor.set_precision(precision)
        return self.engine.set_precision(precision)

    # ------------------------------------------------------------------ extras used by the build's own trainer
    def flat_parameters(self) -> torch.Tensor:
        return self.engine.flat

    def flat_gradients(self) -> torch.Tensor:
        return self.engine.grads
