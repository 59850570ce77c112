"""ORACLE tooling (test infrastructure): generate tests/golden/*.npz from the REFERENCE's own modules.

Runs only in the build container (needs /root/reference); the GPU box never runs it.  It composes, exactly as
`MyViT` does (src/models/specvit.py:32-35, 68-94):

    vc  = src.models.builder.get_vit_config(cfg)            (reference)
    vit = transformers.ViTModel(vc)                         (installed 5.15; the reference pins 4.56, same math)
    vit.embeddings = src.models.embedding.SpectraEmbeddings(vc)   (reference)
    cls = vit(x)[0][:, 0, :];  logits = Linear(cls);  loss = MSE|L1|CE

`MyViT` itself cannot be constructed against transformers 5.x (it never calls post_init()), and
`lightning`/`h5py` are not installed, so those *absent third-party packages* are stubbed in-memory just far enough for
`import src.models` to succeed.  Nothing of the reference is copied: fixtures hold inputs and expected outputs only.

Every fixture is cross-checked against oracle/refvit.py (the restatement) before it is written, which is what pins
the oracle.  Usage:  python oracle/make_golden.py
"""
from __future__ import annotations

import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, ROOT)
REF = "/root/reference"

from oracle import refvit  # noqa: E402


def _import_reference():
    sys.path.insert(0, REF)

    def stub(name, **attrs):
        m = types.ModuleType(name)
        m.__dict__.update(attrs)
        sys.modules[name] = m
        return m

    if "lightning" not in sys.modules:
        L = stub("lightning", LightningModule=type("LightningModule", (object,), {}),
                 LightningDataModule=type("LightningDataModule", (object,), {}),
                 Trainer=type("Trainer", (object,), {}), Callback=type("Callback", (object,), {}),
                 seed_everything=lambda *a, **k: None)
        pt = stub("lightning.pytorch")
        cb = stub("lightning.pytorch.callbacks", Callback=type("Callback", (object,), {}))
        pt.callbacks = cb
        L.pytorch = pt
    if "h5py" not in sys.modules:
        stub("h5py")
    import transformers.models.vit.modeling_vit as mv
    if not hasattr(mv, "ViTSelfAttention"):  # name removed in transformers 5 (vit_with_rope.py:8 imports it)
        mv.ViTSelfAttention = mv.ViTAttention
    from src.models.builder import get_vit_config
    from src.models.embedding import SpectraEmbeddings
    from src.utils import make_dummy_spectra
    return get_vit_config, SpectraEmbeddings, make_dummy_spectra


def name_456_to_515(name: str) -> str:
    """transformers-4.56 state_dict name (what the reference's checkpoints hold) -> the installed 5.15 ViTModel name."""
    n = name
    assert n.startswith("vit.")
    n = n[4:]
    n = n.replace("encoder.layer.", "layers.")
    n = n.replace("attention.attention.query", "attention.q_proj")
    n = n.replace("attention.attention.key", "attention.k_proj")
    n = n.replace("attention.attention.value", "attention.v_proj")
    n = n.replace("attention.output.dense", "attention.o_proj")
    n = n.replace("intermediate.dense", "mlp.fc1")
    if ".output.dense" in n:
        n = n.replace("output.dense", "mlp.fc2")
    return n


def build_reference(cfg_dict, sd, get_vit_config, SpectraEmbeddings):
    from transformers import ViTModel

    vc = get_vit_config(cfg_dict)
    vc._attn_implementation = "eager"  # so attentions are returned (same math as sdpa, SURVEY 8c: diff 7e-7)
    vit = ViTModel(vc)
    vit.embeddings = SpectraEmbeddings(vc)
    own = vit.state_dict()
    mapped = {}
    head = {}
    for k, v in sd.items():
        if k.startswith("vit."):
            mapped[name_456_to_515(k)] = v.clone()
        else:
            head[k] = v.clone()
    missing = set(own.keys()) - set(mapped.keys())
    extra = set(mapped.keys()) - set(own.keys())
    assert not missing and not extra, (missing, extra)
    vit.load_state_dict(mapped, strict=True)
    hname = "classifier" if vc.task_type == "cls" else "regressor"
    lin = torch.nn.Linear(vc.hidden_size, vc.num_labels)
    lin.weight.data.copy_(head[hname + ".weight"])
    lin.bias.data.copy_(head[hname + ".bias"])
    return vc, vit, lin


def ref_forward(vc, vit, lin, rcfg, x, labels, training=False):
    vit.train(training)
    o = vit(x, output_hidden_states=True, output_attentions=True)
    cls = o.last_hidden_state[:, 0, :]
    logits = lin(cls)
    loss = refvit.loss_fn(rcfg, logits, labels)  # specvit.py:81-89 (three torch loss calls; restated in refvit.loss_fn)
    return o, logits, loss


def cfg_dict_for(rc: refvit.RefConfig, param: str = "log_g"):
    return {
        "model": dict(name="vit", task_type=rc.task_type, image_size=rc.image_size, patch_size=rc.patch_size,
                      hidden_size=rc.hidden_size, num_hidden_layers=rc.num_hidden_layers,
                      num_attention_heads=rc.num_attention_heads, stride_size=rc.stride_size, proj_fn=rc.proj_fn,
                      num_labels=rc.num_labels, pos_encoding_type=rc.pos_encoding_type,
                      max_position_embeddings=rc.max_position_embeddings, rope_base=rc.rope_base),
        "loss": {"name": rc.loss_name},
        "data": {"param": param},
    }


def rel(a, b):
    return float((a - b).norm() / (b.norm() + 1e-30))


def sample_idx(numel, k, seed):
    rng = np.random.Generator(np.random.PCG64(seed))
    return np.sort(rng.choice(numel, size=min(k, numel), replace=False)).astype(np.int64)


def make_one(tag, rc, param, batch, wseed, xseed, full_grads, x_override=None, steps=3):
    get_vit_config, SpectraEmbeddings, make_dummy_spectra = _import_reference()
    cfgd = cfg_dict_for(rc, param)
    # the reference's own config mapping must agree with the restatement
    vc0 = get_vit_config({k: dict(v) for k, v in cfgd.items()})
    rc2 = refvit.config_from_dict({k: dict(v) for k, v in cfgd.items()})
    assert (vc0.num_labels, vc0.hidden_size, vc0.intermediate_size, vc0.layer_norm_eps) == \
           (rc2.num_labels, rc2.hidden_size, rc2.intermediate_size, rc2.layer_norm_eps)
    assert rc2 == rc, (rc2, rc)
    sd = refvit.make_state_dict(rc, wseed)
    flux, error, labels = refvit.make_inputs(rc, batch, xseed)
    if x_override is not None:
        flux = x_override(make_dummy_spectra, flux)
    vc, vit, lin = build_reference(cfgd, sd, get_vit_config, SpectraEmbeddings)
    assert vit.embeddings.num_patches == rc.num_patches

    # ---- eval forward, fp32: reference vs restatement
    with torch.no_grad():
        o, logits, loss = ref_forward(vc, vit, lin, rc, flux, labels)
        mine = refvit.forward(rc, sd, flux, labels, output_hidden_states=True, output_attentions=True)
        tok_ref = vit.embeddings.patch_embeddings(flux)
    errs = {
        "tokens": rel(mine.tokens, tok_ref),
        "last": rel(mine.last_hidden_state, o.last_hidden_state),
        "logits": rel(mine.logits, logits),
        "loss": abs(float(mine.loss) - float(loss)) / (abs(float(loss)) + 1e-30),
    }
    for i, (a, b) in enumerate(zip(mine.hidden_states, o.hidden_states)):
        errs[f"hs{i}"] = rel(a, b)
    for i, (a, b) in enumerate(zip(mine.attentions, o.attentions)):
        errs[f"att{i}"] = rel(a, b)
    worst = max(errs.values())
    print(f"[{tag}] oracle-vs-reference eval fwd: worst rel err {worst:.3e}")
    assert worst < 2e-5, errs

    # ---- the reference under bf16 autocast (precision='bf16-mixed', basemodule.py:233) for the bf16 criterion
    with torch.no_grad(), torch.autocast("cpu", dtype=torch.bfloat16):
        ob, logits_b, loss_b = ref_forward(vc, vit, lin, rc, flux, labels)
    print(f"[{tag}] reference bf16-autocast vs fp32: logits rel {rel(logits_b.float(), logits):.3e} "
          f"last rel {rel(ob.last_hidden_state.float(), o.last_hidden_state):.3e}")

    # ---- gradients with dropout off (train mode arithmetic == eval arithmetic at p=0): reference autograd
    for p in list(vit.parameters()) + list(lin.parameters()):
        p.grad = None
    vit.eval()
    o2, logits2, loss2 = ref_forward(vc, vit, lin, rc, flux, labels)
    loss2.backward()
    ref_grads = {}
    inv = {name_456_to_515(k): k for k in sd if k.startswith("vit.")}
    for k, p in vit.named_parameters():
        ref_grads[inv[k]] = None if p.grad is None else p.grad.detach().clone()
    ref_grads[rc.head_name + ".weight"] = lin.weight.grad.detach().clone()
    ref_grads[rc.head_name + ".bias"] = lin.bias.grad.detach().clone()

    tr = refvit.RefTrainer(rc, sd, training=False)
    o3 = refvit.forward(rc, tr.params, flux, labels)
    o3.loss.backward()
    gworst = 0.0
    for k, p in tr.params.items():
        g = ref_grads[k]
        if g is None or float(g.norm()) < 1e-6:
            # pooler: no grad at all; key.bias: analytically zero (softmax is shift-invariant), rounding noise only
            assert p.grad is None or float(p.grad.norm()) < 1e-6, k
            continue
        gworst = max(gworst, rel(p.grad, g))
    print(f"[{tag}] oracle-vs-reference grads: worst rel err {gworst:.3e}")
    assert gworst < 5e-4, gworst

    # ---- 1..steps optimisation steps (clip 0.5, AdamW lr 1e-3 wd 0), dropout off
    tr = refvit.RefTrainer(rc, sd, training=False)
    losses, gnorms, after = [], [], {}
    for s in range(steps):
        losses.append(tr.step(flux, labels))
        gnorms.append(tr.last_grad_norm)
        if s in (0, steps - 1):
            after[s + 1] = tr.state_dict()

    out = dict(
        wseed=np.int64(wseed), xseed=np.int64(xseed), batch=np.int64(batch),
        flux=flux.numpy(), labels=labels.numpy(),
        tokens=tok_ref.numpy(), last_hidden_state=o.last_hidden_state.numpy(), logits=logits.numpy(),
        loss=np.float32(loss), attn0=o.attentions[0].numpy(), attn_last=o.attentions[-1].numpy(),
        hidden_states=np.stack([h.numpy() for h in o.hidden_states]),
        bf16_logits=logits_b.float().numpy(), bf16_last_hidden_state=ob.last_hidden_state.float().numpy(),
        bf16_loss=np.float32(loss_b.float()),
        step_losses=np.asarray(losses, np.float64), step_grad_norms=np.asarray(gnorms, np.float64),
        weight_checksum=np.float64(sum(float(v.double().sum()) for v in sd.values())),
    )
    names = list(sd.keys())
    out["param_names"] = np.asarray(names)
    out["grad_norms"] = np.asarray([0.0 if ref_grads[k] is None else float(ref_grads[k].norm()) for k in names])
    for i, k in enumerate(names):
        g = ref_grads[k]
        if g is None:
            continue
        if full_grads:
            out[f"grad/{k}"] = g.numpy()
            out[f"after1/{k}"] = after[1][k].numpy()
            out[f"after{steps}/{k}"] = after[steps][k].numpy()
        else:
            idx = sample_idx(g.numel(), 64, 1000 + i)
            out[f"gidx/{k}"] = idx
            out[f"grad/{k}"] = g.flatten()[idx].numpy()
            out[f"after1/{k}"] = after[1][k].flatten()[idx].numpy()
            out[f"after{steps}/{k}"] = after[steps][k].flatten()[idx].numpy()
    path = os.path.join(ROOT, "tests", "golden", f"{tag}.npz")
    np.savez_compressed(path, **out)
    print(f"[{tag}] wrote {path} ({os.path.getsize(path) / 1024:.0f} KiB)")


def make_rope():
    """Rotary position embedding: the reference's RotaryPositionEmbedding (src/models/rope.py) run on
    random q/k pins the restated tables and rotation bit for bit; the model-level fixture p1 then comes from the
    restatement with the REFERENCE's module doing the rotation inside it (ViTSelfAttentionWithRoPE subclasses the
    transformers-4.56 attention class and cannot be constructed against the installed 5.15: SURVEY 8c)."""
    _import_reference()
    from src.models.rope import RotaryPositionEmbedding

    out = {}
    rng = np.random.Generator(np.random.PCG64(77))
    for tag, (B, H, T, dh, base, maxlen) in dict(a=(2, 3, 37, 16, 10000.0, 512), b=(1, 1, 520, 64, 500.0, 512)).items():
        q = torch.from_numpy(rng.standard_normal((B, H, T, dh)).astype(np.float32))
        k = torch.from_numpy(rng.standard_normal((B, H, T, dh)).astype(np.float32))
        mod = RotaryPositionEmbedding(dim=dh, max_seq_len=maxlen, base=base)
        qr, kr = mod.forward_qk(q, k)
        cos, sin = refvit.rope_tables(dh, max(T, maxlen), base)
        assert torch.equal(cos, mod.cos_cached) and torch.equal(sin, mod.sin_cached)
        assert torch.equal(refvit.apply_rope(q, cos, sin), qr) and torch.equal(refvit.apply_rope(k, cos, sin), kr)
        out.update({f"{tag}_q": q.numpy(), f"{tag}_k": k.numpy(), f"{tag}_q_rot": qr.numpy(), f"{tag}_k_rot": kr.numpy(),
                    f"{tag}_cos": cos[:T].numpy(), f"{tag}_sin": sin[:T].numpy(),
                    f"{tag}_meta": np.asarray([B, H, T, dh, base, maxlen], np.float64)})
    print("[rope] restated tables + rotation == reference RotaryPositionEmbedding (bit for bit)")

    # model level: 2 layers, 2 heads of 16, rope_base 1000; the rotation inside the restatement done by the reference module
    rc = refvit.RefConfig(image_size=640, patch_size=32, hidden_size=32, num_hidden_layers=2, num_attention_heads=2,
                          stride_size=32, pos_encoding_type="rope", rope_base=1000.0, loss_name="mae")
    sd = refvit.make_state_dict(rc, 51)
    flux, _, labels = refvit.make_inputs(rc, 3, 52)
    mod = RotaryPositionEmbedding(dim=rc.head_dim, max_seq_len=rc.max_position_embeddings, base=rc.rope_base)
    own = refvit.apply_rope
    tr = refvit.RefTrainer(rc, sd, training=False)
    o = refvit.forward(rc, tr.params, flux, labels, output_hidden_states=True, output_attentions=True)
    o.loss.backward()
    try:
        refvit.apply_rope = lambda x, cos, sin: mod(x)
        with torch.no_grad():
            o_ref = refvit.forward(rc, sd, flux, labels, output_hidden_states=True, output_attentions=True)
    finally:
        refvit.apply_rope = own
    assert torch.equal(o_ref.logits, o.logits.detach()) and torch.equal(o_ref.attentions[0], o.attentions[0].detach())
    out.update(dict(p1_flux=flux.numpy(), p1_labels=labels.numpy(), p1_logits=o.logits.detach().numpy(),
                    p1_loss=np.float32(o.loss.detach()), p1_last=o.last_hidden_state.detach().numpy(),
                    p1_attn0=o.attentions[0].detach().numpy(), p1_wseed=np.int64(51)))
    for k, p in tr.params.items():
        if p.grad is not None:
            out[f"p1_grad/{k}"] = p.grad.detach().numpy()
    path = os.path.join(ROOT, "tests", "golden", "rope.npz")
    np.savez_compressed(path, **out)
    print(f"[rope] wrote {path} ({os.path.getsize(path) / 1024:.0f} KiB)")


def make_prep():
    """Linear input preprocessors: the reference's compute_zca_matrix / compute_pca_matrix /
    LinearPreprocessor (src/models/preprocessor.py) on a synthetic 48-dim covariance; tests compare the MI355X host
    module's matrices and its kernel forward against these outputs."""
    _import_reference()
    from src.models.preprocessor import LinearPreprocessor, compute_pca_matrix, compute_zca_matrix

    rng = np.random.Generator(np.random.PCG64(88))
    Dm = 48
    a = rng.standard_normal((Dm, 3 * Dm))
    cov = torch.from_numpy((a @ a.T / (3 * Dm)).astype(np.float64))
    cov = cov * torch.logspace(0, -3, Dm, dtype=torch.float64)[None, :] * torch.logspace(0, -3, Dm, dtype=torch.float64)[:, None]
    lam, vec = torch.linalg.eigh(cov)
    lam, vec = lam.flip(0).float(), vec.flip(1).float()          # descending, as the reference's cov files store them
    mean = torch.from_numpy(rng.standard_normal(Dm).astype(np.float32))
    x = torch.from_numpy(rng.standard_normal((8, Dm)).astype(np.float32))
    out = dict(eigvecs=vec.numpy(), eigvals=lam.numpy(), mean=mean.numpy(), x=x.numpy())
    cases = {"zca_full_s1": dict(r=None, shrinkage=0.1), "zca_full_s0": dict(r=None, shrinkage=0.0),
             "zca_r16_s2": dict(r=16, shrinkage=0.2), "zca_r40_s0": dict(r=40, shrinkage=0.0)}
    for k, kw in cases.items():
        out[k] = compute_zca_matrix(vec, lam, eps=1e-5, **kw).numpy()
    out["pca_r16"] = compute_pca_matrix(vec, r=16).numpy()
    out["pca_full"] = compute_pca_matrix(vec, r=None).numpy()
    P = torch.from_numpy(out["zca_r16_s2"])
    bias = -mean @ P.t()                                          # builder.py:66-68
    with torch.no_grad():
        out["y_zca_r16_s2"] = LinearPreprocessor(P, bias=bias, freeze=True)(x).numpy()
        Pp = torch.from_numpy(out["pca_r16"])
        out["y_pca_r16"] = LinearPreprocessor(Pp, bias=-mean @ Pp.t(), freeze=True)(x).numpy()
    # warmup.preprocessor: attention -- on 2-D spectra the module is its query projection (attention.py:81-84)
    from src.models.attention import PrefilledAttention
    for key, kw in {"attn_r16": dict(r=16, scale_by_eigvals=True), "attn_r24_noscale": dict(r=24, scale_by_eigvals=False)}.items():
        torch.manual_seed(3)
        m = PrefilledAttention(input_dim=Dm, eigvecs=vec, eigvals=lam, eps=1e-5, **kw)
        out[key + "_wq"] = m.q_lin.weight.detach().numpy()
        with torch.no_grad():
            out[key + "_y"] = m(x).numpy()
    path = os.path.join(ROOT, "tests", "golden", "prep.npz")
    np.savez_compressed(path, **out)
    print(f"[prep] wrote {path} ({os.path.getsize(path) / 1024:.0f} KiB)")

def make_deep(tag, rc, batch, wseed, xseed, steps=2):
    """C3 / C5 ("too big to commit: pin them by per-tensor checksums/norms + 16 sampled rows").  The reference
    composition (get_vit_config + HF ViTModel + the reference's SpectraEmbeddings) at the BENCHMARKED depth/width pins the
    restatement there too; the fixture keeps norms of every hidden state, 16 sampled token rows of three hidden states and
    of the final LayerNorm output, 16 sampled attention rows of the first and last layer, logits / loss in fp32 and under
    the reference's bf16 autocast, per-parameter gradient norms + 64 sampled entries, and a short clipped-AdamW loss
    trajectory.  Inputs / weights are regenerated from the seeds (checksums stored)."""
    get_vit_config, SpectraEmbeddings, _ = _import_reference()
    cfgd = cfg_dict_for(rc, "log_g")
    sd = refvit.make_state_dict(rc, wseed)
    flux, _, labels = refvit.make_inputs(rc, batch, xseed)
    vc, vit, lin = build_reference(cfgd, sd, get_vit_config, SpectraEmbeddings)
    L, T, D = rc.num_hidden_layers, rc.seq_len, rc.hidden_size
    with torch.no_grad():
        o, logits, loss = ref_forward(vc, vit, lin, rc, flux, labels)
        mine = refvit.forward(rc, sd, flux, labels, output_hidden_states=True, output_attentions=True)
    errs = {"last": rel(mine.last_hidden_state, o.last_hidden_state), "logits": rel(mine.logits, logits)}
    for i, (a, b) in enumerate(zip(mine.hidden_states, o.hidden_states)):
        errs[f"hs{i}"] = rel(a, b)
    for i, (a, b) in enumerate(zip(mine.attentions, o.attentions)):
        errs[f"att{i}"] = rel(a, b)
    print(f"[{tag}] oracle-vs-reference eval fwd at depth {L} x {D}: worst rel err {max(errs.values()):.3e}")
    assert max(errs.values()) < 5e-5, errs
    with torch.no_grad(), torch.autocast("cpu", dtype=torch.bfloat16):
        ob, logits_b, loss_b = ref_forward(vc, vit, lin, rc, flux, labels)
    print(f"[{tag}] reference bf16-autocast vs fp32: logits rel {rel(logits_b.float(), logits):.3e} "
          f"last rel {rel(ob.last_hidden_state.float(), o.last_hidden_state):.3e}")

    vit.eval()
    _, _, loss2 = ref_forward(vc, vit, lin, rc, flux, labels)
    loss2.backward()
    inv = {name_456_to_515(k): k for k in sd if k.startswith("vit.")}
    ref_grads = {inv[k]: (None if p.grad is None else p.grad.detach().clone()) for k, p in vit.named_parameters()}
    ref_grads[rc.head_name + ".weight"] = lin.weight.grad.detach().clone()
    ref_grads[rc.head_name + ".bias"] = lin.bias.grad.detach().clone()
    tr = refvit.RefTrainer(rc, sd, training=False)
    refvit.forward(rc, tr.params, flux, labels).loss.backward()
    gworst = 0.0
    gmax = max(float(x.norm()) for x in ref_grads.values() if x is not None)
    for k, p in tr.params.items():
        g = ref_grads[k]
        if g is None or float(g.norm()) < 1e-6 * (1 + gmax):
            continue
        gworst = max(gworst, rel(p.grad, g))
    print(f"[{tag}] oracle-vs-reference grads: worst rel err {gworst:.3e}")
    assert gworst < 2e-3, gworst
    # the reference's OWN bf16-autocast gradients (precision='bf16-mixed': autocast forward, backward through it): how far
    # they sit from its fp32 gradients is the yardstick for the bf16 gradient gate at this depth
    for p in list(vit.parameters()) + list(lin.parameters()):
        p.grad = None
    with torch.autocast("cpu", dtype=torch.bfloat16):
        _, _, loss_bg = ref_forward(vc, vit, lin, rc, flux, labels)
    loss_bg.float().backward()
    bf16_grads = {inv[k]: (None if p.grad is None else p.grad.detach().float()) for k, p in vit.named_parameters()}
    bf16_grads[rc.head_name + ".weight"] = lin.weight.grad.detach().float()
    bf16_grads[rc.head_name + ".bias"] = lin.bias.grad.detach().float()
    bf16_err = {}
    for k, g in ref_grads.items():
        if g is None or float(g.norm()) < 1e-6 * (1 + gmax):
            bf16_err[k] = 0.0
        else:
            bf16_err[k] = rel(bf16_grads[k], g)
    kw = max(bf16_err, key=bf16_err.get)
    print(f"[{tag}] reference bf16-autocast gradients vs its fp32 gradients: worst rel {bf16_err[kw]:.3e} ({kw}), "
          f"median {float(np.median([v for v in bf16_err.values() if v > 0])):.3e}")
    del vit, lin
    tr = refvit.RefTrainer(rc, sd, training=False)
    losses, gnorms = [], []
    for _ in range(steps):
        losses.append(tr.step(flux, labels))
        gnorms.append(tr.last_grad_norm)

    rows = sample_idx(batch * T, 16, 7)
    hs_layers = np.asarray([0, L // 2, L], np.int64)
    arow = sample_idx(batch * rc.num_attention_heads * T, 16, 8)
    out = dict(
        wseed=np.int64(wseed), xseed=np.int64(xseed), batch=np.int64(batch),
        weight_checksum=np.float64(sum(float(v.double().sum()) for v in sd.values())),
        flux_checksum=np.float64(flux.double().sum()), labels=labels.numpy(),
        hs_norms=np.asarray([float(h.double().norm()) for h in o.hidden_states]), hs_layers=hs_layers, rows=rows,
        hs_rows=np.stack([o.hidden_states[int(i)].reshape(-1, D)[rows].numpy() for i in hs_layers]),
        last_rows=o.last_hidden_state.reshape(-1, D)[rows].numpy(), last_norm=np.float64(o.last_hidden_state.double().norm()),
        attn_rows_idx=arow, attn0_rows=o.attentions[0].reshape(-1, T)[arow].numpy(),
        attn_last_rows=o.attentions[-1].reshape(-1, T)[arow].numpy(),
        logits=logits.numpy(), loss=np.float32(loss), bf16_logits=logits_b.float().numpy(), bf16_loss=np.float32(loss_b.float()),
        bf16_last_rows=ob.last_hidden_state.float().reshape(-1, D)[rows].numpy(),
        step_losses=np.asarray(losses, np.float64), step_grad_norms=np.asarray(gnorms, np.float64),
    )
    names = list(sd.keys())
    out["param_names"] = np.asarray(names)
    out["grad_norms"] = np.asarray([0.0 if ref_grads[k] is None else float(ref_grads[k].double().norm()) for k in names])
    out["has_grad"] = np.asarray([ref_grads[k] is not None for k in names])
    out["bf16_grad_err"] = np.asarray([bf16_err.get(k, 0.0) for k in names])
    gi, gv = [], []
    for i, k in enumerate(names):
        g = ref_grads[k]
        if g is None:
            gi.append(np.zeros(64, np.int64)); gv.append(np.zeros(64, np.float32))
            continue
        idx = sample_idx(g.numel(), 64, 1000 + i)
        idx = np.pad(idx, (0, 64 - len(idx)), mode="edge")
        gi.append(idx); gv.append(g.flatten()[idx].numpy())
    out["grad_idx"], out["grad_samples"] = np.stack(gi), np.stack(gv)
    path = os.path.join(ROOT, "tests", "golden", f"{tag}.npz")
    np.savez_compressed(path, **out)
    print(f"[{tag}] wrote {path} ({os.path.getsize(path) / 1024:.0f} KiB)")


def make_conv():
    """Conv1DPatchTokenizer (src/models/tokenization.py:53-69): `num_patches = (L - P) // S + 1` (no
    padded tail, unlike the sliding-window tokenizer), weight [D, 1, P].  The reference's own module, inside the
    reference's SpectraEmbeddings, inside the same composition as the other fixtures (proj_fn 'C1D' / 'CNN',
    embedding.py:33-44); one case with stride == patch, one with stride < patch and a remainder that is dropped."""
    get_vit_config, SpectraEmbeddings, _ = _import_reference()
    from src.models.tokenization import Conv1DPatchTokenizer

    out = {}
    cases = {
        "a": refvit.RefConfig(image_size=1024, patch_size=32, hidden_size=64, num_hidden_layers=2, num_attention_heads=2,
                              stride_size=32, proj_fn="C1D", loss_name="mae"),
        "b": refvit.RefConfig(image_size=1000, patch_size=64, hidden_size=64, num_hidden_layers=2, num_attention_heads=4,
                              stride_size=40, proj_fn="CNN", num_labels=2, loss_name="l1"),
    }
    for tag, rc in cases.items():
        param = "log_g" if rc.num_labels == 1 else "Teff,log_g"
        cfgd = cfg_dict_for(rc, param)
        sd = refvit.make_state_dict(rc, 61)
        flux, _, labels = refvit.make_inputs(rc, 3, 62)
        vc, vit, lin = build_reference(cfgd, sd, get_vit_config, SpectraEmbeddings)
        tk = vit.embeddings.patch_embeddings
        assert isinstance(tk, Conv1DPatchTokenizer), type(tk)
        assert tk.num_patches == rc.num_patches == (rc.image_size - rc.patch_size) // rc.stride + 1
        assert tuple(tk.projection.weight.shape) == (rc.hidden_size, 1, rc.patch_size)
        with torch.no_grad():
            o, logits, loss = ref_forward(vc, vit, lin, rc, flux, labels)
            tok = tk(flux)
            mine = refvit.forward(rc, sd, flux, labels, output_hidden_states=True)
        assert rel(mine.tokens, tok) < 2e-6 and rel(mine.logits, logits) < 2e-5, (rel(mine.tokens, tok), rel(mine.logits, logits))
        vit.eval()
        _, _, loss2 = ref_forward(vc, vit, lin, rc, flux, labels)
        loss2.backward()
        inv = {name_456_to_515(k): k for k in sd if k.startswith("vit.")}
        grads = {inv[k]: p.grad for k, p in vit.named_parameters() if p.grad is not None}
        grads[rc.head_name + ".weight"], grads[rc.head_name + ".bias"] = lin.weight.grad, lin.bias.grad
        tr = refvit.RefTrainer(rc, sd, training=False)
        refvit.forward(rc, tr.params, flux, labels).loss.backward()
        gw = max(rel(tr.params[k].grad, g) for k, g in grads.items() if float(g.norm()) > 1e-6)
        assert gw < 5e-4, gw
        print(f"[conv/{tag}] {rc.proj_fn}: N={rc.num_patches}; oracle-vs-reference tokens/logits/grads ok (grads {gw:.2e})")
        out.update({f"{tag}_flux": flux.numpy(), f"{tag}_labels": labels.numpy(), f"{tag}_tokens": tok.numpy(),
                    f"{tag}_hidden_states": np.stack([h.numpy() for h in o.hidden_states]),
                    f"{tag}_logits": logits.numpy(), f"{tag}_loss": np.float32(loss), f"{tag}_wseed": np.int64(61)})
        for k, g in grads.items():
            out[f"{tag}_grad/{k}"] = g.detach().numpy()
    path = os.path.join(ROOT, "tests", "golden", "conv.npz")
    np.savez_compressed(path, **out)
    print(f"[conv] wrote {path} ({os.path.getsize(path) / 1024:.0f} KiB)")

def make_data():
    """Dataset contract (src/dataloader/spec_datasets.py:37-110, src/dataloader/base.py:312-326): the
    reference's own `RegSpecDataset` built through its `from_config`, handed in-memory tensors in place of the HDF5 read
    (h5py is absent: `load_data`, which also clips the flux at zero and thresholds log_g for classification, cannot run, so
    those two lines stay unpinned), then its own `_maybe_normalize_labels`, `_set_noise` and `__getitem__`: label
    normalisation with the training split's statistics re-used on the validation split (what
    `ViTDataModule.setup_test_dataset` copies, src/vit.py:43-52), fixed-seed validation noise, 3- / 4-tuple items."""
    _import_reference()
    from src.dataloader.spec_datasets import RegSpecDataset

    rng = np.random.Generator(np.random.PCG64(99))
    out = {}
    n_tr, n_va, Lp = 12, 7, 40
    flux_tr = np.abs(rng.standard_normal((n_tr, Lp))).astype(np.float32)
    err_tr = (0.1 * np.abs(rng.standard_normal((n_tr, Lp)))).astype(np.float32)
    flux_va = np.abs(rng.standard_normal((n_va, Lp))).astype(np.float32)
    err_va = (0.1 * np.abs(rng.standard_normal((n_va, Lp)))).astype(np.float32)
    out.update(flux_tr=flux_tr, err_tr=err_tr, flux_va=flux_va, err_va=err_va)
    for tag, ncol in (("one", 1), ("three", 3)):
        p_tr = (3.0 + 2.0 * rng.standard_normal((n_tr, ncol))).astype(np.float32)
        p_va = (3.0 + 2.0 * rng.standard_normal((n_va, ncol))).astype(np.float32)
        if ncol == 1:
            p_tr, p_va = p_tr[:, 0], p_va[:, 0]
        out[f"{tag}_p_tr"], out[f"{tag}_p_va"] = p_tr, p_va
        for norm in ("minmax", "standard", "none"):
            cfg = {"data": {"file_path": "unused", "param": "log_g" if ncol == 1 else "T_eff,log_g,M_H", "label_norm": norm},
                   "noise": {"noise_level": 0.5}}
            tr = RegSpecDataset.from_config(cfg)
            tr.flux, tr.error = torch.from_numpy(flux_tr), torch.from_numpy(err_tr)
            tr.labels = torch.tensor(p_tr).float()          # spec_datasets.py:60
            tr._maybe_normalize_labels("fit")
            va = RegSpecDataset.from_config(cfg)
            va.flux, va.error = torch.from_numpy(flux_va), torch.from_numpy(err_va)
            va.labels = torch.tensor(p_va).float()
            for k in ("label_norm", "label_mean", "label_std", "label_min", "label_max"):  # vit.py:47-51
                setattr(va, k, getattr(tr, k))
            va._maybe_normalize_labels("val")
            va._set_noise()                                   # base.py:312-326 (seed 42)
            key = f"{tag}_{norm}"
            out[f"{key}_labels_tr"], out[f"{key}_labels_va"] = tr.labels.numpy(), va.labels.numpy()
            out[f"{key}_noisy_va"] = va.noisy.numpy()
            item_tr, item_va = tr[3], va[2]
            assert len(item_tr) == 3 and len(item_va) == 4
            out[f"{key}_item_va_noisy"], out[f"{key}_item_va_label"] = item_va[0].numpy(), np.asarray(item_va[3].numpy())
            out[f"{key}_item_tr_label"] = np.asarray(item_tr[2].numpy())
    path = os.path.join(ROOT, "tests", "golden", "data.npz")
    np.savez_compressed(path, **out)
    print(f"[data] wrote {path} ({os.path.getsize(path) / 1024:.0f} KiB)")

OPT_CASES = {
    "baseline_plateau": {"type": "AdamW", "lr": 0.001, "lr_sch": "plateau", "factor": 0.8, "patience": 10, "monitor_metric": "mae"},
    "plain_adam": {"lr": 3e-4},
    "sgd_wd": {"type": "sgd", "lr": 0.01, "weight_decay": 0.1},
    "cosine": {"type": "adamw", "lr": 1e-3, "lr_sch": "cosine", "ep": 40, "eta_min": 1e-5},
    "cosine_warmup_epochs": {"lr": 1e-3, "lr_sch": "cosine", "ep": 40, "warmup": {"epochs": 2}},
    "cosine_warmup_ratio": {"lr": 2e-3, "lr_sch": "cosineannealing", "T_max": 30, "warmup_ratio": 0.1},
    "constant": {"lr": 1e-3, "lr_sch": "constant", "factor": 0.5, "total_iters": 3},
    "onecycle": {"lr": 1e-3, "lr_sch": "onecycle", "steps_per_epoch": 7, "epochs": 3, "pct_start": 0.25},
    "plateau_max": {"type": "adamw", "lr": 1e-3, "lr_sch": "plateau", "mode": "max", "monitor_metric": "acc"},
}


def make_opt():
    """Optimizer / scheduler factory: the reference's own `OptModule` (src/opt/optimizer.py:1-173, imports
    torch only) on a small nn.Linear for a table of `opt:` sections -> optimizer class and defaults, scheduler classes,
    Lightning scheduler-config keys, and the learning-rate trace of 12 scheduler steps.  tests/golden/opt.json."""
    import json

    sys.path.insert(0, REF)
    from src.opt.optimizer import OptModule as RefOpt

    def describe(conf):
        opt = conf["optimizer"] if isinstance(conf, dict) else conf
        d = {"optimizer": type(opt).__name__, "lr": opt.defaults["lr"], "weight_decay": opt.defaults.get("weight_decay", 0)}
        if isinstance(conf, dict):
            sc = conf["lr_scheduler"]
            sch = sc["scheduler"]
            d["scheduler"] = type(sch).__name__
            d["inner"] = [type(x).__name__ for x in getattr(sch, "_schedulers", [])]
            d["keys"] = {k: v for k, v in sc.items() if k != "scheduler"}
            trace = []
            for i in range(12):
                opt.step()
                if sc.get("reduce_on_plateau"):
                    sch.step(1.0 if sch.mode == "min" else 0.0)   # never improves -> reductions after `patience`
                else:
                    sch.step()
                trace.append(opt.param_groups[0]["lr"])
            d["lr_trace"] = trace
        return d

    out = {}
    for name, cfg in OPT_CASES.items():
        lin = torch.nn.Linear(4, 4)
        out[name] = describe(RefOpt.from_config(dict(cfg))(lin))
    # the module-level wrapper (src/basemodule.py:152-182): plateau needs data.val_path, one-cycle gets its run length
    _import_reference()
    from src.basemodule import BaseLightningModule as RefBase

    class _Self:  # the attributes configure_optimizers reads
        def __init__(self, config):
            self.config, self.model, self.loss_name, self.monitor_metric = config, torch.nn.Linear(4, 4), "mae", "mae"

    mod_cases = {
        "plateau_no_val": {"opt": {"type": "AdamW", "lr": 1e-3, "lr_sch": "plateau"}, "data": {}},
        "plateau_with_val": {"opt": {"type": "AdamW", "lr": 1e-3, "lr_sch": "plateau", "factor": 0.5}, "data": {"val_path": "x.h5"}},
        "onecycle_run_length": {"opt": {"lr": 2e-3, "lr_sch": "onecycle"}, "data": {"num_samples": 1000},
                                "train": {"batch_size": 48, "ep": 3}},
        "no_scheduler": {"opt": {"lr": 1e-3}},
    }
    mod_out = {}
    for name, cfg in mod_cases.items():
        import copy
        d = describe(RefBase.configure_optimizers(_Self(copy.deepcopy(cfg))))
        if name == "onecycle_run_length":
            d["total_steps"] = RefBase.configure_optimizers(_Self(copy.deepcopy(cfg)))["lr_scheduler"]["scheduler"].total_steps
        mod_out[name] = d
    path = os.path.join(ROOT, "tests", "golden", "opt.json")
    with open(path, "w") as f:
        json.dump({"cases": OPT_CASES, "expected": out, "module_cases": mod_cases, "module_expected": mod_out}, f, indent=1)
    print(f"[opt] wrote {path}")

def make_evalstats():
    """Epoch-level regression statistics and the eval-step batch contract: the reference's own
    `ViTLModule.on_validation_epoch_end` and `_shared_eval_step` (src/vit.py:94-125, 157-187), called unbound on an object
    with the attributes they read (`torchmetrics`, absent here, is stubbed just far enough for `import src.vit`; the metric
    objects the methods call are simple recorders).  Records: val_bias_median / val_p90 / val_beta for one and for three
    targets, and which tensor of a 3- / 4-tuple batch reaches the model at noise_level 0 and > 0."""
    import json

    _import_reference()
    if "torchmetrics" not in sys.modules:
        tm = types.ModuleType("torchmetrics")
        for n in ("Accuracy", "MeanAbsoluteError", "MeanSquaredError", "R2Score"):
            setattr(tm, n, type(n, (object,), {}))
        sys.modules["torchmetrics"] = tm
    from src.vit import ViTLModule as RefModule, _normalize_task

    rng = np.random.Generator(np.random.PCG64(123))
    out = {"stats": {}, "tasks": {}}
    for tag, ncol in (("one", 1), ("three", 3)):
        lab = rng.random((50, ncol)).astype(np.float32)
        pred = (0.1 + 0.8 * lab + 0.05 * rng.standard_normal((50, ncol))).astype(np.float32)
        if ncol == 1:
            lab, pred = lab[:, 0], pred[:, 0]
        logged = {}

        class Host:
            task_type = "reg"
            val_dict = {"preds": [torch.from_numpy(pred[:20]), torch.from_numpy(pred[20:])],
                        "labels": [torch.from_numpy(lab[:20]), torch.from_numpy(lab[20:])]}

            def log(self, name, value, **kw):
                logged[name] = float(value)

        RefModule.on_validation_epoch_end(Host())
        out["stats"][tag] = {"pred": pred.tolist(), "label": lab.tolist(), "logged": logged}
    for cfg in ({"model": {"task_type": "reg"}}, {"model": {"task": "Classification"}}, {"model": {}}, {"model": {"task_type": "cls"}},
                {"model": {"task_type": "regression"}}):
        out["tasks"][json.dumps(cfg)] = _normalize_task(cfg)

    # which tensor reaches the model
    seen = {}
    for nl in (0.0, 0.3):
        for n_items in (3, 4):
            class Out:
                loss = torch.tensor(0.5)
                logits = torch.zeros(4, 1)

            class Host2:
                task_type, noise_level, loss_name = "reg", nl, "mae"

                def forward(self, x, labels, loss_only=True):
                    seen[f"nl{nl}_n{n_items}"] = float(x[0, 0])
                    return Out()

                def log(self, *a, **k):
                    pass

                mae = mse = r2 = staticmethod(lambda p, t: torch.tensor(0.0))

            noisy, flux, err, lab4 = torch.full((4, 8), 9.0), torch.full((4, 8), 1.0), torch.ones(4, 8), torch.zeros(4)
            batch = (noisy, flux, err, lab4) if n_items == 4 else (flux, err, lab4)
            RefModule._shared_eval_step(Host2(), batch, "val")
    out["eval_input"] = seen  # 9.0 = the pre-generated noisy copy, 1.0 = the clean flux
    path = os.path.join(ROOT, "tests", "golden", "evalstats.json")
    with open(path, "w") as f:
        json.dump(out, f)
    print(f"[evalstats] wrote {path}: {out['eval_input']} {out['tasks']}")

NAME_CASES = [
    dict(patch_size=32, hidden_size=32, num_hidden_layers=3, num_attention_heads=2, stride_size=32, stride_ratio=1, proj_fn="SW", noise=0),
    dict(patch_size=256, hidden_size=768, num_hidden_layers=12, num_attention_heads=12, stride_size=None, stride_ratio=0.5, proj_fn="C1D", noise=0.5),
    dict(patch_size=64, hidden_size=64, num_hidden_layers=2, num_attention_heads=4, stride_size=0, stride_ratio=1, proj_fn="CNN", noise=1.25),
]
WARM_CASES = [
    ("zca", dict(r=16, shrinkage=0.2, freeze_epochs=5)), ("zca", dict(freeze_epochs=-1, bias=False)), ("zca", dict(r=None, shrinkage=0.0)),
    ("pca", dict(r=8)), ("pca", dict(r=None, bias=False, freeze_epochs=3)),
    ("attention", dict(r=8, freeze_epochs=2)), ("attention", dict(scale_by_eigvals=False)),
]


def make_names():
    """Run names (checkpoints and logs are named after them): the reference's own `build_model_name`
    (src/models/model_utils.py:9-45) and the name prefix / output width its `_build_preprocessor` gives each `warmup:` variant
    (src/models/builder.py:45-133) on synthetic covariance statistics.  tests/golden/names.json."""
    import json

    _import_reference()
    from src.models.builder import _build_preprocessor as ref_build
    from src.models.model_utils import build_model_name as ref_name

    out = {"names": [], "warm": []}
    for c in NAME_CASES:
        ns = types.SimpleNamespace(**{k: v for k, v in c.items() if k != "noise"})
        out["names"].append({"case": c, "ViT": ref_name(ns, "ViT", full_config={"noise": {"noise_level": c["noise"]}}),
                             "plain": ref_name(ns, "ZCA_ViT")})
    g = torch.Generator().manual_seed(3)
    q, _ = torch.linalg.qr(torch.randn(64, 64, generator=g))
    stats = {"eigvecs": q, "eigvals": torch.logspace(0, -2, 64), "mean": torch.randn(64, generator=g)}
    for kind, warm in WARM_CASES:
        pre, out_dim, prefix, _desc = ref_build(kind, dict(warm), stats, 64, warm.get("freeze_epochs", 0) != 0)
        out["warm"].append({"kind": kind, "warmup": warm, "prefix": prefix, "out_dim": int(out_dim)})
    path = os.path.join(ROOT, "tests", "golden", "names.json")
    with open(path, "w") as f:
        json.dump(out, f, indent=1)
    print(f"[names] wrote {path}: " + ", ".join(w["prefix"] for w in out["warm"]))

CONFIG_CASES = {
    "reg_one": {"model": dict(task_type="reg", image_size=4096, patch_size=32, hidden_size=32, num_hidden_layers=3, num_attention_heads=2, stride_size=32, proj_fn="SW"), "data": {"param": "log_g"}},
    "reg_three_str": {"model": dict(task_type="reg", image_size=1000, patch_size=64, hidden_size=64, num_hidden_layers=2, num_attention_heads=4, stride_ratio=0.5, proj_fn="C1D", num_labels=7), "data": {"param": "T_eff, log_g ,M_H"}},
    "reg_list": {"model": dict(task_type="regression", image_size=512, patch_size=32, hidden_size=64, num_hidden_layers=1, num_attention_heads=2, proj_fn="SW"), "data": {"param": ["a", "b"]}},
    "reg_no_param": {"model": dict(task_type="reg", image_size=512, patch_size=32, hidden_size=64, num_hidden_layers=1, num_attention_heads=2, proj_fn="SW", num_labels=4), "data": {}},
    "cls": {"model": dict(task_type="cls", image_size=512, patch_size=32, hidden_size=64, num_hidden_layers=2, num_attention_heads=2, stride_size=16, proj_fn="SW", num_labels=5, pos_encoding_type="learned", max_position_embeddings=64, rope_base=500.0)},
}


def make_config():
    """`get_vit_config`: the reference's own function (src/models/builder.py:200-258) on a table of configs;
    the fields the path reads, plus the `num_labels` it writes back into config['model'].  tests/golden/config.json."""
    import copy
    import json

    get_vit_config, _, _ = _import_reference()
    fields = ["task_type", "image_size", "patch_size", "hidden_size", "num_hidden_layers", "num_attention_heads", "proj_fn",
              "stride_ratio", "stride_size", "num_labels", "num_channels", "hidden_act", "hidden_dropout_prob",
              "attention_probs_dropout_prob", "layer_norm_eps", "qkv_bias", "intermediate_size", "pos_encoding_type",
              "max_position_embeddings", "rope_base", "initializer_range"]
    out = {}
    for name, cfg in CONFIG_CASES.items():
        c = copy.deepcopy(cfg)
        vc = get_vit_config(c)
        out[name] = {"fields": {k: getattr(vc, k, None) for k in fields}, "written_back_num_labels": c["model"].get("num_labels")}
    path = os.path.join(ROOT, "tests", "golden", "config.json")
    with open(path, "w") as f:
        json.dump({"cases": CONFIG_CASES, "expected": out}, f, indent=1)
    print(f"[config] wrote {path}")

def make_freeze():
    """Freeze schedule of the input preprocessor: the reference's own `PreprocessorFreezeCallback`
    (src/prepca/callbacks.py) driven by recorder objects over 5 epochs for freeze_epochs in {0, 1, 3, -1}: which
    `set_preprocessor_trainable(...)` calls it makes and when.  tests/golden/freeze.json."""
    import json

    _import_reference()
    from src.prepca.callbacks import PreprocessorFreezeCallback

    out = {}
    for fe in (0, 1, 3, -1):
        calls = []

        class Model:
            def set_preprocessor_trainable(self, flag):
                calls.append([tr.current_epoch, bool(flag)])

        class Trainer:
            current_epoch = -1

        class Module:
            model = Model()

        tr, mod = Trainer(), Module()
        cb = PreprocessorFreezeCallback(freeze_epochs=fe)
        cb.on_train_start(tr, mod)
        for ep in range(5):
            tr.current_epoch = ep
            cb.on_train_epoch_start(tr, mod)
        out[str(fe)] = calls
    path = os.path.join(ROOT, "tests", "golden", "freeze.json")
    with open(path, "w") as f:
        json.dump(out, f)
    print(f"[freeze] wrote {path}: {out}")

LOADCFG_YAML = """project: demo
data:
  file_path: ${VIT_GOLD_ROOT}/train.h5
  val_path: $VIT_GOLD_ROOT/val.h5
  extra: ["~/cov.pt", 3, {nested: "${VIT_GOLD_ROOT}/x"}]
  num_samples: 100
model:
  hidden_size: 32
  name: "plain string"
train:
  precision: 'bf16-mixed'
  grad_clip: 0.5
"""


def make_loadcfg():
    """`load_config` (src/utils.py:311-359) on a plain experiment YAML: the reference's own function with VIT_GOLD_ROOT and HOME
    fixed; tests/golden/loadcfg.json holds the YAML text, the environment and the dict it returned (the W&B-export
    unwrapping of that function is out of scope here and not exercised)."""
    import json
    import tempfile

    _import_reference()
    from src.utils import load_config as ref_load

    env = {"VIT_GOLD_ROOT": "/data/gold", "HOME": "/home/u"}
    old = {k: os.environ.get(k) for k in env}
    os.environ.update(env)
    try:
        with tempfile.NamedTemporaryFile("w", suffix=".yaml", delete=False) as f:
            f.write(LOADCFG_YAML)
        got = ref_load(f.name)
    finally:
        for k, v in old.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v
        os.unlink(f.name)
    path = os.path.join(ROOT, "tests", "golden", "loadcfg.json")
    with open(path, "w") as fo:
        json.dump({"yaml": LOADCFG_YAML, "env": env, "expected": got}, fo, indent=1)
    print(f"[loadcfg] wrote {path}: {got['data']}")


WANDB_NESTED_YAML = """_wandb:
  value:
    cli_version: 0.21.0
config:
  value:
    model: {name: vit, image_size: 4096, patch_size: 32, hidden_size: 32}
    train: {ep: 3, save: false}
    data:
      file_path: ${VIT_GOLD_ROOT}/train.h5
      param: log_g
"""
WANDB_PERKEY_YAML = """_wandb:
  value: {cli_version: 0.21.0, python_version: 3.10.12}
model:
  desc: null
  value: {name: vit, image_size: 4096, patch_size: 32, hidden_size: 32, num_attention_heads: 2}
train:
  value: {ep: 5, batch_size: 64, precision: bf16-mixed}
loss:
  value: {name: mae}
data:
  value:
    file_path: ~/spec/train.h5
    val_path: ${VIT_GOLD_ROOT}/val.h5
    param: "T_eff,log_g"
lr:
  value: 0.001
tags:
  value: [a, b]
project: plain-key
"""


def make_wandbcfg():
    """`load_config` (src/utils.py:311-359) on the two W&B-export shapes it unwraps -- the real config nested under
    `config.value`, and every top-level key wrapped as `{value: ...}` with the `_wandb` section dropped -- the reference's own
    function, environment fixed; tests/golden/wandbcfg.json holds the YAML texts and the dicts it returned."""
    import json
    import tempfile

    _import_reference()
    from src.utils import load_config as ref_load

    env = {"VIT_GOLD_ROOT": "/data/gold", "HOME": "/home/u"}
    old = {k: os.environ.get(k) for k in env}
    os.environ.update(env)
    out = {"env": env, "cases": []}
    try:
        for text in (WANDB_NESTED_YAML, WANDB_PERKEY_YAML):
            with tempfile.NamedTemporaryFile("w", suffix=".yaml", delete=False) as f:
                f.write(text)
            try:
                out["cases"].append({"yaml": text, "expected": ref_load(f.name)})
            finally:
                os.unlink(f.name)
    finally:
        for k, v in old.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v
    path = os.path.join(ROOT, "tests", "golden", "wandbcfg.json")
    with open(path, "w") as fo:
        json.dump(out, fo, indent=1)
    print(f"[wandbcfg] wrote {path}: {[sorted(c['expected']) for c in out['cases']]}")


def main():
    torch.manual_seed(0)
    torch.set_num_threads(8)

    # C1: configs/exp/att_clp/baseline.yaml; two of the four rows come from the reference's make_dummy_spectra
    def c1_inputs(make_dummy_spectra, flux):
        d = make_dummy_spectra(n=2, length=4096, seed=0)  # src/utils.py:131-139
        return torch.cat([flux[:2], d], dim=0)

    make_one("c1", refvit.named_config("C1"), "log_g", 4, 11, 12, True, c1_inputs)
    make_one("c2", refvit.named_config("C2"), "log_g", 2, 21, 22, False)
    # ragged: overlapping stride with a zero-padded tail patch (tokenization.py:46-48), 3 regression targets, L1 loss
    r1 = refvit.RefConfig(image_size=1000, patch_size=64, hidden_size=64, num_hidden_layers=2, num_attention_heads=4,
                          stride_size=48, num_labels=3, loss_name="l1")
    make_one("r1", r1, "Teff,log_g,M_H", 3, 31, 32, True)
    # classification head + learned position embeddings (embedding.py:61-66, 95-97; specvit.py:45-48)
    k1 = refvit.RefConfig(image_size=512, patch_size=32, hidden_size=64, num_hidden_layers=2, num_attention_heads=2,
                          stride_size=32, task_type="cls", num_labels=5, pos_encoding_type="learned", loss_name="ce")
    make_one("k1", k1, "log_g", 6, 41, 42, True)
    make_rope()
    make_prep()
    make_conv()
    make_data()
    make_opt()
    make_evalstats()
    make_names()
    make_config()
    make_freeze()
    make_loadcfg()
    make_wandbcfg()
    # the benchmarked geometries
    make_deep("c3", refvit.named_config("C3"), 4, 71, 72)
    make_deep("c5", refvit.named_config("C5"), 2, 81, 82)
    # four corners of the reference's sweep space (configs/sweep.yaml:10-21), same recipe
    for tag, (b, ws) in SWEEP_CASES.items():
        make_deep(tag, refvit.named_config(tag.upper()), b, ws, ws + 1)


SWEEP_CASES = {"s1": (2, 91), "s2": (4, 93), "s3": (8, 95), "s4": (16, 97)}


if __name__ == "__main__":
    if len(sys.argv) > 1:  # regenerate selected fixtures only: python oracle/make_golden.py conv c3 c5
        torch.manual_seed(0)
        torch.set_num_threads(8)
        for what in sys.argv[1:]:
            {"rope": make_rope, "prep": make_prep, "conv": make_conv, "data": make_data, "opt": make_opt, "evalstats": make_evalstats, "names": make_names, "config": make_config, "freeze": make_freeze, "loadcfg": make_loadcfg, "wandbcfg": make_wandbcfg,
             "c3": lambda: make_deep("c3", refvit.named_config("C3"), 4, 71, 72),
             "c5": lambda: make_deep("c5", refvit.named_config("C5"), 2, 81, 82),
             **{t: (lambda t=t: make_deep(t, refvit.named_config(t.upper()), SWEEP_CASES[t][0], SWEEP_CASES[t][1],
                                          SWEEP_CASES[t][1] + 1)) for t in SWEEP_CASES}}[what]()
    else:
        main()
