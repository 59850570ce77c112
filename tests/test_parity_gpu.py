"""Parity of the HIP path (through MyViT -> engine -> C ABI) with the golden vectors generated from the reference's
own modules (tests/golden/*.npz, oracle/make_golden.py) and with the CPU oracle on the same seeded inputs.

Tolerances (stated per north_star):
  * precision='32' (the reference's default; split-bf16 x3 GEMMs + fp32 attention / LayerNorm): the north_star gate
    "forward within 1e-3 rel of reference" is asserted at 1e-4 on every hidden state, attention map and the logits
    (measured <= 1.2e-5), gradients at 2e-4 per tensor (measured <= 1.7e-5), 3-step loss trajectory at 5e-4.
  * precision='bf16-mixed': the MFMA contractions consume bf16 operands (fp32 accumulate, fp32 residual stream / LN / softmax statistics), i.e.
    the arithmetic of the reference under precision='bf16-mixed'.  The fixtures hold the reference's fp32 outputs AND
    its own bf16-autocast outputs; the reference's bf16 run differs from its fp32 run by 4e-3..7e-3 (relative L2), so a
    1e-3 match to the fp32 outputs is not reachable by ANY bf16 pipeline.  The gate is therefore:
        err(ours, ref_fp32) <= 1.5 * err(ref_bf16_autocast, ref_fp32) + 1e-3        (no worse than the reference's own
    bf16 mode), and in absolute terms relative-L2 <= 1.5e-2 on every hidden state.
  * gradients: relative L2 <= 4e-2 and cosine >= 0.999 per tensor against the reference's fp32 autograd.
  * fused AdamW vs torch.optim.AdamW on the same gradients: <= 2e-6 relative after one step (kernel test: 1e-6 over 3).
"""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

GOLD = os.path.join(os.path.dirname(__file__), "golden")


def rel(a, b):
    a, b = a.double().cpu(), b.double().cpu()
    return float((a - b).norm() / (b.norm() + 1e-30))


def setup(tag, dev, precision="bf16-mixed"):
    from oracle import refvit
    from vit_amd.config import ViTConfig
    from vit_amd.specvit import MyViT

    rcs = {
        "c1": refvit.named_config("C1"),
        "c2": refvit.named_config("C2"),
        "r1": refvit.RefConfig(image_size=1000, patch_size=64, hidden_size=64, num_hidden_layers=2,
                               num_attention_heads=4, stride_size=48, num_labels=3, loss_name="l1"),
        "k1": refvit.RefConfig(image_size=512, patch_size=32, hidden_size=64, num_hidden_layers=2,
                               num_attention_heads=2, stride_size=32, task_type="cls", num_labels=5,
                               pos_encoding_type="learned", loss_name="ce"),
    }
    rc = rcs[tag]
    g = np.load(os.path.join(GOLD, f"{tag}.npz"))
    sd = refvit.make_state_dict(rc, int(g["wseed"]))
    assert abs(sum(float(v.double().sum()) for v in sd.values()) - float(g["weight_checksum"])) < 1e-6
    cfg = ViTConfig(task_type=rc.task_type, image_size=rc.image_size, patch_size=rc.patch_size,
                    hidden_size=rc.hidden_size, num_hidden_layers=rc.num_hidden_layers,
                    num_attention_heads=rc.num_attention_heads, proj_fn=rc.proj_fn, stride_size=rc.stride_size,
                    num_labels=rc.num_labels, pos_encoding_type=rc.pos_encoding_type)
    model = MyViT(cfg, loss_name=rc.loss_name)
    model.set_precision(precision)
    missing = model.load_state_dict(sd, strict=True)
    model = model.to(dev)
    x = torch.from_numpy(g["flux"]).to(dev)
    labels = torch.from_numpy(g["labels"]).to(dev)
    return rc, g, sd, model, x, labels


@pytest.mark.parametrize("tag", ["c1", "c2", "r1", "k1"])
def test_eval_forward_matches_reference(dev, tag):
    rc, g, sd, model, x, labels = setup(tag, dev)
    model.eval()
    out = model(x, labels=labels, output_hidden_states=True, output_attentions=True)
    T = lambda k: torch.from_numpy(g[k])
    ref_err = rel(T("bf16_last_hidden_state"), T("last_hidden_state"))
    hs = torch.stack([h.cpu() for h in out.hidden_states])
    assert hs.shape == T("hidden_states").shape
    # tokens (patch embedding) and every hidden state
    N = rc.num_patches
    emb = out.hidden_states[0].cpu()
    if rc.pos_encoding_type == "learned":
        emb = emb - sd["vit.embeddings.position_embeddings"]
    assert rel(emb[:, 1:], T("tokens")) < 6e-3
    assert rel(emb[:, 0], sd["vit.embeddings.cls_token"].view(1, -1).expand(emb.shape[0], -1)) < 1e-6
    errs = [rel(hs[i], T("hidden_states")[i]) for i in range(hs.shape[0])]
    assert max(errs) < 1.5e-2, errs
    assert rel(out.attentions[0], T("attn0")) < 1.5e-2
    assert rel(out.attentions[-1], T("attn_last")) < 2e-2
    # logits / loss: no worse than the reference's own bf16 mode
    e_logits = rel(out.logits, T("logits"))
    e_ref = rel(T("bf16_logits"), T("logits"))
    assert e_logits <= 1.5 * max(e_ref, ref_err) + 1e-3, (e_logits, e_ref)
    assert abs(float(out.loss) - float(g["loss"])) <= 3e-2 * abs(float(g["loss"])) + 1e-4
    print(f"[{tag}] hidden-state rel errs max {max(errs):.2e}; logits {e_logits:.2e} (reference bf16 mode: {e_ref:.2e})")


@pytest.mark.parametrize("tag", ["c1", "c2", "r1", "k1"])
def test_gradients_match_reference(dev, tag):
    rc, g, sd, model, x, labels = setup(tag, dev)
    model.eval()  # dropout off: masks are implementation-defined, so gradient parity is asserted at p = 0
    loss = model(x, labels=labels).loss
    loss.backward()
    names = [str(n) for n in g["param_names"]]
    gn = g["grad_norms"]
    worst = 0.0
    for name, p in model.named_parameters():
        i = names.index(name)
        if f"grad/{name}" not in g.files:
            assert p.grad is None, name  # pooler
            continue
        ref = torch.from_numpy(g[f"grad/{name}"])
        mine = p.grad.detach().cpu().flatten()
        if f"gidx/{name}" in g.files:
            mine = mine[torch.from_numpy(g[f"gidx/{name}"])]
            ref = ref.flatten()
        else:
            ref = ref.flatten()
        if gn[i] < 1e-6:  # key.bias: analytically zero
            assert float(p.grad.norm()) < 1e-3 * (1 + float(max(gn))), name
            continue
        e = rel(mine, ref)
        cos = float(torch.dot(mine.double(), ref.double()) / (mine.double().norm() * ref.double().norm() + 1e-30))
        # per-tensor norms too (covers the un-sampled entries of the big tensors)
        nrm = float(p.grad.double().norm())
        assert abs(nrm - gn[i]) <= 4e-2 * gn[i] + 1e-7, (name, nrm, gn[i])
        assert e < 4e-2 and cos > 0.999, (name, e, cos)
        worst = max(worst, e)
    print(f"[{tag}] worst gradient rel err {worst:.2e}")


@pytest.mark.parametrize("tag", ["c1", "k1"])
def test_training_steps_track_reference(dev, tag):
    """3 steps of: fwd -> bwd -> clip_grad_norm_(0.5) -> AdamW(lr 1e-3, wd 0), dropout off; fused optimizer path."""
    from vit_amd.optimizer import OptModule

    rc, g, sd, model, x, labels = setup(tag, dev)
    model.eval()
    opt = OptModule.from_config({"type": "AdamW", "lr": 1e-3})(model)
    opt.set_grad_clip(0.5)
    losses, norms = [], []
    for s in range(3):
        opt.zero_grad()
        loss = model(x, labels=labels).loss
        loss.backward()
        opt.step()
        losses.append(float(loss))
        norms.append(float(opt.last_grad_norm.sqrt()))
    ref_l, ref_n = g["step_losses"], g["step_grad_norms"]
    # absolute floor: bf16 logits carry ~4e-3 relative noise (the reference's own autocast outputs differ from its fp32
    # ones by 3.9e-3 .. 6.3e-3), i.e. d(loss) ~ 2 * residual * 4e-3 * |logit| -- ~2e-3 of the starting loss once the loss
    # itself has fallen 25x (c1, step 3: 0.043)
    floor = 2e-3 * float(max(ref_l))
    for a, b in zip(losses, ref_l):
        assert abs(a - b) <= 3e-2 * abs(b) + floor, (losses, ref_l)
    # step 1 sees identical weights; later steps compound Adam's sign-like first updates (+-lr per element, so bf16
    # noise on near-zero gradients flips whole steps) -- in the reference's own bf16 mode too
    for s_, (a, b) in enumerate(zip(norms, ref_n)):
        assert abs(a - b) <= (4e-2 if s_ == 0 else 1e-1) * abs(b), (norms, ref_n)
    # parameter movement after 3 steps: compare the update direction on tensors with a healthy gradient
    moved = 0
    for name, p in model.named_parameters():
        k = f"after3/{name}"
        if k not in g.files or f"gidx/{name}" in g.files:
            continue
        ref_delta = torch.from_numpy(g[k]).flatten() - sd[name].flatten()
        my_delta = p.detach().cpu().flatten() - sd[name].flatten()
        if float(ref_delta.norm()) < 1e-6:
            continue
        cos = float(torch.dot(my_delta.double(), ref_delta.double()) / (my_delta.double().norm() * ref_delta.double().norm()))
        if "key.bias" in name:
            continue  # zero gradient: Adam turns rounding noise into +-lr steps, in the reference too
        assert cos > 0.9, (name, cos)  # Adam's first steps are sign-like: bf16 noise flips a few +-lr element steps
        moved += 1
    assert moved > 10


def test_fused_adamw_equals_torch_adamw(dev):
    """Same model, same gradients: FusedAdamW (+ fused clip) vs clip_grad_norm_ + torch.optim.AdamW on the .grad views."""
    from vit_amd.optimizer import FusedAdamW

    rc, g, sd, model, x, labels = setup("c1", dev)
    rc2, g2, sd2, model2, _, _ = setup("c1", dev)
    model.eval()
    model2.eval()
    fused = FusedAdamW(model, lr=1e-3, weight_decay=0.01)
    fused.set_grad_clip(0.5)
    ref = torch.optim.AdamW(model2.parameters(), lr=1e-3, weight_decay=0.01)
    for s in range(3):
        fused.zero_grad()
        ref.zero_grad(set_to_none=True)
        model(x, labels=labels).loss.backward()
        model2(x, labels=labels).loss.backward()
        torch.nn.utils.clip_grad_norm_([p for p in model2.parameters() if p.grad is not None], 0.5)
        fused.step()
        ref.step()
        for (n1, p1), (n2, p2) in zip(model.named_parameters(), model2.named_parameters()):
            if "key.bias" in n1:
                continue  # analytically zero gradient: Adam normalises pure rounding noise into +-lr steps
            # step 0 is exact to rounding; afterwards a 1-ulp difference in an updated f32 weight (torch divides by
            # sqrt(bc2), the kernel multiplies by its reciprocal) can flip that weight's bf16 rounding, and Adam's
            # normalisation amplifies the resulting gradient noise.  Exact equivalence on identical gradients is
            # asserted at kernel level (test_kernels_gpu.py::test_sqnorm_adamw).
            assert rel(p1.detach(), p2.detach()) < (2e-6 if s == 0 else 5e-3), (s, n1)


def test_state_dict_names_and_roundtrip(dev):
    rc, g, sd, model, x, labels = setup("k1", dev)
    got = model.state_dict()
    assert list(got.keys()) == list(sd.keys()) or sorted(got.keys()) == sorted(sd.keys())
    for k in sd:
        assert torch.equal(got[k].cpu(), sd[k]), k
    # hot reload of new weights is picked up by the bf16 shadow
    model.eval()
    l0 = float(model(x, labels=labels).loss)
    sd2 = {k: v * 0.5 for k, v in sd.items()}
    model.load_state_dict(sd2)
    l1 = float(model(x, labels=labels).loss)
    assert l0 != l1
    model.load_state_dict(sd)
    assert abs(float(model(x, labels=labels).loss) - l0) < 1e-6


def test_training_mode_dropout_statistics(dev):
    """Dropout on (train mode): loss differs between steps/seeds, is finite, and its mean over seeds stays close to the
    eval loss scale; backward runs and yields finite gradients."""
    rc, g, sd, model, x, labels = setup("c1", dev)
    model.train()
    vals = []
    for _ in range(8):
        for p in model.parameters():
            p.grad = None
        loss = model(x, labels=labels).loss
        loss.backward()
        vals.append(float(loss))
        gn = torch.sqrt(sum((p.grad.double() ** 2).sum() for p in model.parameters() if p.grad is not None))
        assert torch.isfinite(gn)
    assert len(set(round(v, 7) for v in vals)) > 1
    assert all(np.isfinite(vals))


def test_cpu_tensor_fails_loudly():
    from vit_amd._cabi import VitError
    from vit_amd.config import ViTConfig
    from vit_amd.specvit import MyViT

    cfg = ViTConfig(task_type="reg", image_size=256, patch_size=32, hidden_size=32, num_hidden_layers=1,
                    num_attention_heads=2, stride_size=32)
    m = MyViT(cfg, loss_name="mae")
    with pytest.raises(VitError):
        m(torch.zeros(2, 256), labels=torch.zeros(2))


# ------------------------------------------------------------------ precision='32' (the reference's default): fp32-class
@pytest.mark.parametrize("tag", ["c1", "c2", "r1", "k1"])
def test_f32_mode_forward_within_1e3_of_reference(dev, tag):
    """north_star's gate, literally: forward within 1e-3 (relative L2) of the reference's fp32 outputs -- here on every
    hidden state, attention map, the logits and the loss, with the split-bf16 x3 GEMMs / fp32 attention / fp32 LN."""
    rc, g, sd, model, x, labels = setup(tag, dev, precision="32")
    model.eval()
    out = model(x, labels=labels, output_hidden_states=True, output_attentions=True)
    T = lambda k: torch.from_numpy(g[k])
    hs = torch.stack([h.cpu() for h in out.hidden_states])
    errs = [rel(hs[i], T("hidden_states")[i]) for i in range(hs.shape[0])]
    e_att = max(rel(out.attentions[0], T("attn0")), rel(out.attentions[-1], T("attn_last")))
    e_log = rel(out.logits, T("logits"))
    print(f"[{tag}] f32 mode: hidden states max {max(errs):.2e}, attention {e_att:.2e}, logits {e_log:.2e}")
    # gate: 1e-3 (north_star); measured on MI355X: <= 1.2e-5 everywhere -- assert an order of magnitude inside the gate
    assert max(errs) < 1e-4 and e_att < 1e-4 and e_log < 1e-4
    assert abs(float(out.loss) - float(g["loss"])) <= 1e-4 * abs(float(g["loss"])) + 1e-7


@pytest.mark.parametrize("tag", ["c1", "r1", "k1"])
def test_f32_mode_gradients_and_steps(dev, tag):
    from vit_amd.optimizer import FusedAdamW

    rc, g, sd, model, x, labels = setup(tag, dev, precision="32")
    model.eval()
    model(x, labels=labels).loss.backward()
    names = [str(n) for n in g["param_names"]]
    gn = g["grad_norms"]
    worst = 0.0
    for name, p in model.named_parameters():
        if f"grad/{name}" not in g.files:
            assert p.grad is None
            continue
        if gn[names.index(name)] < 1e-6:
            continue
        e = rel(p.grad.detach().cpu().flatten(), torch.from_numpy(g[f"grad/{name}"]).flatten())
        worst = max(worst, e)
        assert e < 2e-4, (name, e)  # measured <= 1.7e-5
    print(f"[{tag}] f32 mode: worst gradient rel err {worst:.2e}")
    # three optimisation steps against the oracle's trajectory (dropout off)
    rc, g, sd, model, x, labels = setup(tag, dev, precision="32")
    model.eval()
    opt = FusedAdamW(model, lr=1e-3)
    opt.set_grad_clip(0.5)
    losses = []
    for _ in range(3):
        opt.zero_grad()
        loss = model(x, labels=labels).loss
        loss.backward()
        opt.step()
        losses.append(float(loss))
    assert np.allclose(losses, g["step_losses"], rtol=5e-4), (losses, g["step_losses"])


def test_f32_mode_train_step_runs_with_dropout(dev):
    rc, g, sd, model, x, labels = setup("c1", dev, precision="32")
    model.train()
    l1 = model(x, labels=labels).loss
    l1.backward()
    assert torch.isfinite(l1) and all(torch.isfinite(p.grad).all() for p in model.parameters() if p.grad is not None)


# ------------------------------------------------------------------ rotary position embedding
@pytest.mark.parametrize("dt", [torch.float32, torch.bfloat16])
def test_rope_kernel_matches_reference_vectors(dev, dt):
    """vit_rope_qk on the reference's own vectors (tests/golden/rope.npz: RotaryPositionEmbedding.forward_qk outputs);
    f32: <= 1 ulp-level (the kernel evaluates x*cos - x2*sin with FMAs), bf16: one rounding of the result."""
    import vit_amd.functional as vf

    g = np.load(os.path.join(GOLD, "rope.npz"))
    for tag in ("a", "b"):
        B, H, T, dh, base, maxlen = (int(v) if i < 4 else float(v) for i, v in enumerate(g[f"{tag}_meta"]))
        q, k = torch.from_numpy(g[f"{tag}_q"]), torch.from_numpy(g[f"{tag}_k"])
        v = torch.randn(B, H, T, dh)
        tok = lambda t: t.transpose(1, 2).reshape(B * T, H * dh)          # [B,H,T,dh] -> token-major [B*T, H*dh]
        qkv = torch.cat([tok(q), tok(k), tok(v)], dim=1).to(dt).to(dev).contiguous()
        cos = torch.from_numpy(g[f"{tag}_cos"])[:, : dh // 2].contiguous().to(dev)
        sin = torch.from_numpy(g[f"{tag}_sin"])[:, : dh // 2].contiguous().to(dev)
        before = qkv.clone()
        vf.rope_qk(qkv, cos, sin, T, H, dh)
        D = H * dh
        tol = 2e-7 if dt == torch.float32 else 4e-3
        if dt == torch.bfloat16:  # the kernel rotates the bf16-rounded inputs
            from oracle import refvit
            c_full, s_full = torch.from_numpy(g[f"{tag}_cos"]), torch.from_numpy(g[f"{tag}_sin"])
            ref_q = refvit.apply_rope(q.to(dt).float(), c_full, s_full)
            ref_k = refvit.apply_rope(k.to(dt).float(), c_full, s_full)
        else:
            ref_q, ref_k = torch.from_numpy(g[f"{tag}_q_rot"]), torch.from_numpy(g[f"{tag}_k_rot"])
        assert rel(qkv[:, :D].float(), tok(ref_q)) < tol
        assert rel(qkv[:, D:2 * D].float(), tok(ref_k)) < tol
        assert torch.equal(qkv[:, 2 * D:], before[:, 2 * D:])              # v untouched
        vf.rope_qk(qkv, cos, sin, T, H, dh, inverse=True)                   # rotation by -angle undoes it
        assert rel(qkv.float(), before.float()) < (5e-7 if dt == torch.float32 else 6e-3)


@pytest.mark.parametrize("precision", ["32", "bf16-mixed"])
def test_rope_model_matches_oracle(dev, precision):
    """Full model with pos_encoding_type='rope' (vit_with_rope.py:58-60) against the p1 fixture (the restatement with
    the reference's RotaryPositionEmbedding doing the rotation): logits, first attention map, loss, every gradient."""
    from oracle import refvit
    from vit_amd.config import ViTConfig
    from vit_amd.specvit import MyViT

    g = np.load(os.path.join(GOLD, "rope.npz"))
    rc = refvit.RefConfig(image_size=640, patch_size=32, hidden_size=32, num_hidden_layers=2, num_attention_heads=2,
                          stride_size=32, pos_encoding_type="rope", rope_base=1000.0, loss_name="mae")
    sd = refvit.make_state_dict(rc, int(g["p1_wseed"]))
    cfg = ViTConfig(task_type="reg", image_size=640, patch_size=32, hidden_size=32, num_hidden_layers=2,
                    num_attention_heads=2, stride_size=32, pos_encoding_type="rope", rope_base=1000.0)
    model = MyViT(cfg, loss_name="mae")
    model.set_precision(precision)
    model.load_state_dict(sd, strict=True)
    model = model.to(dev).eval()
    x, labels = torch.from_numpy(g["p1_flux"]).to(dev), torch.from_numpy(g["p1_labels"]).to(dev)
    out = model(x, labels=labels, output_attentions=True)
    tight = precision == "32"
    assert rel(out.logits, torch.from_numpy(g["p1_logits"])) < (1e-4 if tight else 1.5e-2)
    assert rel(out.attentions[0], torch.from_numpy(g["p1_attn0"])) < (1e-4 if tight else 1.5e-2)
    # MSE loss: d(loss) = 2 * residual * d(logit); residual ~ sqrt(loss) = 0.26, logits ~ 1 with bf16 error <= 1.5e-2
    loss_tol = 1e-4 * float(g["p1_loss"]) if tight else 2 * float(g["p1_loss"]) ** 0.5 * 1.5e-2
    assert abs(float(out.loss) - float(g["p1_loss"])) <= loss_tol + 1e-6
    model(x, labels=labels).loss.backward()
    checked = 0
    for name, p in model.named_parameters():
        k = f"p1_grad/{name}"
        if k not in g.files:
            continue
        ref = torch.from_numpy(g[k])
        if float(ref.norm()) < 1e-6:
            continue
        assert p.grad is not None, name
        mine = p.grad.reshape(ref.shape).cpu()
        if tight:
            assert rel(mine, ref) < 2e-4, name
        else:
            # every gradient scales with dL/dlogit = 2 (logit - label) / n, whose own bf16 error is d(logit) / residual
            # = 1.5e-2 / 0.26 ~ 6 % on this fixture: bound the direction tightly and the magnitude by that
            cos = float(torch.dot(mine.flatten().double(), ref.flatten().double()) / (mine.double().norm() * ref.double().norm()))
            assert cos > 0.999 and rel(mine, ref) < 8e-2, (name, cos, rel(mine, ref))
        checked += 1
    assert checked > 20


def test_rope_model_with_the_rotating_epilogue(dev):
    """pos_encoding_type='rope' at a tile-aligned geometry (hidden 256, 4 heads of 64, T = 129, rows padded to 256): in
    bf16-mixed the fused QKV projection of every layer runs the ping-pong core's ROTATING epilogue (r04: the rotation is no longer
    a separate pass there; the backward still rotates dq / dk back in its own pass).  Against the CPU oracle run here on the same
    weights and inputs: logits, first-layer attention map, loss and every gradient, at the bf16 gates of the other model tests;
    and the library did launch the rotating epilogue."""
    from oracle import refvit
    from vit_amd import _cabi
    from vit_amd.config import ViTConfig
    from vit_amd.specvit import MyViT

    kw = dict(image_size=4096, patch_size=32, hidden_size=256, num_hidden_layers=2, num_attention_heads=4, stride_size=32,
              pos_encoding_type="rope", rope_base=10000.0)
    rc = refvit.RefConfig(loss_name="mae", **kw)
    sd = refvit.make_state_dict(rc, 5)
    x, _, labels = refvit.make_inputs(rc, 6, 9)
    params = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
    ref = refvit.forward(rc, params, x, labels, output_attentions=True)
    ref.loss.backward()
    grads = {k: p.grad.detach() for k, p in params.items() if p.grad is not None}
    model = MyViT(ViTConfig(task_type="reg", **kw), loss_name="mae")
    model.set_precision("bf16-mixed")
    model.load_state_dict(sd, strict=True)
    model = model.to(dev).eval()
    seen = []
    import vit_amd.functional as vf
    orig = vf.gemm

    def spy(a, b, **k):
        out = orig(a, b, **k)
        if k.get("rope") is not None:
            seen.append(_cabi.load().vit_last_gemm_kernel().decode())
        return out

    vf.gemm = spy
    try:
        out = model(x.to(dev), labels=labels.to(dev), output_attentions=True)
    finally:
        vf.gemm = orig
    assert len(seen) == 2 and all("<0, 0, 8, 8>" in s_ for s_ in seen), seen
    assert rel(out.logits, ref.logits.detach()) < 1.5e-2
    assert rel(out.attentions[0], ref.attentions[0].detach()) < 1.5e-2
    assert abs(float(out.loss) - float(ref.loss)) <= 2 * float(ref.loss) ** 0.5 * 1.5e-2 + 1e-6
    model(x.to(dev), labels=labels.to(dev)).loss.backward()
    checked = 0
    for name, p in model.named_parameters():
        if name not in grads or p.grad is None or float(grads[name].norm()) < 1e-6:
            continue
        mine, g_ = p.grad.reshape(grads[name].shape).cpu(), grads[name]
        cos = float(torch.dot(mine.flatten().double(), g_.flatten().double()) / (mine.double().norm() * g_.double().norm()))
        assert cos > 0.999 and rel(mine, g_) < 8e-2, (name, cos, rel(mine, g_))
        checked += 1
    assert checked > 20


@pytest.mark.parametrize("precision", ["32", "bf16-mixed"])
def test_padded_rows_do_not_leak(dev, precision):
    """hidden % 256 == 0 with B*T NOT a multiple of 256: the engine pads the GEMM row count to the 256-row tiles of the
    ping-pong core (zeroed pad rows); outputs, loss and every gradient must equal the oracle's on the real rows, twice in
    a row (the pad rows of the re-used buffers must stay inert), and for two batch sizes."""
    from oracle import refvit
    from vit_amd.config import ViTConfig
    from vit_amd.specvit import MyViT

    rc = refvit.RefConfig(image_size=288, patch_size=32, hidden_size=256, num_hidden_layers=2, num_attention_heads=4,
                          stride_size=32, loss_name="mae")
    sd = refvit.make_state_dict(rc, 61)
    cfg = ViTConfig(task_type="reg", image_size=288, patch_size=32, hidden_size=256, num_hidden_layers=2,
                    num_attention_heads=4, stride_size=32)
    model = MyViT(cfg, loss_name="mae")
    model.set_precision(precision)
    model.load_state_dict(sd, strict=True)
    model = model.to(dev).eval()
    tight = precision == "32"
    for B, seed in ((5, 62), (27, 63), (5, 62)):
        flux, _, labels = refvit.make_inputs(rc, B, seed)
        tr = refvit.RefTrainer(rc, sd, training=False)
        ref = refvit.forward(rc, tr.params, flux, labels)
        ref.loss.backward()
        model.zero_grad()
        out = model(flux.to(dev), labels=labels.to(dev))
        assert rel(out.logits, ref.logits.detach()) < (1e-4 if tight else 2e-2)
        out.loss.backward()
        for name, p in model.named_parameters():
            g = tr.params[name].grad
            if g is None or float(g.norm()) < 1e-6:
                continue
            mine = p.grad.reshape(g.shape).cpu()
            if tight:
                assert rel(mine, g) < 3e-4, (name, B)
            else:
                cos = float(torch.dot(mine.flatten().double(), g.flatten().double()) / (mine.double().norm() * g.double().norm()))
                assert cos > 0.995, (name, B, cos)


# ------------------------------------------------------------------ linear input preprocessors
def test_linear_preprocessor_forward_matches_reference(dev):
    """LinearPreprocessor forward (one vit_gemm) against the reference module's outputs in tests/golden/prep.npz."""
    from vit_amd.preprocessor import LinearPreprocessor

    g = np.load(os.path.join(GOLD, "prep.npz"))
    x, mean = torch.from_numpy(g["x"]).to(dev), torch.from_numpy(g["mean"])
    for key in ("zca_r16_s2", "pca_r16"):
        P = torch.from_numpy(g[key])
        pre = LinearPreprocessor(P, bias=-mean @ P.t(), freeze=True).to(dev)
        assert rel(pre(x), torch.from_numpy(g["y_" + key])) < 2e-5
        pre.set_precision("bf16-mixed")
        assert rel(pre(x), torch.from_numpy(g["y_" + key])) < 1.5e-2


@pytest.mark.parametrize("stride", [16, 8])
def test_trainable_preprocessor_gradients_and_step(dev, stride):
    """A TRAINABLE preprocessor in front of the ViT (freeze_epochs = 0): the gradient reaches it through the patch
    projection and the overlap-add that undoes the tokenizer's unfold (vit_fold_add; stride 8 < patch 16 overlaps);
    dP, dbias and every model gradient against autograd over the oracle, then one clipped AdamW step of everything."""
    import torch.nn.functional as F
    from oracle import refvit
    from vit_amd.config import ViTConfig
    from vit_amd.optimizer import FusedAdamW
    from vit_amd.preprocessor import LinearPreprocessor
    from vit_amd.specvit import MyViT

    L_in, r = 96, 64
    rc = refvit.RefConfig(image_size=r, patch_size=16, hidden_size=32, num_hidden_layers=2, num_attention_heads=2,
                          stride_size=stride, loss_name="mae")
    sd = refvit.make_state_dict(rc, 71)
    gen = torch.Generator().manual_seed(72)
    P0 = torch.randn(r, L_in, generator=gen) / L_in ** 0.5
    b0 = torch.randn(r, generator=gen) * 0.1
    x = torch.randn(5, L_in, generator=gen)
    labels = torch.rand(5, generator=gen)

    # oracle: F.linear in front of the restated model (specvit.py:72-73), autograd, clip 0.5, AdamW lr 1e-3
    Pr, br = P0.clone().requires_grad_(True), b0.clone().requires_grad_(True)
    params = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
    out_ref = refvit.forward(rc, params, F.linear(x, Pr, br), labels)
    out_ref.loss.backward()
    every = [p for p in params.values() if p.grad is not None] + [Pr, br]
    ref_grads = {id(p): p.grad.clone() for p in every}
    ref_norm = float(torch.nn.utils.clip_grad_norm_(every, 0.5))
    opt_ref = torch.optim.AdamW(every, lr=1e-3, weight_decay=0.0)
    opt_ref.step()

    cfg = ViTConfig(task_type="reg", image_size=r, patch_size=16, hidden_size=32, num_hidden_layers=2,
                    num_attention_heads=2, stride_size=stride)
    pre = LinearPreprocessor(P0.clone(), bias=b0.clone(), freeze=False)
    model = MyViT(cfg, loss_name="mae", preprocessor=pre)
    model.set_precision("32")
    model.load_state_dict({**sd, "preprocessor.linear.weight": P0, "preprocessor.linear.bias": b0}, strict=True)
    model = model.to(dev).eval()
    assert pre.linear.weight.is_cuda and isinstance(pre.linear.weight, torch.nn.Parameter)
    opt = FusedAdamW(model, lr=1e-3)
    opt.set_grad_clip(0.5)
    out = model(x.to(dev), labels=labels.to(dev))
    assert rel(out.logits, out_ref.logits.detach()) < 1e-4
    out.loss.backward()
    assert rel(pre.linear.weight.grad, ref_grads[id(Pr)]) < 3e-4
    assert rel(pre.linear.bias.grad, ref_grads[id(br)]) < 3e-4
    for name, p in model.named_parameters():
        if name in params and params[name].grad is not None and float(ref_grads[id(params[name])].norm()) > 1e-6:
            assert rel(p.grad.reshape(params[name].shape), ref_grads[id(params[name])]) < 3e-4, name
    opt.step()
    assert abs(float(opt.last_grad_norm.sqrt()) - ref_norm) < 2e-4 * ref_norm
    assert rel(pre.linear.weight.detach(), Pr.detach()) < 1e-5 and rel(pre.linear.bias.detach(), br.detach()) < 1e-4
    moved = float((pre.linear.weight.detach().cpu() - P0).abs().max())
    assert moved > 1e-4  # the preprocessor really was updated
    # frozen again: no parameters, no input gradient requested
    model.set_preprocessor_trainable(False)
    assert [n for n, _ in model.named_parameters() if n.startswith("preprocessor")] == []
    model(x.to(dev), labels=labels.to(dev)).loss.backward()


def test_prefilled_attention_forward_matches_reference(dev):
    """`warmup.preprocessor: attention` on 2-D spectra = the prefilled query projection (attention.py:81-84): one vit_gemm,
    against the reference module's outputs (prep.npz)."""
    from vit_amd.preprocessor import PrefilledAttention

    g = np.load(os.path.join(GOLD, "prep.npz"))
    vec, lam = torch.from_numpy(g["eigvecs"]), torch.from_numpy(g["eigvals"])
    x = torch.from_numpy(g["x"]).to(dev)
    for key, kw in {"attn_r16": dict(r=16, scale_by_eigvals=True), "attn_r24_noscale": dict(r=24, scale_by_eigvals=False)}.items():
        m = PrefilledAttention(48, vec, lam, eps=1e-5, **kw).to(dev)
        assert rel(m(x), torch.from_numpy(g[key + "_y"])) < 2e-5


@pytest.mark.parametrize("tag", ["a", "b"])
@pytest.mark.parametrize("precision", ["32", "bf16-mixed"])
def test_conv1d_tokenizer_matches_reference(dev, tag, precision):
    """proj_fn 'C1D' / 'CNN' -> Conv1DPatchTokenizer (tokenization.py:53-69): `num_patches = (L - P) // S + 1` (the
    remainder of the signal is DROPPED, no padded tail patch), weight [D, 1, P].  tests/golden/conv.npz holds the outputs
    of the reference's own module inside the reference's SpectraEmbeddings: case a = stride == patch, case b = stride <
    patch with a dropped remainder, 2 regression targets, L1 loss.  Eval forward (tokens, every hidden state, logits,
    loss) and every gradient (dropout off)."""
    from oracle import refvit
    from vit_amd.config import ViTConfig
    from vit_amd.specvit import MyViT

    rc = {
        "a": refvit.RefConfig(image_size=1024, patch_size=32, hidden_size=64, num_hidden_layers=2, num_attention_heads=2,
                              stride_size=32, proj_fn="C1D", loss_name="mae"),
        "b": refvit.RefConfig(image_size=1000, patch_size=64, hidden_size=64, num_hidden_layers=2, num_attention_heads=4,
                              stride_size=40, proj_fn="CNN", num_labels=2, loss_name="l1"),
    }[tag]
    g = np.load(os.path.join(GOLD, "conv.npz"))
    sd = refvit.make_state_dict(rc, int(g[f"{tag}_wseed"]))
    cfg = ViTConfig(task_type="reg", image_size=rc.image_size, patch_size=rc.patch_size, hidden_size=rc.hidden_size,
                    num_hidden_layers=rc.num_hidden_layers, num_attention_heads=rc.num_attention_heads, proj_fn=rc.proj_fn,
                    stride_size=rc.stride_size, num_labels=rc.num_labels)
    assert cfg.num_patches == g[f"{tag}_tokens"].shape[1]
    model = MyViT(cfg, loss_name=rc.loss_name)
    model.set_precision(precision)
    assert tuple(model.state_dict()["vit.embeddings.patch_embeddings.projection.weight"].shape) == (rc.hidden_size, 1, rc.patch_size)
    model.load_state_dict(sd, strict=True)
    model = model.to(dev).eval()
    x, labels = torch.from_numpy(g[f"{tag}_flux"]).to(dev), torch.from_numpy(g[f"{tag}_labels"]).to(dev)
    out = model(x, labels=labels, output_hidden_states=True)
    T = lambda k: torch.from_numpy(g[f"{tag}_{k}"])
    tol = 1e-4 if precision == "32" else 1.5e-2
    e_tok = rel(out.hidden_states[0][:, 1:], T("tokens"))
    e_hs = rel(torch.stack([h.cpu() for h in out.hidden_states]), T("hidden_states"))
    e_log = rel(out.logits, T("logits"))
    assert e_tok < (1e-4 if precision == "32" else 6e-3) and e_hs < tol and e_log < (1e-4 if precision == "32" else 2e-2), (e_tok, e_hs, e_log)
    assert abs(float(out.loss) - float(g[f"{tag}_loss"])) <= (2e-4 if precision == "32" else 3e-2) * float(g[f"{tag}_loss"]) + 1e-5
    model(x, labels=labels).loss.backward()
    worst = 0.0
    gtol = 2e-4 if precision == "32" else 4e-2
    for name, p in model.named_parameters():
        key = f"{tag}_grad/{name}"
        if key not in g.files:
            assert p.grad is None, name
            continue
        ref = torch.from_numpy(g[key])
        if float(ref.norm()) < 1e-6:
            continue
        e = rel(p.grad.reshape(ref.shape), ref)
        worst = max(worst, e)
        assert e < gtol, (name, e)
    print(f"[conv/{tag} {precision}] tokens {e_tok:.2e}, hidden states {e_hs:.2e}, logits {e_log:.2e}, worst grad {worst:.2e}")


def test_attention_forward_hooks_receive_maps(dev):
    """The reference's viz callback registers forward hooks on `...encoder.layer.N.attention.attention` and reads output[1]
    as the attention map (src/viz/viz_callback.py:183-214, 231-235).  MyViT has those module names; a registered hook is
    called with (context, probabilities) of that layer, equal to what output_attentions=True returns."""
    rc, g, sd, model, x, labels = setup("c1", dev, precision="32")
    model.eval()
    seen = {}
    names = [n for n, _ in model.named_modules() if "encoder.layer" in n and n.endswith(".attention.attention")]
    assert names == [f"vit.encoder.layer.{i}.attention.attention" for i in range(3)]
    mods = dict(model.named_modules())
    handles = [mods[n].register_forward_hook(lambda m, inp, out, n=n: seen.__setitem__(n, out)) for n in names[:2]]
    out = model(x, labels=labels)
    assert out.attentions is None and sorted(seen) == names[:2]
    ref = model(x, labels=labels, output_attentions=True)
    for i, n in enumerate(names[:2]):
        ctx, probs = seen[n]
        assert probs.shape == (x.shape[0], 2, 129, 129) and ctx.shape == (x.shape[0], 129, 32)
        assert torch.equal(probs, ref.attentions[i])
        assert rel(probs, torch.from_numpy(g["attn0"])) < 1e-4 if i == 0 else True
    for h in handles:
        h.remove()
    seen.clear()
    model(x, labels=labels)
    assert not seen


@pytest.mark.parametrize("precision,tol", [("32", 1e-2), ("bf16-mixed", 1.5e-1)])
def test_long_trajectory_tracks_the_oracle(dev, precision, tol):
    """120 optimisation steps (fwd -> bwd -> clip 0.5 -> AdamW 1e-3, dropout off) at C1 over 24 batches, the HIP path against
    the CPU oracle stepping from the same weights on the same batches: the loss curves must stay together for the whole run,
    not only for the three steps the fixtures hold.  Adam turns rounding differences into +-lr element steps, so the gate is
    on the curve, not on the weights: |loss - oracle loss| relative to the oracle's loss (floored at 5 % of the initial
    loss), worst step, for precision '32' (measured on MI355X: worst 1.7e-3, mean 1.8e-4, epoch means equal to 2e-4); for
    bf16-mixed the per-step losses of this small, noisy fit scatter (a 32-wide model on random labels), so the gate is on
    the five epoch means (measured: within 11 %; both curves fall 0.184 -> 0.09)."""
    from oracle import refvit
    from vit_amd.optimizer import FusedAdamW

    rc, g, sd, model, _, _ = setup("c1", dev, precision=precision)
    model.eval()  # dropout off on both sides; the engine still trains (the optimizer steps)
    gen = torch.Generator().manual_seed(77)
    batches = [(torch.randn(16, rc.image_size, generator=gen), torch.rand(16, generator=gen)) for _ in range(24)]
    ref = refvit.RefTrainer(rc, sd, lr=1e-3, training=False)
    opt = FusedAdamW(model, lr=1e-3)
    opt.set_grad_clip(0.5)
    worst, ours, theirs = 0.0, [], []
    for i in range(120):
        x, y = batches[i % 24]
        lr_ = ref.step(x, y)
        opt.zero_grad()
        loss = model(x.to(dev), labels=y.to(dev)).loss
        loss.backward()
        opt.step()
        ours.append(float(loss))
        theirs.append(lr_)
    floor = 0.05 * theirs[0]
    dev_ = [abs(a - b) / max(b, floor) for a, b in zip(ours, theirs)]
    ep_o = [float(np.mean(ours[e * 24:(e + 1) * 24])) for e in range(5)]
    ep_t = [float(np.mean(theirs[e * 24:(e + 1) * 24])) for e in range(5)]
    ep_dev = max(abs(a - b) / b for a, b in zip(ep_o, ep_t))
    print(f"[{precision}] 120 steps, epoch means oracle {[round(v, 4) for v in ep_t]} HIP {[round(v, 4) for v in ep_o]}: worst epoch "
          f"deviation {ep_dev:.2e}; per step: worst {max(dev_):.2e} at step {int(np.argmax(dev_))}, mean {np.mean(dev_):.2e}")
    assert ep_t[-1] < 0.5 * theirs[0]  # the run actually trains
    worst = max(dev_) if precision == "32" else ep_dev  # bf16: per-step losses of a noisy fit scatter; the epoch means are the curve
    assert worst < tol, (worst, ours[-5:], theirs[-5:])
