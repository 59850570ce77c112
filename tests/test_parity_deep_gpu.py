"""Full-model parity AT THE BENCHMARKED GEOMETRIES (VERDICT r1 "next" #1): C3 = ViT-B/16 224^2
restated (L 50176, P 256, T 197, D 768, 12 heads, 12 layers) and C5 = ViT-L/16 384^2 restated (L 147456, T 577, D 1024,
16 heads, 24 layers), reference path src/models/specvit.py:68-94.

Two checkers per case:
  * tests/golden/{c3,c5}.npz -- written by oracle/make_golden.py from the reference's own composition (get_vit_config +
    HF ViTModel + the reference's SpectraEmbeddings) at that depth: norms of every hidden state, 16 sampled token rows of
    three hidden states and of the final LayerNorm output, sampled attention rows, logits, loss (fp32 AND the reference's
    own bf16-autocast outputs), per-parameter gradient norms + 64 sampled entries, a 2-step clipped-AdamW trajectory;
  * oracle/refvit.py run on the CPU inside the test on the same seeded inputs: EVERY hidden state, the first / last
    attention maps and EVERY gradient tensor, full size.

Tolerances: the same gates as tests/test_parity_gpu.py.  precision '32': rel-L2 <= 1e-4 forward (north_star asks 1e-3),
2e-4 per gradient tensor.  precision 'bf16-mixed': hidden states <= 1.5e-2, logits no worse than 1.5 x the reference's
own bf16-autocast error + 1e-3; gradients: every tensor within 1.5 x the WORST per-tensor error of the reference's own
bf16-autocast gradients at that depth (fixture `bf16_grad_err`: 1.4e-2 at C3, 1.1e-2 at C5) + 2e-3, cosine >= 0.999.  Batch 4 / 5 (C3) and 2 (C5) give 788 / 985 /
1154 token rows: all three run the GEMMs on zero-padded 256-row tiles, which is the padded-row case at the real T.
"""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

GOLD = os.path.join(os.path.dirname(__file__), "golden")
CASES = {"c3": ("C3", 4), "c5": ("C5", 2),
         # four corners of the reference's own sweep space (configs/sweep.yaml:10-21, image_size 4096), same fixture recipe:
         # s1 = patch 8 / stride 1 / hidden 32 / 8 heads (T 4090, head_dim 4), s2 = patch 64 / stride 1 / hidden 128 / 2 heads
         # (T 4034, head_dim 64: the tiled attention kernels inside a model, unpadded row count 8068), s3 = Conv1D tokenizer,
         # patch 16 / stride 2 / hidden 128 / 8 heads (T 2042, head_dim 16), s4 = patch 256 / stride 32 / hidden 32 / 4 heads
         # (T 122, head_dim 8, 6 layers)
         "s1": ("S1", 2), "s2": ("S2", 4), "s3": ("S3", 8), "s4": ("S4", 16)}
ALL = ["c3", "c5", "s1", "s2", "s3", "s4"]
_cache = {}


def rel(a, b):
    a, b = torch.as_tensor(a).double().cpu(), torch.as_tensor(b).double().cpu()
    return float((a - b).norm() / (b.norm() + 1e-30))


def bf16_factor(tag):
    """bf16-mixed gate = factor x the reference's own bf16-autocast error + 1e-3.  C3 / C5: 1.15 (measured 0.37-0.99 x).  Sweep
    corners: 1.5 -- with 2..16 samples both our error and the yardstick are small-sample draws of the same rounding noise
    (one near-zero residual logit - label moves every gradient of that sample); measured on MI355X, logits / worst gradient:
    s1 0.65 / 1.22 x, s2 0.37 / 0.89 x, s3 1.34 / 1.26 x, s4 0.23 / 0.91 x."""
    return 1.5 if tag.startswith("s") else 1.15


def oracle_run(tag, batch=None):
    """One CPU oracle forward + backward (dropout off) per case, shared by the tests of this module."""
    key = (tag, batch)
    if key in _cache:
        return _cache[key]
    from oracle import refvit

    name, b0 = CASES[tag]
    rc = refvit.named_config(name)
    g = np.load(os.path.join(GOLD, f"{tag}.npz"))
    B = batch or b0
    sd = refvit.make_state_dict(rc, int(g["wseed"]))
    flux, _, labels = refvit.make_inputs(rc, B, int(g["xseed"]))
    if batch is None:
        assert abs(sum(float(v.double().sum()) for v in sd.values()) - float(g["weight_checksum"])) < 1e-6
        assert abs(float(flux.double().sum()) - float(g["flux_checksum"])) < 1e-6
        assert np.array_equal(labels.numpy(), g["labels"])
    torch.set_num_threads(max(1, min(16, len(os.sched_getaffinity(0)))))
    tr = refvit.RefTrainer(rc, sd, training=False)
    out = refvit.forward(rc, tr.params, flux, labels, output_hidden_states=True, output_attentions=True)
    out.loss.backward()
    res = dict(rc=rc, g=g, sd=sd, flux=flux, labels=labels, ref_bf16_grad_err=g["bf16_grad_err"],
               hs=[h.detach() for h in out.hidden_states], last=out.last_hidden_state.detach(),
               attn0=out.attentions[0].detach(), attnL=out.attentions[-1].detach(),
               logits=out.logits.detach(), loss=float(out.loss.detach()),
               grads={k: (None if p.grad is None else p.grad.detach()) for k, p in tr.params.items()})
    _cache.clear()  # one case resident at a time (C5: 1.2 GB of gradients)
    _cache[key] = res
    return res


def build(o, dev, precision):
    from vit_amd.config import ViTConfig
    from vit_amd.specvit import MyViT

    rc = o["rc"]
    cfg = ViTConfig(task_type=rc.task_type, image_size=rc.image_size, patch_size=rc.patch_size, hidden_size=rc.hidden_size,
                    num_hidden_layers=rc.num_hidden_layers, num_attention_heads=rc.num_attention_heads,
                    proj_fn=rc.proj_fn, stride_size=rc.stride_size, num_labels=rc.num_labels)
    model = MyViT(cfg, loss_name=rc.loss_name)
    model.set_precision(precision)
    model.load_state_dict(o["sd"], strict=True)
    return model.to(dev).eval()


@pytest.mark.parametrize("precision", ["bf16-mixed", "32"])  # the top decorator varies fastest: one oracle run per tag
@pytest.mark.parametrize("tag", ALL)
def test_deep_eval_forward(dev, tag, precision):
    o = oracle_run(tag)
    g, rc = o["g"], o["rc"]
    model = build(o, dev, precision)
    out = model(o["flux"].to(dev), labels=o["labels"].to(dev), output_hidden_states=True, output_attentions=True)
    D, T = rc.hidden_size, rc.seq_len
    errs = [rel(a, b) for a, b in zip(out.hidden_states, o["hs"])]
    e_att = max(rel(out.attentions[0], o["attn0"]), rel(out.attentions[-1], o["attnL"]))
    e_logits, e_loss = rel(out.logits, o["logits"]), abs(float(out.loss) - o["loss"]) / abs(o["loss"])
    # the fixture (reference composition): norms of every hidden state, sampled rows, sampled attention rows, logits
    rows = torch.from_numpy(g["rows"])
    nrm = [abs(float(h.double().norm()) - n) / n for h, n in zip(out.hidden_states, g["hs_norms"])]
    e_rows = max(rel(out.hidden_states[int(i)].reshape(-1, D)[rows.to(dev)], g["hs_rows"][j])
                 for j, i in enumerate(g["hs_layers"]))
    arow = torch.from_numpy(g["attn_rows_idx"]).to(dev)
    e_arow = max(rel(out.attentions[0].reshape(-1, T)[arow], g["attn0_rows"]),
                 rel(out.attentions[-1].reshape(-1, T)[arow], g["attn_last_rows"]))
    e_fix_logits = rel(out.logits, g["logits"])
    # yardstick of the bf16 gate: the reference's own bf16-autocast error on the logits -- a relative error over `batch`
    # numbers, i.e. noisy for the 2..16-sample sweep cases (s1: 6.9e-4 over two logits while its hidden states sit at
    # 5.6e-3) -- so never less than its error on the sampled rows of the final LayerNorm output the logits are read from
    e_ref_last = rel(g["bf16_last_rows"], g["last_rows"])
    e_ref_bf16 = max(rel(g["bf16_logits"], g["logits"]), e_ref_last)
    print(f"[{tag} {precision}] hidden states max {max(errs):.2e} (layer {int(np.argmax(errs))}), attention {e_att:.2e}, "
          f"logits {e_logits:.2e}, loss {e_loss:.2e}; vs fixture: rows {e_rows:.2e}, attn rows {e_arow:.2e}, "
          f"logits {e_fix_logits:.2e} (reference's own bf16 autocast: {e_ref_bf16:.2e})")
    if precision == "32":
        assert max(errs) < 1e-4 and e_att < 1e-4 and e_rows < 1e-4 and e_arow < 1e-4, (errs, e_att, e_rows, e_arow)
        assert e_logits < 1e-4 and e_fix_logits < 1e-4 and e_loss < 2e-4 and max(nrm) < 1e-4
    else:
        assert max(errs) < 1.5e-2 and e_rows < 1.5e-2 and max(nrm) < 5e-3, (errs, e_rows, nrm)
        assert e_att < 2e-2 and e_arow < 2e-2, (e_att, e_arow)
        # measured 0.63-0.82 x the reference's own bf16-autocast error; the gate sits just above 1 x
        assert max(e_logits, e_fix_logits) <= bf16_factor(tag) * e_ref_bf16 + 1e-3, (e_logits, e_fix_logits, e_ref_bf16)
        # MSE: d(loss) ~ 2 residual d(logit); bound the loss through the measured logit error
        lg = torch.from_numpy(g["logits"]).double().flatten()
        floor = 4.0 * o["loss"] ** 0.5 * e_logits * float(lg.pow(2).mean().sqrt())
        assert abs(float(out.loss) - o["loss"]) <= 3e-2 * o["loss"] + floor


def _check_grads(model, o, precision, tag):
    g = o["g"]
    names = [str(n) for n in g["param_names"]] if g is not None else None
    worst, worst_name, worst_cos = 0.0, "", 1.0
    tol, tol_cos = (2e-4, 1 - 1e-7) if precision == "32" else (4e-2, 0.999)
    ref_bf16 = float(np.max(o["ref_bf16_grad_err"]))
    if precision != "32":
        tol = bf16_factor(tag) * ref_bf16 + 1e-3  # measured 0.85 (C3) / 0.99 (C5) x the reference's WORST bf16 gradient error
    gmax = max(float(v.norm()) for v in o["grads"].values() if v is not None)
    for name, p in model.named_parameters():
        ref = o["grads"][name]
        if ref is None:
            assert p.grad is None, name  # pooler: unused output (specvit.py:78)
            continue
        mine = p.grad.detach().double().cpu().flatten()
        r = ref.double().flatten()
        if float(r.norm()) < 1e-6 * gmax:  # key.bias: analytically zero
            assert float(mine.norm()) < 2e-3 * gmax, name
            continue
        e = float((mine - r).norm() / r.norm())
        cos = float(torch.dot(mine, r) / (mine.norm() * r.norm()))
        if e > worst:
            worst, worst_name = e, name
        worst_cos = min(worst_cos, cos)
        assert e < tol and cos > tol_cos, (name, e, cos)
        if names is not None:  # the fixture: norm + 64 sampled entries from the reference's autograd
            i = names.index(name)
            assert abs(float(mine.norm()) - g["grad_norms"][i]) <= tol * g["grad_norms"][i] + 1e-9, name
            idx = torch.from_numpy(g["grad_idx"][i])
            es = rel(mine[idx], g["grad_samples"][i])
            assert es < 3 * tol + 1e-6, (name, es)
    print(f"[{tag} {precision}] worst gradient rel err {worst:.2e} ({worst_name}), worst cosine {worst_cos:.6f}; "
          f"the reference's own bf16-autocast gradients: worst {ref_bf16:.2e}")


@pytest.mark.parametrize("precision", ["bf16-mixed", "32"])  # the top decorator varies fastest: one oracle run per tag
@pytest.mark.parametrize("tag", ALL)
def test_deep_gradients(dev, tag, precision):
    o = oracle_run(tag)
    model = build(o, dev, precision)
    loss = model(o["flux"].to(dev), labels=o["labels"].to(dev)).loss
    loss.backward()
    _check_grads(model, o, precision, tag)


def test_deep_c3_batch5_padded_rows(dev):
    """B = 5 -> 985 token rows on 1024-row GEMM tiles (39 pad rows), bf16-mixed: forward + every gradient vs the oracle."""
    o = dict(oracle_run("c3", batch=5))
    o["g"] = None
    model = build(o, dev, "bf16-mixed")
    out = model(o["flux"].to(dev), labels=o["labels"].to(dev), output_hidden_states=True)
    errs = [rel(a, b) for a, b in zip(out.hidden_states, o["hs"])]
    assert max(errs) < 1.5e-2, errs
    loss = model(o["flux"].to(dev), labels=o["labels"].to(dev)).loss
    loss.backward()
    _check_grads(model, o, "bf16-mixed", "c3/B5")


@pytest.mark.parametrize("precision", ["bf16-mixed", "32"])
def test_deep_c3_training_steps(dev, precision):
    """Two steps of fwd -> bwd -> clip 0.5 -> AdamW(1e-3) at C3 depth, dropout off, against the fixture's trajectory."""
    from vit_amd.optimizer import FusedAdamW

    o = oracle_run("c3")
    g = o["g"]
    model = build(o, dev, precision)
    opt = FusedAdamW(model, lr=1e-3)
    opt.set_grad_clip(0.5)
    x, y = o["flux"].to(dev), o["labels"].to(dev)
    for s in range(len(g["step_losses"])):
        opt.zero_grad()
        loss = model(x, labels=y).loss
        loss.backward()
        opt.step()
        gn = float(opt.last_grad_norm.sqrt())
        print(f"[c3 {precision}] step {s}: loss {float(loss):.6f} (ref {g['step_losses'][s]:.6f}), "
              f"grad norm {gn:.4f} (ref {g['step_grad_norms'][s]:.4f})")
        tl, tg = (5e-4, 5e-4) if precision == "32" else (5e-2, 5e-2)
        assert abs(float(loss) - g["step_losses"][s]) <= tl * g["step_losses"][s] + (0 if precision == "32" else 2e-3)
        assert abs(gn - g["step_grad_norms"][s]) <= tg * g["step_grad_norms"][s]
