"""The trainer loop of the path (`Trainer.fit / validate / test`) through `ViTLModule` on the MI355X: VERDICT r1 #6, #7 and
ADVICE r1.  Reference behaviour: src/basemodule.py:203-251 (clip, validation every epoch, precision), src/vit.py:94-187
(eval step, epoch statistics), src/vit.py:365-424 (checkpoint / early stop), src/opt/optimizer.py:150-172 (plateau),
src/prepca/callbacks.py (freeze schedule), scripts/test.py:26-48 (evaluation only)."""
import copy
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def c1_config(**train):
    cfg = {
        "model": dict(name="vit", task_type="reg", image_size=4096, patch_size=32, hidden_size=32, num_hidden_layers=3,
                      num_attention_heads=2, stride_size=32, proj_fn="SW"),
        "train": dict(batch_size=16, ep=2, precision="32"),
        "loss": {"name": "mae"}, "opt": {"type": "AdamW", "lr": 1e-3}, "data": {"param": "log_g"},
        "noise": {"noise_level": 0},
    }
    cfg["train"].update(train)
    return cfg


class Batches:
    """A re-iterable list of (flux, error, labels) batches; the last one is short (40 = 16 + 16 + 8)."""

    def __init__(self, n, seed, bs=16, four=False):
        g = torch.Generator().manual_seed(seed)
        self.flux = torch.randn(n, 4096, generator=g)
        self.err = 0.1 * torch.rand(n, 4096, generator=g)
        self.lab = torch.rand(n, generator=g)
        self.bs, self.four = bs, four

    def __iter__(self):
        for i in range(0, self.flux.shape[0], self.bs):
            s = slice(i, i + self.bs)
            if self.four:
                yield self.flux[s] + 1.0, self.flux[s], self.err[s], self.lab[s]  # the "noisy" copy must NOT be used at nl=0
            else:
                yield self.flux[s], self.err[s], self.lab[s]


def make(cfg, seed=42):
    from vit_amd.module import ViTLModule
    from vit_amd.trainer import Trainer, seed_everything

    seed_everything(seed)
    return ViTLModule(config=cfg), Trainer(cfg["train"], device=torch.device("cuda", 0), verbose=False)


def test_fit_validation_metrics_match_fp64(dev):
    cfg = c1_config()
    module, trainer = make(cfg)
    val = Batches(40, 2, four=True)
    hist = trainer.fit(module, Batches(64, 1), val)
    assert len(hist) == 2 and trainer.global_step == 8
    # recompute everything from the final model's logits in float64 (the last validation saw exactly this model)
    module.eval()
    preds, labs, per_batch = [], [], []
    with torch.no_grad():
        for noisy, flux, err, lab in val:
            out = module.model(flux.cuda(), labels=lab.cuda())
            p, t = out.logits.squeeze().double().cpu(), lab.double()
            preds.append(p); labs.append(t)
            per_batch.append((len(t), float(((p - t) ** 2).mean()), 1 - float(((t - p) ** 2).sum() / ((t - t.mean()) ** 2).sum())))
    p, t = torch.cat(preds), torch.cat(labs)
    logs = hist[-1]
    mae, mse = float((p - t).abs().mean()), float(((p - t) ** 2).mean())
    r2_global = 1 - float(((t - p) ** 2).sum() / ((t - t.mean()) ** 2).sum())
    r2_logged = sum(n * r for n, _, r in per_batch) / sum(n for n, _, _ in per_batch)  # Lightning: batch-weighted mean
    assert abs(logs["val_mae"] - mae) < 1e-5 * mae and abs(logs["val_mse"] - mse) < 1e-5 * mse
    assert abs(logs["val_mae_loss"] - mse) < 1e-5 * mse      # loss.name 'mae' resolves to MSE (specvit.py:52-53)
    assert abs(logs["val_r2"] - r2_logged) < 1e-4 * max(1, abs(r2_logged))
    assert abs(trainer.metric_totals["val_r2"] - r2_global) < 1e-4 * max(1, abs(r2_global))
    assert abs(trainer.metric_totals["val_mae"] - mae) < 1e-5 * mae
    res = (p - t).numpy()
    assert abs(logs["val_bias_median"] - np.median(res)) < 1e-6
    assert abs(logs["val_p90"] - np.percentile(np.abs(res), 90)) < 1e-6
    assert abs(logs["val_beta"] - np.polyfit(t.numpy(), p.numpy(), 1)[0]) < 1e-6
    assert "mae_loss" in logs and logs["lr"] == 1e-3
    print(f"[fit] val_mae {logs['val_mae']:.5f} val_mse {logs['val_mse']:.5f} val_r2 {logs['val_r2']:.4f} "
          f"(epoch-total r2 {trainer.metric_totals['val_r2']:.4f})")


def test_plateau_scheduler_and_early_stop(dev):
    """ReduceLROnPlateau(factor, patience) on val_mae and EarlyStopping(patience) see the per-epoch monitored value: a scripted
    validation curve must give the learning-rate trace of torch's scheduler and stop where Lightning's counter would."""
    from vit_amd.trainer import Trainer

    cfg = c1_config(ep=12, patience=3)
    cfg["opt"].update(lr_sch="plateau", factor=0.5, patience=1)
    cfg["data"]["val_path"] = "/dev/null"  # the plateau guard only asks that a validation set is configured
    curve = [1.0, 0.8, 0.9, 0.85, 0.7, 0.75, 0.76, 0.77, 0.78, 0.79, 0.80, 0.81]

    class Scripted(Trainer):
        def validate(self, module, loader, prefix="val"):
            logs = super().validate(module, loader, prefix)
            logs["val_mae"] = curve[self.current_epoch]
            return logs

    from vit_amd.module import ViTLModule
    from vit_amd.trainer import seed_everything

    seed_everything(42)
    module = ViTLModule(config=cfg)
    trainer = Scripted(cfg["train"], device=torch.device("cuda", 0), verbose=False)
    hist = trainer.fit(module, Batches(16, 1), Batches(16, 2))
    # expectation from torch's own scheduler on a dummy optimizer
    dummy = torch.optim.SGD([torch.nn.Parameter(torch.zeros(1))], lr=1e-3)
    sch = torch.optim.lr_scheduler.ReduceLROnPlateau(dummy, factor=0.5, patience=1)
    want = []
    for v in curve:
        sch.step(v)
        want.append(dummy.param_groups[0]["lr"])
    got = [h["lr"] for h in hist]
    # best 0.7 at epoch 4; epochs 5, 6, 7 do not improve -> stop after epoch 7 (patience 3)
    assert len(hist) == 8 and trainer.should_stop
    assert got == want[:8], (got, want)
    assert got[-1] < 1e-3


def test_plateau_without_validation_set_is_dropped(dev):
    cfg = c1_config()
    cfg["opt"].update(lr_sch="plateau")
    module, trainer = make(cfg)
    with pytest.warns(UserWarning, match="plateau"):
        conf = module.configure_optimizers()
    assert not isinstance(conf, dict)  # optimizer only, no scheduler


def _cov_file(tmp_path, dim=4096, r=None):
    g = torch.Generator().manual_seed(5)
    q, _ = torch.linalg.qr(torch.randn(dim, 64, generator=g))
    path = tmp_path / "cov.pt"
    torch.save({"eigvecs": q.contiguous(), "eigvals": torch.logspace(0, -2, 64), "mean": 0.01 * torch.randn(dim, generator=g)}, path)
    return str(path)


def test_freeze_schedule_unfreezes_preprocessor(dev, tmp_path):
    """warmup.freeze_epochs = 1: the input preprocessor is frozen during epoch 0 and trainable from epoch 1
    (src/prepca/callbacks.py:31-60); its weights must not move while frozen and must move afterwards."""
    cfg = c1_config(ep=2)
    cfg["model"].update(image_size=4096)
    cfg["warmup"] = dict(preprocessor="pca", cov_path=_cov_file(tmp_path), r=64, freeze_epochs=1)
    cfg["model"]["patch_size"] = 8
    cfg["model"]["stride_size"] = 8
    module, trainer = make(cfg)
    assert module.model.name.startswith("PCA64_fz1") and module.model.config.image_size == 64
    pre = module.model.preprocessor
    seen = []
    step0 = module.training_step

    def spy(batch, idx):
        seen.append((trainer.current_epoch, pre._is_frozen, len(list(pre.parameters())), pre.linear.weight.detach().clone()))
        return step0(batch, idx)

    module.training_step = spy
    trainer.fit(module, Batches(32, 1), Batches(16, 2))
    e0 = [s for s in seen if s[0] == 0]
    e1 = [s for s in seen if s[0] == 1]
    assert len(e0) == 2 and len(e1) == 2
    assert all(s[1] and s[2] == 0 for s in e0)            # epoch 0: frozen = buffers, nothing for autograd
    assert all((not s[1]) and s[2] == 2 for s in e1)      # epoch 1 on: weight + bias are Parameters
    assert torch.equal(e0[0][3], e1[0][3])                # untouched while frozen
    w = pre.linear.weight
    assert isinstance(w, torch.nn.Parameter) and w.requires_grad and w.grad is not None and float(w.grad.abs().sum()) > 0
    assert trainer.freeze.released
    # permanent freeze never unfreezes
    cfg["warmup"]["freeze_epochs"] = -1
    cfg["model"]["image_size"] = 4096
    module2, trainer2 = make(cfg)
    trainer2.fit(module2, Batches(16, 1), Batches(16, 2))
    assert module2.model.name.startswith("PCA64_fzperm") and module2.model.preprocessor._is_frozen and not trainer2.freeze.released


def test_save_resume_and_eval_only(dev, tmp_path, monkeypatch):
    """--save writes the best-by-monitor checkpoint and last.ckpt in Lightning's layout; fit(ckpt_path=) continues a run
    exactly (weights, AdamW moments and step, epoch, dropout stream); test(ckpt_path=) evaluates without touching a weight
    and applies train.precision itself (scripts/test.py:26-48)."""
    from vit_amd.trainer import load_checkpoint_file

    monkeypatch.setenv("CKPT_DIR", str(tmp_path / "ck"))
    train, val = Batches(48, 1), Batches(24, 2)
    # uninterrupted 3 epochs
    cfg = c1_config(ep=3, save=False, precision="bf16-mixed")
    m_full, t_full = make(cfg)
    t_full.fit(m_full, train, val)
    # 2 epochs with saving, then resume for the third in a fresh process state
    cfg2 = c1_config(ep=2, save=True, precision="bf16-mixed")
    m_a, t_a = make(cfg2)
    t_a.fit(m_a, train, val)
    ck = t_a.checkpointer
    assert os.path.basename(ck.last_path) == "last.ckpt" and os.path.exists(ck.best_path)
    assert os.path.basename(ck.best_path).startswith("epoch=") and "val_mae=" in ck.best_path
    files = sorted(os.listdir(tmp_path / "ck"))
    assert len(files) == 2, files  # save_top_k=1 + last
    raw = load_checkpoint_file(ck.last_path)  # weights_only=True loader
    assert raw["epoch"] == 1 and raw["global_step"] == 6
    assert all(k.startswith("model.") for k in raw["state_dict"])
    assert "model.vit.encoder.layer.0.attention.attention.query.weight" in raw["state_dict"]
    assert set(raw["optimizer_states"][0]) == {"state", "param_groups"}
    cfg3 = c1_config(ep=3, save=False, precision="bf16-mixed")
    m_b, t_b = make(cfg3, seed=42)
    hist = t_b.fit(m_b, train, val, ckpt_path=ck.last_path)
    assert len(hist) == 1 and t_b.global_step == 9
    for (n1, p1), (n2, p2) in zip(m_full.model.state_dict().items(), m_b.model.state_dict().items()):
        assert n1 == n2 and torch.equal(p1.cpu(), p2.cpu()), n1
    # ---- evaluation only
    m_c, t_c = make(c1_config(precision="bf16-mixed"), seed=7)
    logs = t_c.test(m_c, val, ckpt_path=ck.last_path)
    assert m_c.model.engine.precision == "bf16"   # Trainer.test applies train.precision itself
    sd_ck = {k[len("model."):]: v for k, v in raw["state_dict"].items()}
    for n, p in m_c.model.state_dict().items():
        assert torch.equal(p.cpu(), sd_ck[n]), n  # bit-identical to the checkpoint: nothing was trained
    assert t_c.optimizer is None
    m_c.eval()
    with torch.no_grad():
        ps, ts = [], []
        for flux, err, lab in val:
            ps.append(m_c.model(flux.cuda(), labels=lab.cuda()).logits.squeeze().double().cpu()); ts.append(lab.double())
    p, t = torch.cat(ps), torch.cat(ts)
    assert abs(logs["test_mae"] - float((p - t).abs().mean())) < 1e-6
    assert abs(logs["test_mse"] - float(((p - t) ** 2).mean())) < 1e-6


def test_fused_adamw_state_is_torch_adamw_state(dev):
    """FusedAdamW.state_dict() loads into torch.optim.AdamW (and back): after one fused step, both optimizers continue
    identically from that state on the same gradients."""
    from vit_amd.optimizer import FusedAdamW

    module, trainer = make(c1_config())
    model = module.model.to(dev)
    model.set_precision("32")
    model.eval()
    b = next(iter(Batches(16, 3)))
    opt = FusedAdamW(model, lr=1e-3, weight_decay=0.01)
    model(b[0].cuda(), labels=b[2].cuda()).loss.backward()
    opt.step()
    sd = opt.state_dict()
    twins = [torch.nn.Parameter(p.detach().clone()) for p in model.parameters()]
    ref = torch.optim.AdamW(twins, lr=1e-3, weight_decay=0.01)
    ref.load_state_dict(copy.deepcopy(sd))
    opt.zero_grad()
    model(b[0].cuda(), labels=b[2].cuda()).loss.backward()
    for tw, p in zip(twins, model.parameters()):
        tw.grad = None if p.grad is None else p.grad.detach().clone()
    opt.step()
    ref.step()
    worst = max(float((tw - p).abs().max()) for tw, p in zip(twins, model.parameters()))
    assert worst < 2e-7, worst
    # and back: torch's state -> a fresh FusedAdamW
    opt2 = FusedAdamW(model, lr=1e-3, weight_decay=0.01)
    opt2.load_state_dict(ref.state_dict())
    assert opt2._step == 2
    name = "vit.encoder.layer.1.intermediate.dense.weight"
    i = [n for n, _ in model.named_parameters()].index(name)
    assert torch.allclose(opt2.state_dict()["state"][i]["exp_avg"], ref.state_dict()["state"][i]["exp_avg"].cpu(), atol=1e-9)


def test_fused_adamw_skips_frozen_and_gradless(dev):
    """torch.optim semantics: a parameter with requires_grad=False, and every parameter when no backward ran,
    keeps its value and its moments."""
    from vit_amd.optimizer import FusedAdamW

    module, _ = make(c1_config())
    model = module.model.to(dev)
    model.set_precision("32")
    model.eval()
    opt = FusedAdamW(model, lr=1e-2)
    before = {n: p.detach().clone() for n, p in model.named_parameters()}
    opt.step()  # no backward yet: nothing may move
    assert all(torch.equal(before[n], p) for n, p in model.named_parameters()) and opt._step == 0
    frozen = ["vit.encoder.layer.1.intermediate.dense.weight", "vit.embeddings.cls_token"]
    for n, p in model.named_parameters():
        if n in frozen:
            p.requires_grad_(False)
    b = next(iter(Batches(16, 3)))
    model(b[0].cuda(), labels=b[2].cuda()).loss.backward()
    opt.step()
    moved = {n: not torch.equal(before[n], p) for n, p in model.named_parameters()}
    assert not any(moved[n] for n in frozen)
    assert moved["vit.encoder.layer.1.output.dense.weight"] and moved["vit.encoder.layer.0.intermediate.dense.weight"]
    assert not moved["vit.pooler.dense.weight"]
    off, _ = model.engine.layout.entries[frozen[0]]
    assert float(opt._m[off:off + 100].abs().max()) == 0.0


def test_engine_keeps_one_forward(dev):
    """a second forward before backward invalidates the first one's activations -> error, not wrong gradients;
    num_hidden_layers = 0 trains; output_hidden_states with labels keeps the loss differentiable."""
    from vit_amd._cabi import VitError
    from vit_amd.config import ViTConfig
    from vit_amd.specvit import MyViT

    module, _ = make(c1_config())
    model = module.model.to(dev)
    model.set_precision("32")
    b = next(iter(Batches(16, 3)))
    x, y = b[0].cuda(), b[2].cuda()
    l1 = model(x, labels=y).loss
    l2 = model(x, labels=y).loss
    l2.backward()  # the latest forward is fine
    with pytest.raises((VitError, RuntimeError), match="activations"):
        l1.backward()
    out = model(x, labels=y, output_hidden_states=True, output_attentions=True)
    assert out.loss.requires_grad and len(out.hidden_states) == 4 and out.attentions[0].shape == (16, 2, 129, 129)
    out.loss.backward()
    # zero encoder layers
    cfg0 = ViTConfig(task_type="reg", image_size=512, patch_size=32, hidden_size=32, num_hidden_layers=0, num_attention_heads=2,
                     stride_size=32)
    m0 = MyViT(cfg0, loss_name="mae").to(dev)
    m0.set_precision("32")
    m0.eval()
    xs = torch.randn(4, 512, device=dev)
    ys = torch.rand(4, device=dev)
    m0(xs, labels=ys).loss.backward()
    w = m0.state_dict()
    ln = torch.nn.functional.layer_norm
    emb = torch.cat([w["vit.embeddings.cls_token"].expand(4, -1, -1),
                     xs.view(4, 16, 32) @ w["vit.embeddings.patch_embeddings.projection.weight"].t()
                     + w["vit.embeddings.patch_embeddings.projection.bias"]], 1).detach().requires_grad_(True)
    g = w["vit.layernorm.weight"].detach().clone().requires_grad_(True)
    ref = torch.nn.functional.mse_loss((ln(emb, (32,), g, w["vit.layernorm.bias"], 1e-12)[:, 0] @ w["regressor.weight"].t()
                                        + w["regressor.bias"]).view(-1), ys)
    ref.backward()
    got = dict(m0.named_parameters())["vit.layernorm.weight"].grad
    assert float((got - g.grad).norm() / g.grad.norm()) < 1e-4


def test_fp16_precision_is_refused(dev):
    module, _ = make(c1_config())
    with pytest.raises(ValueError, match="fp16"):
        module.model.set_precision("16-mixed")


@pytest.mark.parametrize("precision", ["32", "bf16-mixed"])
def test_hip_graph_step_equals_eager(dev, precision):
    """train.hip_graph: the captured step (vit_amd/graph.py) must be the eager step.  With dropout switched off in the
    config both paths are deterministic functions of (weights, batch): 4 graph replays == 4 eager steps bit for bit, incl.
    AdamW's bias corrections (step count read from the device record) and a mid-run learning-rate change; the capture's
    warm-up steps leave no trace.  With dropout on, every replay draws new masks."""
    from vit_amd.graph import GraphedTrainStep

    batches = list(Batches(64, 5))
    finals = {}
    for mode in ("eager", "graph"):
        cfg = c1_config(precision=precision, hip_graph=(mode == "graph"))
        module, trainer = make(cfg)
        module.model.config.hidden_dropout_prob = 0.0
        module.model.config.attention_probs_dropout_prob = 0.0
        trainer._setup(module)
        module.train()
        losses = []
        for i in range(4):
            if i == 2:
                trainer.optimizer.param_groups[0]["lr"] = 3e-4
            b = tuple(t.cuda() for t in batches[i % 2])
            losses.append(float(trainer.training_step(module, b, i)))
        finals[mode] = (losses, {k: v.detach().cpu().clone() for k, v in module.model.state_dict().items()},
                        trainer.optimizer._step, float(trainer.optimizer.last_grad_norm))
        if mode == "graph":
            assert len(trainer._graphed) == 1 and all(isinstance(g, GraphedTrainStep) for g in trainer._graphed.values())
    assert finals["eager"][0] == finals["graph"][0], (finals["eager"][0], finals["graph"][0])
    assert finals["eager"][2] == finals["graph"][2] == 4
    assert finals["eager"][3] == finals["graph"][3]
    for k in finals["eager"][1]:
        assert torch.equal(finals["eager"][1][k], finals["graph"][1][k]), k
    # dropout on: same inputs, same weights (lr = 0) -> different masks on every replay, hence different losses
    cfg = c1_config(precision=precision, hip_graph=True)
    cfg["opt"]["lr"] = 0.0
    module, trainer = make(cfg)
    trainer._setup(module)
    module.train()
    b = tuple(t.cuda() for t in batches[0])
    seen = {round(float(trainer.training_step(module, b, i)), 7) for i in range(4)}
    assert len(seen) == 4, seen


@pytest.mark.parametrize("precision", ["32", "bf16-mixed"])
def test_hip_graph_survives_validation_and_shape_changes(dev, precision):
    """a validation pass (another arena: need_grad = False), a partial last batch (another batch shape)
    and a workspace re-allocation between replays must not leave the captured graph on freed memory: the run replay ->
    validate -> replay (short batch) -> grow the workspace -> replay equals the eager run bit for bit, and the graph of the
    full batch is NOT re-captured on the way (same object, same arena tensors held)."""
    from vit_amd import _cabi

    batches = list(Batches(40, 5))  # 16, 16, 8
    val = Batches(24, 6, four=True)
    finals = {}
    for mode in ("eager", "graph"):
        cfg = c1_config(precision=precision, hip_graph=(mode == "graph"))
        module, trainer = make(cfg)
        module.model.config.hidden_dropout_prob = 0.0
        module.model.config.attention_probs_dropout_prob = 0.0
        trainer._setup(module)
        module.train()
        losses, first = [], None
        for rnd in range(2):
            for i, b in enumerate(batches):
                losses.append(float(trainer.training_step(module, tuple(t.cuda() for t in b), i)))
                if mode == "graph" and first is None:
                    first = next(iter(trainer._graphed.values()))
                    held = [t.data_ptr() for t in first._held[1].values() if torch.is_tensor(t)]
            logs = trainer.validate(module, val)  # evaluation forward between replays
            losses.append(logs["val_mae"])
            h = _cabi.handle_for(torch.device("cuda", 0))
            h.set_workspace(h.workspace_bytes + (1 << 20))  # the handle moves to another workspace allocation
            torch.cuda.empty_cache()
            junk = torch.full((64 << 20,), float("nan"), device="cuda")  # whatever was freed gets poisoned
            del junk
        finals[mode] = (losses, {k: v.detach().cpu().clone() for k, v in module.model.state_dict().items()})
        if mode == "graph":
            assert len(trainer._graphed) == 2 and next(iter(trainer._graphed.values())) is first
            assert held == [t.data_ptr() for t in first._held[1].values() if torch.is_tensor(t)]
            assert trainer.use_graph
    assert finals["eager"][0] == finals["graph"][0], (finals["eager"][0], finals["graph"][0])
    for k in finals["eager"][1]:
        assert torch.equal(finals["eager"][1][k], finals["graph"][1][k]), k


def test_hip_graph_replay_rezeroes_dlast(dev):
    """`d last_hidden` is zero-filled by a kernel inside the captured step (a hipMemsetAsync node was observed
    not to run before its consumers).  Poison the buffer with NaN before a replay: the replay must still produce finite
    gradients equal to the eager step's."""
    cfg = c1_config(precision="bf16-mixed", hip_graph=True)
    module, trainer = make(cfg)
    module.model.config.hidden_dropout_prob = 0.0
    module.model.config.attention_probs_dropout_prob = 0.0
    trainer._setup(module)
    module.train()
    b = tuple(t.cuda() for t in next(iter(Batches(16, 5))))
    trainer.training_step(module, b, 0)
    g = next(iter(trainer._graphed.values()))
    eng = module.model.engine
    for _ in range(3):
        g._held[1]["dlast"].fill_(float("nan"))
        trainer.training_step(module, b, 0)
        assert torch.isfinite(eng.grads[:eng.layout.n_trainable]).all()
        assert torch.isfinite(trainer.optimizer.last_grad_norm).all()
    cfg2 = c1_config(precision="bf16-mixed")
    module2, trainer2 = make(cfg2)
    module2.model.config.hidden_dropout_prob = 0.0
    module2.model.config.attention_probs_dropout_prob = 0.0
    trainer2._setup(module2)
    module2.train()
    for _ in range(4):
        trainer2.training_step(module2, b, 0)
    for (k, v), (_, w) in zip(module.model.state_dict().items(), module2.model.state_dict().items()):
        assert torch.equal(v, w), k


def test_hip_graph_falls_back_to_eager_with_noise(dev):
    """a configuration the capture cannot take (on-the-fly noise) trains eagerly with one warning instead of
    raising out of fit()."""
    cfg = c1_config(precision="bf16-mixed", hip_graph=True)
    cfg["noise"] = {"noise_level": 0.5}
    module, trainer = make(cfg)
    trainer._setup(module)
    module.train()
    b = tuple(t.cuda() for t in next(iter(Batches(16, 5))))
    with pytest.warns(UserWarning, match="eager"):
        l0 = float(trainer.training_step(module, b, 0))
    assert not trainer.use_graph and np.isfinite(l0)
    assert np.isfinite(float(trainer.training_step(module, b, 1)))


def test_cli_run_save_then_test_only(dev, tmp_path):
    """`launch.sh run --save` then `launch.sh test --ckpt last` (scripts/run.py, scripts/test.py): the test entry evaluates the
    checkpoint without training (reference: scripts/test.py:26-48) and reports the same test metrics the run
    printed after its own fit."""
    import re
    import subprocess
    import sys

    import yaml

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    cfg = c1_config(ep=2, precision="bf16-mixed")
    cfg["project"] = "t"
    path = tmp_path / "c.yaml"
    path.write_text(yaml.safe_dump(cfg))
    env = dict(os.environ, CKPT_DIR=str(tmp_path / "ck"))
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE"):
        env.pop(k, None)
    r1 = subprocess.run(["bash", os.path.join(root, "launch.sh"), "run", "-c", str(path), "-g", "1", "--save", "--synthetic", "256"],
                        capture_output=True, text=True, env=env, timeout=600)
    assert r1.returncode == 0, r1.stderr[-2000:]
    assert sorted(f for f in os.listdir(tmp_path / "ck") if f.endswith(".ckpt"))[-1] == "last.ckpt"
    m1 = dict(re.findall(r"(test_\w+)=([-\d.e+]+)", [l for l in r1.stdout.splitlines() if l.startswith("[test] ")][-1]))
    r2 = subprocess.run(["bash", os.path.join(root, "launch.sh"), "test", "-c", str(path), "-g", "1", "--ckpt", "last", "--synthetic", "256"],
                        capture_output=True, text=True, env=env, timeout=600)
    assert r2.returncode == 0, r2.stderr[-2000:]
    assert "[epoch" not in r2.stdout  # nothing was trained
    m2 = dict(re.findall(r"(test_\w+)=([-\d.e+]+)", [l for l in r2.stdout.splitlines() if l.startswith("[test] ") and "=" in l][-1]))
    assert m1.keys() == m2.keys() and "test_mae" in m1
    for k in m1:
        assert abs(float(m1[k]) - float(m2[k])) <= 1e-4 * max(1.0, abs(float(m1[k]))), (k, m1[k], m2[k])


def test_cli_file_backed_data_run_then_test_best(dev, tmp_path):
    """VERDICT r3 #7 (SURVEY 8f-3 on the GPU box): `launch.sh run -c cfg --save` WITHOUT --synthetic -- the config's
    data.file_path / val_path / test_path go through SpecDataModule (reference: scripts/run.py:32-50 -> src/vit.py:29-42;
    label normalisation with the TRAINING split's statistics, src/dataloader/spec_datasets.py:73-91; fixed-seed evaluation
    noise, base.py:312-326) -- then `launch.sh test --ckpt best`.  The split files carry the arrays the reference's own
    RegSpecDataset was fed (tests/golden/data.npz).  Checked: both entries print the same test metrics, and those equal a
    float64 recomputation with the CPU oracle from the saved checkpoint on the fixture's reference-normalised labels and
    reference-made noisy test spectra."""
    import re
    import subprocess

    import yaml

    from oracle import refvit
    from vit_amd.trainer import load_checkpoint_file, model_state_from_checkpoint

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    g = np.load(os.path.join(root, "tests", "golden", "data.npz"))
    paths = {}
    for split, sfx in (("train", "tr"), ("val", "va"), ("test", "va")):
        path = tmp_path / f"{split}.npz"
        np.savez(path, flux=g[f"flux_{sfx}"], error=g[f"err_{sfx}"], log_g=g[f"one_p_{sfx}"])
        paths[split] = str(path)
    cfg = {
        "project": "t",
        "model": dict(name="vit", task_type="reg", image_size=40, patch_size=8, hidden_size=32, num_hidden_layers=2,
                      num_attention_heads=2, stride_size=8, proj_fn="SW"),
        "train": dict(batch_size=5, ep=3, precision="32"),
        "loss": {"name": "mae"}, "opt": {"type": "AdamW", "lr": 1e-3},
        "data": {"file_path": paths["train"], "val_path": paths["val"], "test_path": paths["test"], "num_samples": 100,
                 "num_test_samples": 100, "param": "log_g", "label_norm": "minmax"},
        "noise": {"noise_level": 0.5},
    }
    cpath = tmp_path / "c.yaml"
    cpath.write_text(yaml.safe_dump(cfg))
    env = dict(os.environ, CKPT_DIR=str(tmp_path / "ck"))
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE"):
        env.pop(k, None)
    r1 = subprocess.run(["bash", os.path.join(root, "launch.sh"), "run", "-c", str(cpath), "-g", "1", "--save"],
                        capture_output=True, text=True, env=env, timeout=600)
    assert r1.returncode == 0, r1.stderr[-2000:]
    assert "loading data from " + paths["train"] in r1.stdout and "synthetic" not in r1.stdout.lower()
    m1 = dict(re.findall(r"(test_\w+)=([-\d.e+]+)", [l for l in r1.stdout.splitlines() if l.startswith("[test] ")][-1]))
    r2 = subprocess.run(["bash", os.path.join(root, "launch.sh"), "test", "-c", str(cpath), "-g", "1", "--ckpt", "best"],
                        capture_output=True, text=True, env=env, timeout=600)
    assert r2.returncode == 0, r2.stderr[-2000:]
    assert "[epoch" not in r2.stdout
    tl = [l for l in r2.stdout.splitlines() if l.startswith("[test] ")]
    ck = re.search(r"ckpt=(\S+)", tl[0]).group(1)
    assert os.path.basename(ck).startswith("epoch=") and os.path.exists(ck)  # 'best' resolved to the monitored checkpoint
    m2 = dict(re.findall(r"(test_\w+)=([-\d.e+]+)", [l for l in tl if "=" in l][-1]))
    assert "test_mae" in m2 and "test_mse" in m2
    # float64 recomputation from the checkpoint alone: oracle forward on the reference-made noisy test spectra, labels as the
    # reference's dataset normalised them with the TRAINING split's min / max
    sd = {k: v.double() for k, v in model_state_from_checkpoint(load_checkpoint_file(ck)).items()}
    rc = refvit.RefConfig(image_size=40, patch_size=8, hidden_size=32, num_hidden_layers=2, num_attention_heads=2,
                          stride_size=8, loss_name="mae")
    noisy = torch.from_numpy(g["one_minmax_noisy_va"]).double()
    lab = torch.from_numpy(g["one_minmax_labels_va"]).double()
    # independent check of the statistics' origin: (p - min_train) / (max_train - min_train)
    ptr, pva = g["one_p_tr"].astype(np.float64), g["one_p_va"].astype(np.float64)
    assert np.allclose((pva - ptr.min()) / (ptr.max() - ptr.min()), lab.numpy(), rtol=1e-6, atol=1e-7)
    pred = refvit.forward(rc, sd, noisy, None).logits.view(-1)
    mae, mse = float((pred - lab).abs().mean()), float(((pred - lab) ** 2).mean())
    for m in (m1, m2) if os.path.basename(ck) == "last.ckpt" else (m2,):
        assert abs(float(m["test_mae"]) - mae) <= 2e-4 * max(1.0, abs(mae)), (m["test_mae"], mae)
        assert abs(float(m["test_mse"]) - mse) <= 2e-4 * max(1.0, abs(mse)), (m["test_mse"], mse)
    # the run's own final test used the LAST weights; re-evaluating `last` reproduces what it printed
    r3 = subprocess.run(["bash", os.path.join(root, "launch.sh"), "test", "-c", str(cpath), "-g", "1", "--ckpt", "last"],
                        capture_output=True, text=True, env=env, timeout=600)
    assert r3.returncode == 0, r3.stderr[-2000:]
    m3 = dict(re.findall(r"(test_\w+)=([-\d.e+]+)", [l for l in r3.stdout.splitlines() if l.startswith("[test] ") and "=" in l][-1]))
    for k in m1:
        assert abs(float(m1[k]) - float(m3[k])) <= 1e-4 * max(1.0, abs(float(m1[k]))), (k, m1[k], m3[k])


def test_fused_adamw_refuses_replaced_grad_under_reducer(dev):
    """under data parallelism the flat gradient buffer holds the rank-averaged gradient when step() runs; a
    `.grad` that is no longer the flat buffer's view (a hook / an accumulation replaced it) must not silently overwrite it."""
    from vit_amd.optimizer import FusedAdamW

    module, _ = make(c1_config())
    model = module.model.to(dev)
    model.set_precision("32")
    model.eval()
    b = next(iter(Batches(16, 3)))
    opt = FusedAdamW(model, lr=1e-3)
    model(b[0].cuda(), labels=b[2].cuda()).loss.backward()
    name = "vit.encoder.layer.0.output.dense.weight"
    p = dict(model.named_parameters())[name]
    p.grad = p.grad.clone()  # what a gradient hook or `loss1.backward(); loss2.backward()` accumulation leaves behind

    class Reducer:  # the one attribute FusedAdamW looks at
        mode = "allreduce"

    opt.attach_reducer(Reducer())
    with pytest.raises(RuntimeError, match="flat gradient buffer"):
        opt.step()
    opt.attach_reducer(None)  # single process: the replaced gradient is folded back and the step proceeds
    before = p.detach().clone()
    opt.step()
    assert not torch.equal(before, p)


def test_full_size_vit_b_training_step(dev):
    """The BASELINE configuration itself under -m gpu ('C3 at the full B = 256 runs only in bench.py'):
    ViT-B/16 224^2 restated, B = 256, bf16-mixed, one rank.  Size-independent properties at full size:
    (1) with dropout off the step is a deterministic function of (weights, batch): two runs from the same seed give the same
        three losses and the same clipped gradient norms bit for bit (reference: deterministic=True, src/basemodule.py:250);
    (2) with dropout off the training-mode loss of step 0 is the evaluation-mode loss of the same weights and batch;
    (3) the loss is a mean over samples: L(256) = mean of the four quarter-batch losses to fp32 rounding (per-sample
        independence: nothing in the path mixes samples), and of the two half-batch losses to bf16 rounding (that shape takes
        the split-K tail path);
    (4) with dropout on, three steps stay finite, clip to 0.5 and move every trainable tensor."""
    import vit_amd.functional  # noqa: F401
    from vit_amd.module import ViTLModule
    from vit_amd.trainer import Trainer, seed_everything

    def cfg(B):
        return {
            "model": dict(name="vit", task_type="reg", image_size=50176, patch_size=256, hidden_size=768, num_hidden_layers=12,
                          num_attention_heads=12, stride_size=256, proj_fn="SW"),
            "train": dict(batch_size=B, ep=1, precision="bf16-mixed"),
            "loss": {"name": "mae"}, "opt": {"type": "AdamW", "lr": 1e-3}, "data": {"param": "log_g"},
            "noise": {"noise_level": 0},
        }

    g = torch.Generator().manual_seed(1234)
    flux, err, lab = torch.randn(256, 50176, generator=g).cuda(), (0.1 * torch.rand(256, 50176, generator=g)).cuda(), \
        torch.rand(256, generator=g).cuda()

    def run(B, dropout, steps, sl=slice(None)):
        seed_everything(42)
        module = ViTLModule(config=cfg(B))
        if not dropout:
            module.model.config.hidden_dropout_prob = 0.0
            module.model.config.attention_probs_dropout_prob = 0.0
        trainer = Trainer(cfg(B)["train"], device=torch.device("cuda", 0), verbose=False)
        trainer._setup(module)
        module.eval()
        with torch.no_grad():
            ev = float(module(flux[sl], labels=lab[sl]))
        module.train()
        w0 = {k: v.detach().clone() for k, v in module.model.state_dict().items()}
        out = []
        for i in range(steps):
            out.append((float(trainer.training_step(module, (flux[sl], err[sl], lab[sl]), i)),
                        float(trainer.optimizer.last_grad_norm)))
        return ev, out, w0, module

    ev_a, a, _, _ = run(256, False, 3)
    ev_b, b, _, _ = run(256, False, 3)
    assert a == b and ev_a == ev_b, (a, b)                                   # (1)
    assert abs(a[0][0] - ev_a) <= 1e-6 * max(1.0, abs(ev_a)), (a[0][0], ev_a)  # (2)
    quarters = [run(64, False, 0, slice(i, i + 64))[0] for i in range(0, 256, 64)]
    assert abs(sum(quarters) / 4 - ev_a) <= 2e-6 * max(1.0, abs(ev_a)), (quarters, ev_a)  # (3): same kernels, same sums per row
    # B = 128 is 99 row tiles: 297 tiles = one round + 41, a short tail -> FC2 / dX run their tail tiles as K-slices
    # (gemm_split_tail), whose f32 partial sums are added in another order: equal to bf16 rounding, not bit for bit
    ev_lo, _, _, _ = run(128, False, 0, slice(0, 128))
    ev_hi, _, _, _ = run(128, False, 0, slice(128, 256))
    assert abs(0.5 * (ev_lo + ev_hi) - ev_a) <= 3e-3 * abs(ev_a), (ev_lo, ev_hi, ev_a)
    _, d, w0, module = run(256, True, 3)
    assert all(np.isfinite(l) and np.isfinite(n) for l, n in d), d           # (4)
    moved = [k for k, v in module.model.state_dict().items() if not torch.equal(v, w0[k])]
    frozen = [k for k in w0 if k not in moved]
    assert all("pooler" in k for k in frozen), frozen  # the pooler is never used (SURVEY section 8 e): everything else moved


def test_full_size_vit_l_training_step(dev):
    """BASELINE config 5 at its own batch under -m gpu ('C5 at its BASELINE batch (B = 32) runs only in the
    builder's bench.py'): ViT-L/16 384^2 restated (L = 147456, 576 + 1 tokens, 24 x 1024, 16 heads, MLP 4096), B = 32,
    bf16-mixed, one rank -- the shape that takes the resident attention kernels at T = 577 (two-kernel backward), the 292-tile
    products with a 36-tile split-K tail and the D = 1024 LayerNorm forms.  Size-independent properties at full size:
    (1) dropout off: two runs from the same seed give the same losses and clipped gradient norms bit for bit;
    (2) dropout off: the training-mode loss of step 0 is the evaluation-mode loss of the same weights and batch;
    (3) the loss is a mean over samples: L(32) = mean of the two half-batch losses to bf16 rounding (other tile counts, hence
        other summation orders in the tail tiles);
    (4) dropout on: three steps stay finite, the clipped norm is reported, every trainable tensor moves.  (The loss itself
        jumps after the first AdamW step -- 303 M weights each move by lr along the gradient's sign with no warm-up, in the
        reference as here -- so it is not asserted to fall.)"""
    import vit_amd.functional  # noqa: F401
    from vit_amd.module import ViTLModule
    from vit_amd.trainer import Trainer, seed_everything

    L = 147456

    def cfg(B):
        return {
            "model": dict(name="vit", task_type="reg", image_size=L, patch_size=256, hidden_size=1024, num_hidden_layers=24,
                          num_attention_heads=16, stride_size=256, proj_fn="SW"),
            "train": dict(batch_size=B, ep=1, precision="bf16-mixed"),
            "loss": {"name": "mae"}, "opt": {"type": "AdamW", "lr": 1e-4}, "data": {"param": "log_g"},
            "noise": {"noise_level": 0},
        }

    g = torch.Generator().manual_seed(4321)
    flux, err, lab = torch.randn(32, L, generator=g).cuda(), (0.1 * torch.rand(32, L, generator=g)).cuda(), \
        torch.rand(32, generator=g).cuda()

    def run(B, dropout, steps, sl=slice(None)):
        seed_everything(42)
        module = ViTLModule(config=cfg(B))
        assert sum(p.numel() for p in module.parameters()) > 300e6  # 303.6 M parameters
        if not dropout:
            module.model.config.hidden_dropout_prob = 0.0
            module.model.config.attention_probs_dropout_prob = 0.0
        trainer = Trainer(cfg(B)["train"], device=torch.device("cuda", 0), verbose=False)
        trainer._setup(module)
        module.eval()
        with torch.no_grad():
            ev = float(module(flux[sl], labels=lab[sl]))
        module.train()
        w0 = {k: v.detach().clone() for k, v in module.model.state_dict().items()} if dropout else None
        out = []
        for i in range(steps):
            out.append((float(trainer.training_step(module, (flux[sl], err[sl], lab[sl]), i)),
                        float(trainer.optimizer.last_grad_norm)))
        return ev, out, w0, module

    ev_a, a, _, m = run(32, False, 2)
    del m
    ev_b, b, _, m = run(32, False, 2)
    del m
    assert a == b and ev_a == ev_b, (a, b)                                    # (1)
    assert abs(a[0][0] - ev_a) <= 1e-6 * max(1.0, abs(ev_a)), (a[0][0], ev_a)   # (2)
    ev_lo, _, _, m = run(16, False, 0, slice(0, 16))
    del m
    ev_hi, _, _, m = run(16, False, 0, slice(16, 32))
    del m
    assert abs(0.5 * (ev_lo + ev_hi) - ev_a) <= 5e-3 * abs(ev_a), (ev_lo, ev_hi, ev_a)  # (3)
    torch.cuda.empty_cache()
    _, d, w0, module = run(32, True, 3)
    assert all(np.isfinite(l) and np.isfinite(n) and n > 0 for l, n in d), d   # (4)
    moved = [k for k, v in module.model.state_dict().items() if not torch.equal(v, w0[k])]
    frozen = [k for k in w0 if k not in moved]
    assert all("pooler" in k for k in frozen), frozen


def test_bench_contract_single_gpu(dev):
    """The driver's contract on the line `bench.py` prints (one JSON object on stdout): metric / value / unit / n_gpus / steps /
    warmup / ms_per_step / higher_is_better / scaling / vs_baseline / dtype / data / config.workload, `roofline` with
    bound / achieved / peak / unit / frac / traffic, `cpu_baseline` with value / unit / cores / kind / sample, and the round-4
    `timing` object: `value` comes from the MEDIAN per-step hipEvent time, the region mean is reported beside it, the start-up
    heap is frozen and no garbage collection ran inside the timed region.  (Tiny workload: a contract test, not a measurement.)"""
    import json
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "VIT_DIST_SINGLE"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--steps", "6", "--warmup", "2", "--workload", "vit_tiny16_32",
                        "--batch", "256"], env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, r.stdout[-2000:]
    doc = json.loads(lines[0])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
              "dtype", "data", "config", "roofline", "cpu_baseline", "timing", "mean_ms_per_step"):
        assert k in doc, k
    assert doc["n_gpus"] == 1 and doc["steps"] == 6 and doc["warmup"] == 2 and doc["higher_is_better"] is True
    assert doc["scaling"] == "weak" and doc["vs_baseline"] is None and doc["dtype"] == "bf16" and doc["data"] == "synthetic"
    assert "workload" in doc["config"] and "model" not in doc["config"]
    rf = doc["roofline"]
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert k in rf, k
    assert rf["bound"] == "mfma" and rf["unit"] == "TFLOP/s" and abs(rf["frac"] - rf["achieved"] / rf["peak"]) < 1e-3
    cb = doc["cpu_baseline"]
    for k in ("value", "unit", "cores", "kind", "sample"):
        assert k in cb, k
    assert cb["kind"] == "port" and cb["cores"] >= 1 and cb["value"] > 0
    tm = doc["timing"]
    assert len(tm["step_ms"]) == 6 and tm["heap_frozen"] is True and tm["gc_in_region"] == []
    srt = sorted(tm["step_ms"])
    assert abs(doc["ms_per_step"] - 0.5 * (srt[2] + srt[3])) < 2e-3
    assert abs(doc["value"] - 256 / (doc["ms_per_step"] * 1e-3)) <= 1e-3 * doc["value"]
    assert abs(doc["mean_ms_per_step"] * 6 - tm["region_wall_ms"]) < 1e-2 * tm["region_wall_ms"]


@pytest.mark.parametrize("opt_cfg", [{"type": "AdamW", "lr": 1e-3, "lr_sch": "onecycle"}, {"type": "SGD", "lr": 1e-2}])
def test_snapshot_restore_puts_every_piece_of_training_state_back(dev, opt_cfg):
    """`Trainer._snapshot / _restore` (what `train.ddp_reserve_cus: auto` wraps its probe steps in): after three real
    optimisation steps with dropout and noise on, a restore leaves parameters, optimizer state and step count, scheduler,
    global_step, the dropout stream, the log sums and torch's generator exactly where they were -- the next three steps then
    equal, bit for bit, the three steps of a run that never probed.  Fused AdamW with a per-step scheduler, and a torch
    optimizer (the reference's other `opt.type`s take that branch)."""
    cfg = c1_config(precision="bf16-mixed")
    cfg["opt"] = dict(opt_cfg)
    cfg["noise"] = {"noise_level": 0.3}
    cfg["data"]["num_samples"] = 40
    data = list(Batches(40, seed=5))

    def run(probe):
        m, t = make(copy.deepcopy(cfg))
        t._setup(m)
        m.train()
        b = [tuple(x.to(dev) for x in bb) for bb in data]
        t.training_step(m, b[0], 0)  # some state to protect
        if probe:
            snap = t._snapshot(m)
            for i in range(3):
                t.training_step(m, b[2], i)
            t._restore(m, snap)
        losses = [float(t.training_step(m, b[1 + (i % 2)], i)) for i in range(3)]
        opt = t.optimizer
        state = [m.model.engine.flat.detach().cpu().clone(), t.global_step, m.model.engine.step_counter,
                 opt.param_groups[0]["lr"], torch.random.get_rng_state().clone()]
        if hasattr(opt, "_m"):
            state += [opt._m.detach().cpu().clone(), opt._v.detach().cpu().clone(), opt._step]
        return losses, state

    l0, s0 = run(False)
    l1, s1 = run(True)
    assert l0 == l1
    for a, b in zip(s0, s1):
        assert torch.equal(a, b) if torch.is_tensor(a) else a == b


def test_launch_geometry_belongs_to_the_engine(dev):
    """`reserve_cus` used to be process-wide.  Two engines of one process with different settings: each
    handle keeps its own (vit_handle_set_option), results are identical bit for bit whatever the other engine was told, and a
    value out of range is an error of the call."""
    from vit_amd import _cabi
    from vit_amd.config import ViTConfig
    from vit_amd.specvit import MyViT

    cfg = ViTConfig(task_type="reg", image_size=50176 // 4, patch_size=256, hidden_size=768, num_hidden_layers=1,
                    num_attention_heads=12, stride_size=256, num_labels=1)
    x = torch.randn(64, cfg.image_size, device=dev)
    y = torch.rand(64, device=dev)
    outs = []
    models = []
    for reserve in (-1, 32):
        torch.manual_seed(3)
        m = MyViT(cfg, loss_name="mae")
        m.set_precision("bf16-mixed")
        m.to(dev).eval()
        m.engine.set_reserve_cus(reserve)
        models.append(m)
    for m in models + models[::-1]:  # interleaved: one engine's setting must not leak into the other's launches
        m.zero_grad(set_to_none=True)
        loss = m(x, labels=y).loss
        loss.backward()
        outs.append((float(loss), m.engine.grads.clone()))
    assert models[0].engine.handle() is not models[1].engine.handle()
    assert outs[0][0] == outs[3][0] and torch.equal(outs[0][1], outs[3][1])
    assert outs[1][0] == outs[2][0] and torch.equal(outs[1][1], outs[2][1])
    with pytest.raises(_cabi.VitError):
        models[0].engine.handle().set_option("reserve_cus", 500)
    with pytest.raises(_cabi.VitError):
        models[0].engine.handle().set_option("gemm_core", 1)  # not a per-handle option


def test_bench_line_carries_secondary_and_input_pipeline(dev):
    """the default `python bench.py` line also carries `secondary` (C5 and C2 timed in the same process),
    `input_pipeline` (the same steps fed by SpecLoader from a host-resident and a device-resident split) and the cost of the
    per-GEMM event brackets, which exist only in the repetition behind `roofline`.  Run at the headline workload with a small
    batch (the keys are what is checked, not the numbers)."""
    import json
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "VIT_DIST_SINGLE"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--steps", "4", "--warmup", "2", "--no-cpu-baseline"],
                       env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, r.stdout[-2000:]
    doc = json.loads(lines[0])
    assert doc["input"] == "resident in HBM" and doc["config"]["launch"].startswith("eager launches")
    sec = doc["secondary"]
    assert set(sec) == {"vit_l16_384", "vit_tiny16_32"}
    for v in sec.values():
        assert v["value"] > 0 and v["ms_per_step"] > 0 and 0 < v["step_frac"] < 1
    ip = doc["input_pipeline"]
    for k in ("host_ms_per_step", "device_ms_per_step", "host_vs_resident_batch", "device_vs_resident_batch", "h2d_bytes_per_step_host"):
        assert k in ip, ip
    assert ip["h2d_bytes_per_step_host"] == 256 * (50176 + 1) * 4  # flux + labels: the error tensor stays on the host
    rf = doc["roofline"]
    assert rf["event_cost_ms_per_step"] is not None and rf["one_stream_uninstrumented_ms_per_step"] > 0
    assert rf["instrumented_ms_per_step"] > rf["one_stream_uninstrumented_ms_per_step"] - 0.5


@pytest.mark.parametrize("placement", ["device", "host"])
def test_classification_fit_through_the_loader(dev, tmp_path, monkeypatch, placement):
    """The classification branch of the path end to end (reference: task_type 'cls' -> CrossEntropyLoss, specvit.py:45-48;
    labels = log_g > 2.5, spec_datasets.py:24; `val_acc` monitored with mode 'max', vit.py:386-424): SpecDataset makes the int64
    labels, SpecLoader produces the batches on the device (both placements), the trainer validates every epoch and keeps the
    best checkpoint by `val_acc`.  The logged accuracy and loss equal the ones recomputed in float64 from the final model's
    logits; the best checkpoint is the epoch with the HIGHEST accuracy."""
    from vit_amd.data import SpecDataset, SpecLoader
    from vit_amd.module import ViTLModule
    from vit_amd.trainer import Trainer, seed_everything

    monkeypatch.setenv("CKPT_DIR", str(tmp_path))
    cfg = c1_config(ep=3, save=True, precision="bf16-mixed")
    cfg["model"].update(task_type="cls", num_labels=2)
    cfg["loss"] = {"name": "ce"}
    g = torch.Generator().manual_seed(31)
    mk = lambda n, stage: SpecDataset(torch.rand((n, 4096), generator=g), 0.1 * torch.rand((n, 4096), generator=g),
                                      5.0 * torch.rand((n,), generator=g), task="cls", stage=stage)
    tr, va = mk(96, "train"), mk(40, "val")
    assert tr.labels.dtype == torch.int64 and set(tr.labels.tolist()) == {0, 1}
    seed_everything(42)
    module = ViTLModule(config=cfg)
    trainer = Trainer(cfg["train"], device=dev, verbose=False)
    assert module.monitor_metric == "acc"
    hist = trainer.fit(module, SpecLoader(tr, 16, shuffle=True, placement=placement), SpecLoader(va, 16, placement=placement))
    assert len(hist) == 3 and trainer.monitor == "val_acc" and trainer.monitor_mode == "max"
    module.eval()
    with torch.no_grad():
        out = module.model(va.flux.to(dev), labels=va.labels.to(dev))
    logits = out.logits.double().cpu()
    acc = float((logits.argmax(-1) == va.labels).double().mean())
    ce = float(torch.nn.functional.cross_entropy(logits, va.labels))
    assert abs(hist[-1]["val_acc"] - acc) < 1e-9 and abs(hist[-1]["val_ce_loss"] - ce) < 2e-5 * max(1.0, ce)
    best = max(range(3), key=lambda e: (hist[e]["val_acc"], -e))  # the first epoch that reached the highest accuracy
    assert trainer.checkpointer.best_score == hist[best]["val_acc"]
    assert os.path.basename(trainer.checkpointer.best_path).startswith(f"epoch={best}-val_acc=")
    assert os.path.exists(os.path.join(str(tmp_path), "last.ckpt"))


def test_noise_run_through_the_loader(dev):
    """noise.noise_level > 0 (reference: vit.py:86-88 training-time injection on the device, base.py:312-326 fixed-seed noisy copy
    for evaluation): the bound loader ships `error` for the training split (the step reads it now), the validation 4-tuples carry
    the noisy copy and the module evaluates THAT one -- the logged validation loss equals the loss recomputed on `noisy`, and
    differs from the loss on the clean flux."""
    from vit_amd.data import SpecDataset, SpecLoader, _step_reads
    from vit_amd.module import ViTLModule
    from vit_amd.trainer import Trainer, seed_everything

    cfg = c1_config(ep=1)
    cfg["noise"] = {"noise_level": 0.5}
    g = torch.Generator().manual_seed(8)
    mk = lambda n, stage: SpecDataset(torch.rand((n, 4096), generator=g), 0.3 * torch.rand((n, 4096), generator=g),
                                      torch.rand((n,), generator=g), task="reg", stage=stage, noise_level=0.5)
    tr, va = mk(48, "train"), mk(32, "val")
    assert _step_reads(tr) == (True, True, True) and va.noisy is not None
    seed_everything(42)
    module = ViTLModule(config=cfg)
    trainer = Trainer(cfg["train"], device=dev, verbose=False)
    hist = trainer.fit(module, SpecLoader(tr, 16, shuffle=True, placement="host"), SpecLoader(va, 16, placement="device"))
    module.eval()
    with torch.no_grad():
        on_noisy = float(module.model(va.noisy.to(dev), labels=va.labels.to(dev)).loss)
        on_clean = float(module.model(va.flux.to(dev), labels=va.labels.to(dev)).loss)
    assert abs(hist[-1]["val_mae_loss"] - on_noisy) < 1e-5 * max(1.0, on_noisy)
    assert abs(on_noisy - on_clean) > 1e-6
    assert all(torch.isfinite(torch.tensor(v)) for v in hist[-1].values())
