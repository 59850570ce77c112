"""CPU suite for the host-side mirror of the reference surface that needs no kernel: epoch metrics, residual statistics,
config loading, run names of the model factory, checkpoint bookkeeping."""
import os

import numpy as np
import pytest
import torch


def test_metrics_match_definitions_and_accumulate():
    from vit_amd.metrics import Accuracy, MeanAbsoluteError, MeanSquaredError, R2Score

    g = torch.Generator().manual_seed(0)
    t = torch.rand(100, generator=g)
    p = t + 0.1 * torch.randn(100, generator=g)
    mae, mse, r2 = MeanAbsoluteError(), MeanSquaredError(), R2Score()
    chunks = [(0, 30), (30, 64), (64, 100)]
    for a, b in chunks:
        v_mae, v_mse, v_r2 = mae(p[a:b], t[a:b]), mse(p[a:b], t[a:b]), r2(p[a:b], t[a:b])
        tt, pp = t[a:b].double(), p[a:b].double()
        assert abs(float(v_mae) - float((pp - tt).abs().mean())) < 1e-6      # the call returns the BATCH value
        assert abs(float(v_r2) - (1 - float(((tt - pp) ** 2).sum() / ((tt - tt.mean()) ** 2).sum()))) < 1e-5
    td, pd = t.double(), p.double()
    assert abs(float(mae.compute()) - float((pd - td).abs().mean())) < 1e-6  # compute() is over everything seen
    assert abs(float(mse.compute()) - float(((pd - td) ** 2).mean())) < 1e-6
    assert abs(float(r2.compute()) - (1 - float(((td - pd) ** 2).sum() / ((td - td.mean()) ** 2).sum()))) < 1e-5
    assert mae.n == 3
    mae.reset()
    assert mae.n == 0
    with pytest.raises(RuntimeError):
        mae.compute()
    acc = Accuracy()
    logits = torch.tensor([[2.0, 1.0, 0.0], [0.0, 3.0, 1.0], [0.0, 0.0, 1.0], [1.0, 0.0, 0.0]])
    assert float(acc(logits, torch.tensor([0, 1, 0, 0]))) == 0.75 and float(acc.compute()) == 0.75


def test_residual_stats_match_numpy():
    from vit_amd.module import ViTLModule

    rng = np.random.default_rng(1)
    for n in (7, 40):
        lab = rng.random(n)
        pred = 0.2 + 0.8 * lab + 0.05 * rng.standard_normal(n)
        st = ViTLModule._residual_stats(torch.from_numpy(pred), torch.from_numpy(lab))
        res = pred - lab
        assert abs(st["bias_median"] - np.median(res)) < 1e-12
        assert abs(st["p90"] - np.percentile(np.abs(res), 90)) < 1e-12
        assert abs(st["beta"] - np.polyfit(lab, pred, 1)[0]) < 1e-9


def test_load_config_expands_and_unwraps_wandb_exports(tmp_path, monkeypatch):
    from vit_amd.utils import load_config

    monkeypatch.setenv("VIT_TEST_ROOT", "/data/x")
    f = tmp_path / "c.yaml"
    f.write_text("data:\n  file_path: ${VIT_TEST_ROOT}/train.h5\n  list: ['~/a', 3]\nmodel:\n  hidden_size: 32\n")
    cfg = load_config(str(f))
    assert cfg["data"]["file_path"] == "/data/x/train.h5"
    assert cfg["data"]["list"] == [os.path.expanduser("~/a"), 3] and cfg["model"]["hidden_size"] == 32
    # the W&B-export shapes the reference unwraps (src/utils.py:330-355): its own function's outputs, tests/golden/wandbcfg.json
    import json

    with open(os.path.join(os.path.dirname(__file__), "golden", "wandbcfg.json")) as fh:
        gold = json.load(fh)
    for k, v in gold["env"].items():
        monkeypatch.setenv(k, v)
    assert len(gold["cases"]) == 2
    for i, case in enumerate(gold["cases"]):
        w = tmp_path / f"w{i}.yaml"
        w.write_text(case["yaml"])
        assert load_config(str(w)) == case["expected"], i
    b = tmp_path / "bare.yaml"  # a `_wandb` section WITHOUT a value wrapper beside wrapped keys is dropped
    b.write_text("_wandb:\n  cli: x\nmodel:\n  value:\n    hidden_size: 32\nseed: 7\n")
    assert load_config(str(b)) == {"model": {"hidden_size": 32}, "seed": 7}
    e = tmp_path / "e.yaml"
    e.write_text("")
    assert load_config(str(e)) == {}


def test_model_factory_names_and_errors(tmp_path):
    """Run names carry the preprocessor tag (builder.py:45-133); errors keep the reference's types."""
    from vit_amd.builder import _build_preprocessor, get_model

    g = torch.Generator().manual_seed(3)
    q, _ = torch.linalg.qr(torch.randn(64, 64, generator=g))
    stats = {"eigvecs": q, "eigvals": torch.logspace(0, -2, 64), "mean": torch.randn(64, generator=g)}
    cases = [
        ("zca", dict(r=16, shrinkage=0.2, freeze_epochs=5), "ZCA16_fz5_s2", 64),  # low-rank ZCA still maps D -> D
        ("zca", dict(freeze_epochs=-1, bias=False), "ZCA_fzperm_nobias", 64),
        ("pca", dict(r=8), "PCA8_fz0", 8),
        ("attention", dict(r=8, freeze_epochs=2), "Attn8_scaled_fz2", 8),
        ("attention", dict(scale_by_eigvals=False), "AttnFull_fz0", 64),
    ]
    for kind, warm, tag, width in cases:
        mod, out_dim, got = _build_preprocessor(kind, warm, stats)
        assert (got, out_dim) == (tag, width), (kind, warm, got)
    with pytest.raises(ValueError, match="Unknown preprocessor type"):
        _build_preprocessor("whiten", {}, stats)
    base = {"model": dict(task_type="reg", image_size=64, patch_size=8, hidden_size=32, num_hidden_layers=1,
                          num_attention_heads=2, stride_size=8, proj_fn="SW"), "loss": {"name": "mae"}, "data": {"param": "a"}}
    with pytest.raises(ValueError, match="cov_path"):
        get_model({**base, "warmup": {"preprocessor": "zca"}})
    path = tmp_path / "cov.pt"
    torch.save(stats, path)
    bad = {**base, "model": dict(base["model"], image_size=128), "warmup": {"preprocessor": "pca", "cov_path": str(path)}}
    with pytest.raises(ValueError, match="Mismatch"):
        get_model(bad)
    cfg = {**base, "model": dict(base["model"]), "warmup": {"preprocessor": "pca", "cov_path": str(path), "r": 16}}
    m = get_model(cfg)
    assert m.name == "PCA16_fz0_ViT_p8_h32_l1_a2_s8_pSW" and cfg["model"]["image_size"] == 16 and m.config.image_size == 16
    assert get_model(base).name == "ViT_p8_h32_l1_a2_s8_pSW"


def test_checkpointer_keeps_best_and_last(tmp_path):
    from vit_amd.trainer import Checkpointer, load_checkpoint_file, model_state_from_checkpoint

    class T:  # the two attributes / one method the callback uses
        rank, current_epoch = 0, 0

        def make_checkpoint(self, module):
            return {"epoch": self.current_epoch, "global_step": 0, "state_dict": {"model.w": torch.tensor([float(self.current_epoch)])},
                    "callbacks": {}}

    ck, t = Checkpointer(str(tmp_path), "val_mae", "min"), T()
    for epoch, v in enumerate([0.5, 0.4, 0.45, 0.3]):
        t.current_epoch = epoch
        ck.after_validation(t, None, {"val_mae": v})
        assert sorted(os.listdir(tmp_path)) == sorted({os.path.basename(ck.best_path), "last.ckpt"})
    assert os.path.basename(ck.best_path) == "epoch=3-val_mae=0.3000.ckpt" and ck.best_score == 0.3
    assert float(model_state_from_checkpoint(load_checkpoint_file(ck.resolve("best")))["w"]) == 3.0
    t.current_epoch = 4
    ck.after_validation(t, None, {"val_mae": 0.9})
    assert os.path.basename(ck.best_path) == "epoch=3-val_mae=0.3000.ckpt"
    assert float(model_state_from_checkpoint(load_checkpoint_file(ck.resolve("last")))["w"]) == 4.0
    mx = Checkpointer(str(tmp_path / "m"), "val_acc", "max")
    assert mx.better(0.1) and not (mx.__setattr__("best_score", 0.5) or mx.better(0.4)) and mx.better(0.6)


def test_metrics_against_scikit_learn():
    """torchmetrics (what the reference uses, src/vit.py:3,66-73) is not in this image, so the metric classes cannot be
    pinned on it; scikit-learn is an independent implementation of the same definitions (MAE, MSE, R^2, accuracy)."""
    sk = pytest.importorskip("sklearn.metrics")
    from vit_amd.metrics import Accuracy, MeanAbsoluteError, MeanSquaredError, R2Score

    rng = np.random.default_rng(7)
    t = rng.random(257).astype(np.float32)
    p = (0.1 + 0.85 * t + 0.07 * rng.standard_normal(257)).astype(np.float32)
    mae, mse, r2 = MeanAbsoluteError(), MeanSquaredError(), R2Score()
    for a in range(0, 257, 50):  # ragged last batch
        mae(torch.from_numpy(p[a:a + 50]), torch.from_numpy(t[a:a + 50]))
        mse(torch.from_numpy(p[a:a + 50]), torch.from_numpy(t[a:a + 50]))
        r2(torch.from_numpy(p[a:a + 50]), torch.from_numpy(t[a:a + 50]))
    assert abs(float(mae.compute()) - sk.mean_absolute_error(t, p)) < 1e-6
    assert abs(float(mse.compute()) - sk.mean_squared_error(t, p)) < 1e-6
    assert abs(float(r2.compute()) - sk.r2_score(t, p)) < 1e-5
    y = rng.integers(0, 5, 300)
    logits = rng.standard_normal((300, 5)).astype(np.float32)
    acc = Accuracy()
    for a in range(0, 300, 64):
        acc(torch.from_numpy(logits[a:a + 64]), torch.from_numpy(y[a:a + 64]))
    assert abs(float(acc.compute()) - sk.accuracy_score(y, logits.argmax(1))) < 1e-7


def _evalstats():
    import json

    with open(os.path.join(os.path.dirname(__file__), "golden", "evalstats.json")) as f:
        return json.load(f)


def test_epoch_statistics_and_eval_contract_match_reference_module():
    """tests/golden/evalstats.json: the reference's own `ViTLModule.on_validation_epoch_end`, `_shared_eval_step` and
    `_normalize_task` (src/vit.py:20-27, 94-125, 157-187), called unbound on recorder objects by oracle/make_golden.py.
    The rewritten module must log the same val_bias_median / val_p90 / val_beta (one and three targets), pick the same
    tensor of a 3- / 4-tuple batch as the model input at noise_level 0 and > 0, and resolve the task the same way."""
    import json

    from vit_amd.module import ViTLModule, _normalize_task

    doc = _evalstats()
    for cfg_json, want in doc["tasks"].items():
        assert _normalize_task(json.loads(cfg_json)) == want, cfg_json
    for tag, rec in doc["stats"].items():
        pred, lab = torch.tensor(rec["pred"]), torch.tensor(rec["label"])
        logged = {}

        class Host:
            task_type = "reg"
            val_dict = {"preds": [pred[:20], pred[20:]], "labels": [lab[:20], lab[20:]]}
            _residual_stats = staticmethod(ViTLModule._residual_stats)

            def log(self, name, value, **kw):
                logged[name] = float(value)

        ViTLModule.on_validation_epoch_end(Host())
        assert logged.keys() == rec["logged"].keys(), (sorted(logged), sorted(rec["logged"]))
        for k, v in rec["logged"].items():
            assert abs(logged[k] - v) <= 2e-6 * max(1.0, abs(v)), (tag, k, logged[k], v)
    noisy, flux, err, lab4 = torch.full((4, 8), 9.0), torch.full((4, 8), 1.0), torch.ones(4, 8), torch.zeros(4)
    for key, want in doc["eval_input"].items():
        nl, n_items = float(key.split("_")[0][2:]), int(key.split("_n")[1])

        class Host2:
            noise_level = nl

        batch = (noisy, flux, err, lab4) if n_items == 4 else (flux, err, lab4)
        x, labels = ViTLModule._eval_inputs(Host2(), batch)
        assert float(x[0, 0]) == want and labels is lab4, key


def test_run_names_match_reference():
    """tests/golden/names.json: the reference's own `build_model_name` (src/models/model_utils.py:9-45) and the prefix / output
    width of its `_build_preprocessor` (src/models/builder.py:45-133) for each `warmup:` variant."""
    import json
    import types

    from vit_amd.builder import _build_preprocessor
    from vit_amd.specvit import build_model_name

    with open(os.path.join(os.path.dirname(__file__), "golden", "names.json")) as f:
        doc = json.load(f)
    for rec in doc["names"]:
        c = rec["case"]
        ns = types.SimpleNamespace(**{k: v for k, v in c.items() if k != "noise"})
        assert build_model_name(ns, "ViT", full_config={"noise": {"noise_level": c["noise"]}}) == rec["ViT"]
        assert build_model_name(ns, "ZCA_ViT") == rec["plain"]
    g = torch.Generator().manual_seed(3)
    q, _ = torch.linalg.qr(torch.randn(64, 64, generator=g))
    stats = {"eigvecs": q, "eigvals": torch.logspace(0, -2, 64), "mean": torch.randn(64, generator=g)}
    for rec in doc["warm"]:
        _, out_dim, prefix = _build_preprocessor(rec["kind"], dict(rec["warmup"]), stats)
        assert (prefix, out_dim) == (rec["prefix"], rec["out_dim"]), rec


def test_get_vit_config_matches_reference_function():
    """tests/golden/config.json: the reference's own `get_vit_config` (src/models/builder.py:200-258) on a table of configs --
    every field the path reads (incl. the hard-coded dropout / eps / activation / 4x MLP width) and the `num_labels` it writes
    back into config['model'] (regression: always derived from data.param)."""
    import copy
    import json

    from vit_amd.config import get_vit_config

    with open(os.path.join(os.path.dirname(__file__), "golden", "config.json")) as f:
        doc = json.load(f)
    for name, cfg in doc["cases"].items():
        c = copy.deepcopy(cfg)
        vc = get_vit_config(c)
        want = doc["expected"][name]
        for k, v in want["fields"].items():
            got = getattr(vc, k)
            assert got == v or (v is None and got is None), (name, k, got, v)
        assert c["model"].get("num_labels") == want["written_back_num_labels"], name


def test_freeze_schedule_matches_reference_callback():
    """tests/golden/freeze.json: the calls the reference's own `PreprocessorFreezeCallback` (src/prepca/callbacks.py) makes to
    `model.set_preprocessor_trainable` over 5 epochs, for freeze_epochs 0 / 1 / 3 / -1."""
    import json

    from vit_amd.trainer import FreezeSchedule

    with open(os.path.join(os.path.dirname(__file__), "golden", "freeze.json")) as f:
        doc = json.load(f)
    for fe, want in doc.items():
        calls, epoch = [], [-1]

        class Model:
            def set_preprocessor_trainable(self, flag):
                calls.append([epoch[0], bool(flag)])

        sch, m = FreezeSchedule(int(fe)), Model()
        sch.on_train_start(m)
        for ep in range(5):
            epoch[0] = ep
            sch.on_epoch_start(m, ep)
        assert calls == want, (fe, calls, want)


def test_load_config_matches_reference_function(tmp_path, monkeypatch):
    """tests/golden/loadcfg.json: the reference's own `load_config` (src/utils.py:311-359) on a plain experiment YAML with the
    environment fixed: $VAR / ${VAR} / ~ expanded in every string, nested lists and dicts included."""
    import json

    from vit_amd.utils import load_config

    with open(os.path.join(os.path.dirname(__file__), "golden", "loadcfg.json")) as f:
        doc = json.load(f)
    for k, v in doc["env"].items():
        monkeypatch.setenv(k, v)
    path = tmp_path / "c.yaml"
    path.write_text(doc["yaml"])
    assert load_config(str(path)) == doc["expected"]


# ---------------------------------------------------------------------------------------------- CLI data wiring (r03, f3)
def _write_split_files(tmp_path, tag="three"):
    """.npz files with the arrays of tests/golden/data.npz (inputs the reference's own RegSpecDataset was fed)."""
    import numpy as np

    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "data.npz"))
    names = ["T_eff", "log_g", "M_H"] if tag == "three" else ["log_g"]
    paths = {}
    for split, sfx in (("train", "tr"), ("val", "va"), ("test", "va")):
        p = g[f"{tag}_p_{sfx}"]
        cols = {n: (p[:, i] if p.ndim == 2 else p) for i, n in enumerate(names)}
        path = tmp_path / f"{split}.npz"
        np.savez(path, flux=g[f"flux_{sfx}"], error=g[f"err_{sfx}"], **cols)
        paths[split] = str(path)
    return g, names, paths


@pytest.mark.parametrize("tag,norm", [("three", "minmax"), ("one", "standard"), ("three", "none")])
def test_spec_datamodule_from_config_matches_reference_datasets(tmp_path, tag, norm):
    """SpecDataModule.from_config (the reference's ViTDataModule.from_config + setup, src/vit.py:29-50) over files: labels of
    every split equal what the reference's RegSpecDataset produced from the same arrays (tests/golden/data.npz) -- the
    TRAINING split's statistics re-used on val and test, the fixed-seed validation noise bit for bit, 3- / 4-tuples."""
    import numpy as np

    from vit_amd.data import SpecDataModule

    g, names, paths = _write_split_files(tmp_path, tag)
    cfg = {"model": {"task_type": "reg"}, "train": {"batch_size": 5, "debug": 1},
           "data": {"file_path": paths["train"], "val_path": paths["val"], "test_path": paths["test"], "num_samples": 100,
                    "num_test_samples": 100, "param": ",".join(names), "label_norm": norm},
           "noise": {"noise_level": 0.5}}
    dm = SpecDataModule.from_config(cfg).setup("fit")
    assert np.array_equal(dm.train.labels.numpy(), g[f"{tag}_{norm}_labels_tr"])
    assert np.array_equal(dm.val.labels.numpy(), g[f"{tag}_{norm}_labels_va"])
    assert np.array_equal(dm.val.noisy.numpy(), g[f"{tag}_{norm}_noisy_va"])
    dm.setup("test")
    assert np.array_equal(dm.test.labels.numpy(), g[f"{tag}_{norm}_labels_va"])
    b = next(iter(dm.train_dataloader()))
    assert len(b) == 3 and b[0].shape == (5, 40) and float(b[0].min()) >= 0.0  # flux clipped at zero
    v = next(iter(dm.val_dataloader()))
    assert len(v) == 4 and v[0].shape == (5, 40)
    # evaluation only: the statistics come from the training split although only setup('test') ran
    dm2 = SpecDataModule.from_config(cfg).setup("test")
    assert np.array_equal(dm2.test.labels.numpy(), g[f"{tag}_{norm}_labels_va"])
    # num_samples truncates like the reference's [:num_samples]
    cfg["data"]["num_samples"] = 8
    assert len(SpecDataModule.from_config(cfg).setup("fit").train) == 8


def test_spec_datamodule_errors_like_the_reference(tmp_path):
    from vit_amd.data import SpecDataModule

    with pytest.raises(ValueError, match="data.param"):  # spec_datasets.py:52-57
        SpecDataModule.from_config({"model": {"task_type": "reg"}, "data": {"file_path": "x"}})
    dm = SpecDataModule.from_config({"model": {"task_type": "reg"}, "data": {"file_path": str(tmp_path / "no.npz"), "param": "log_g"}})
    with pytest.raises(FileNotFoundError, match="Data file not found"):  # base.py:224-225
        dm.setup("fit")
    _, names, paths = _write_split_files(tmp_path, "one")
    dm = SpecDataModule.from_config({"model": {"task_type": "reg"}, "data": {"file_path": paths["train"], "param": "T_eff"}})
    with pytest.raises(KeyError, match="T_eff"):  # base.py:266-269
        dm.setup("fit")


def test_cli_checkpoint_resolution(tmp_path):
    """scripts/test.py: 'best' follows last.ckpt's recorded best_model_path; without it the highest NUMERIC epoch wins
    (a lexicographic sort puts epoch=9 after epoch=10)."""
    import sys

    import torch

    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from scripts.test import resolve_checkpoint

    d = tmp_path / "ck"
    d.mkdir()
    for e in (2, 9, 10):
        torch.save({"state_dict": {}}, d / f"epoch={e}-val_mae=0.1000.ckpt")
    assert resolve_checkpoint("best", str(d)).endswith("epoch=10-val_mae=0.1000.ckpt")
    torch.save({"state_dict": {}, "callbacks": {"checkpoint": {"best_model_path": str(d / "epoch=2-val_mae=0.1000.ckpt")}}},
               d / "last.ckpt")
    assert resolve_checkpoint("best", str(d)).endswith("epoch=2-val_mae=0.1000.ckpt")
    assert resolve_checkpoint("last", str(d)) == str(d / "last.ckpt")
    assert resolve_checkpoint("/some/path.ckpt", str(d)) == "/some/path.ckpt"


def test_loader_ships_only_what_the_step_reads():
    """`error` is dead weight when noise_level = 0.  `_step_reads` decides which tuple items a bound loader moves to
    the device; without a device (here) the loader yields the reference's full host tuples, in DistributedSampler order, with a
    partial last batch, and `len()` counts batches."""
    import torch

    from vit_amd.data import SpecDataset, SpecLoader, _step_reads

    g = torch.Generator().manual_seed(0)
    mk = lambda stage, noise: SpecDataset(torch.rand((37, 16), generator=g), torch.rand((37, 16), generator=g),
                                          torch.rand((37,), generator=g), task="reg", stage=stage, noise_level=noise)
    assert _step_reads(mk("train", 0.0)) == (True, False, True)          # (flux, error, labels)
    assert _step_reads(mk("train", 0.2)) == (True, True, True)           # the noise injection reads error (vit.py:86-88)
    assert _step_reads(mk("val", 0.0)) == (True, False, True)
    assert _step_reads(mk("val", 0.2)) == (True, True, False, True)      # (noisy, flux, error, labels): never error
    ds = mk("train", 0.0)
    ld = SpecLoader(ds, 8, shuffle=True, seed=3)
    batches = list(ld)
    assert len(batches) == len(ld) == 5 and [len(b[0]) for b in batches] == [8, 8, 8, 8, 5]
    assert all(len(b) == 3 and b[1] is not None for b in batches)       # host iteration: the full tuple
    seen = torch.cat([b[2] for b in batches])
    assert torch.equal(seen.sort().values, ds.labels.sort().values)
    assert len(SpecLoader(ds, 8, drop_last=True)) == 4
    with pytest.raises(ValueError, match="placement"):
        SpecLoader(ds, 8, placement="gpu")
    assert SpecLoader(ds, 8).bind("cpu").device is None                   # a CPU device binds nothing


def test_engine_refuses_unsupported_shapes_at_construction():
    """what no kernel takes is a ValueError when the model is built, not a VIT_ERR_UNSUPPORTED in the middle of a
    step.  head_dim 4 (hidden 32 / 8 heads, configs/sweep.yaml:13-18) is supported; head_dim 6 / 132, hidden % heads are not."""
    from vit_amd.config import ViTConfig
    from vit_amd.engine import ViTEngine

    ok = dict(task_type="reg", image_size=4096, patch_size=8, num_hidden_layers=1, stride_size=1, num_labels=1)
    eng = ViTEngine(ViTConfig(hidden_size=32, num_attention_heads=8, **ok))
    assert eng.cfg.head_dim == 4 and eng.cfg.seq_len == 4090
    for hidden, heads in ((48, 8), (264, 2), (40, 3)):
        with pytest.raises(ValueError):
            ViTEngine(ViTConfig(hidden_size=hidden, num_attention_heads=heads, **ok))
    big = ViTEngine(ViTConfig(hidden_size=32, num_attention_heads=2, task_type="reg", image_size=8192, patch_size=8,
                              num_hidden_layers=1, stride_size=1, num_labels=1))
    with pytest.raises(ValueError, match="4096"):
        big.set_precision("32")          # fp32 attention keeps a score row in the LDS: at most 4096 tokens
    assert big.set_precision("bf16-mixed") == "bf16"
