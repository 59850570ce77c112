#!/usr/bin/env python3
"""`scripts/run.py` of the reference (scripts/run.py:14-54) on the MI355X path: same arguments (-f/--config, -w, --save,
-g, --debug, --ckpt, --seed), same seeding and config plumbing, `Experiment(...).run()` replaced by the build's module +
trainer.  `--save` keeps the best-by-monitor checkpoint and `last.ckpt` under $CKPT_DIR (vit.py:386-414); `--ckpt` resumes
(vit.py:464); `-g N` starts N rank processes itself (no `-g`: every visible GPU, as the reference does, run.py:35-38).
Data: `data.file_path` / `val_path` / `test_path` of the config through `vit_amd.data.SpecDataModule` (the reference's
`ViTDataModule.from_config`, vit.py:29-50: `.npz` files here, the reference's HDF5 layout when h5py imports; training-split
label statistics re-used on val / test).  `--synthetic N` is the explicit fallback: N seeded spectra with the same batch
contract (flux, error, labels) instead of files."""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

import torch  # importing torch does not touch the GPU; the launcher parent below never calls into torch.cuda

from vit_amd.module import ViTLModule
from vit_amd.trainer import Trainer, seed_everything
from vit_amd.utils import load_config


def parse_args():
    p = argparse.ArgumentParser(description="ViT experiment runner (MI355X path)")
    p.add_argument("-f", "--config", type=str, default="configs/baseline.yaml")
    p.add_argument("-w", "--wandb", type=int, default=0, help="accepted for CLI compatibility; W&B is out of scope")
    p.add_argument("--save", action="store_true")
    p.add_argument("-g", "--gpu", type=int, default=None)
    p.add_argument("--debug", type=int, default=0)
    p.add_argument("--ckpt", type=str, default=None)
    p.add_argument("--seed", type=int, default=42)
    p.add_argument("--synthetic", type=int, default=None,
                   help="train on N seeded synthetic spectra instead of the files named in the config's data section")
    return p.parse_args()


class SyntheticSpectra:
    """Seeded (flux, error, labels) spectra: make_dummy_spectra-like absorption lines (src/utils.py:131-139) whose depth
    encodes the label, so the loss has something to learn.  Shaped like vit_amd.data.SpecDataset (the attributes
    vit_amd.data.SpecLoader reads), so that `--synthetic` runs get their batches through the same on-device input path as
    file-backed runs; not a SpecDataset itself because the continuum here sits at zero and the reference's loader clips the
    flux at zero (base.py:236)."""

    noisy = None

    def __init__(self, n, length, task, num_labels, seed, stage="train", noise_level=0.0):
        g = torch.Generator().manual_seed(seed)
        x = torch.arange(length)
        self.labels = torch.rand(n, generator=g)
        lines = sum(torch.exp(-0.5 * ((x - c) / 6.0) ** 2)[None, :] for c in (300, 600, 1200, 1600) if c < length)
        self.flux = torch.randn(n, length, generator=g) * 0.05 - (0.2 + 0.8 * self.labels[:, None]) * lines
        self.error = 0.05 * torch.rand(n, length, generator=g)
        if task == "cls":
            self.labels = (self.labels * num_labels).long().clamp_(max=num_labels - 1)
        elif num_labels > 1:
            self.labels = self.labels[:, None].repeat(1, num_labels)
        self.stage, self.noise_level = stage, float(noise_level)

    def __len__(self):
        return self.flux.shape[0]

    def __getitem__(self, j):
        return self.flux[j], self.error[j], self.labels[j]


def visible_gpus() -> int:
    """Devices this process may use, WITHOUT initialising HIP (the launcher parent must never touch the GPU: a process that
    has cannot start rank processes safely on this pool).  torch.cuda.device_count() only enumerates on this stack."""
    try:
        return int(torch.cuda.device_count())
    except Exception:  # noqa: BLE001
        return 0


class DataSource:
    """The three loaders of a run: from the config's files (SpecDataModule) or, with --synthetic N, seeded spectra."""

    def __init__(self, args, config, module):
        from vit_amd.data import SpecDataModule

        self.synthetic = getattr(args, "synthetic", None)
        self.config, self.module = config, module
        self.dm = None
        if self.synthetic is None:
            self.dm = SpecDataModule.from_config(config)
            if not self.dm.paths["train"] and not self.dm.paths["test"]:
                raise SystemExit("the config names no data files (data.file_path / val_path / test_path); "
                                 "pass --synthetic N to run on seeded synthetic spectra")
        from vit_amd.data import SpecLoader

        m, train = config["model"], config["train"]
        noise = float((config.get("noise") or {}).get("noise_level", 0.0) or 0.0)
        placement = str((config.get("data") or {}).get("placement", "auto"))

        def syn(n, seed, shuffle, training=False):
            # training batches carry `error` only when the noise injection reads it (stage 'train' + noise_level > 0)
            ds = SyntheticSpectra(n, m["image_size"], m["task_type"], module.model.config.num_labels, seed,
                                  stage="train" if training else "val", noise_level=noise if training else 0.0)
            return SpecLoader(ds, train.get("batch_size", 64), shuffle=shuffle, seed=seed, placement=placement)

        self._syn = syn

    def fit_loaders(self, debug):
        if self.dm is None:
            n_eval = max(self.config["train"].get("batch_size", 64), self.synthetic // 8)
            return self._syn(self.synthetic, 1, not debug, training=True), self._syn(n_eval, 2, False)
        self.dm.setup("fit")
        return self.dm.train_dataloader(), self.dm.val_dataloader()

    def test_loader(self):
        if self.dm is None:
            n_eval = max(self.config["train"].get("batch_size", 64), self.synthetic // 8)
            return self._syn(n_eval, 3, False)
        self.dm.setup("test")
        return self.dm.test_dataloader()


def build(args, for_test=False):
    seed_everything(args.seed)
    config = load_config(args.config)
    if args.gpu is None:
        args.gpu = visible_gpus()
    train = config.setdefault("train", {})
    train["gpus"] = args.gpu
    train["debug"] = args.debug
    train["save"] = False if for_test else bool(getattr(args, "save", False))  # pure evaluation never saves (test.py:41)
    module = ViTLModule(config=config)
    return config, module, DataSource(args, config, module)


def main(args):
    """`launch.sh run`: fit (optionally resuming from --ckpt: weights, optimizer, scheduler, epoch), then test."""
    config, module, data = build(args)
    trainer = Trainer(config["train"])
    train_loader, val_loader = data.fit_loaders(args.debug)
    hist = trainer.fit(module, train_loader, val_loader, ckpt_path=args.ckpt)
    test_logs = trainer.test(module, data.test_loader())
    if trainer.rank == 0:
        print("[test] " + " ".join(f"{k}={v:.5g}" for k, v in sorted(test_logs.items())))
        if trainer.checkpointer is not None:
            print(f"[save] best {trainer.checkpointer.best_path} ({trainer.checkpointer.monitor}="
                  f"{trainer.checkpointer.best_score}); last {trainer.checkpointer.last_path}")
    return hist


if __name__ == "__main__":
    a = parse_args()
    from vit_amd.launch import launch_ranks, under_launcher

    if not under_launcher():  # one command -> N ranks (hardware_utils.py:86-95 'ddp'); no -g: all visible GPUs (run.py:35-38)
        n = a.gpu if a.gpu is not None else visible_gpus()
        if n and n > 1:
            argv = sys.argv[1:] if a.gpu is not None else sys.argv[1:] + ["-g", str(n)]
            sys.exit(launch_ranks(n, os.path.abspath(__file__), argv))
    main(a)
