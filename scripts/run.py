#!/usr/bin/env python3
"""`scripts/run.py` of the reference (scripts/run.py:14-54) on the MI355X path: same arguments (-f/--config, -w, --save,
-g, --debug, --ckpt, --seed), same seeding and config plumbing, `Experiment(...).run()` replaced by the build's module +
trainer.  `--save` keeps the best-by-monitor checkpoint and `last.ckpt` under $CKPT_DIR (vit.py:386-414); `--ckpt` resumes
(vit.py:464); `-g N` starts N rank processes itself.  The reference's HDF5 datasets are out of scope (SURVEY.md section 2
#10), so data comes from `--synthetic N` seeded spectra with the reference's batch contract (flux, error, labels)."""
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
    p.add_argument("--synthetic", type=int, default=4096, help="number of synthetic training spectra")
    return p.parse_args()


class SyntheticSpectra:
    """Seeded (flux, error, labels) batches: make_dummy_spectra-like absorption lines (src/utils.py:131-139) whose depth
    encodes the label, so the loss has something to learn."""

    def __init__(self, n, length, batch_size, task, num_labels, seed, shuffle):
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
        self.bs, self.shuffle, self.epoch, self.seed = batch_size, shuffle, 0, seed

    def set_epoch(self, e):
        self.epoch = e

    def __iter__(self):
        from vit_amd.ddp import shard_indices
        import torch.distributed as dist

        rank = dist.get_rank() if dist.is_initialized() else 0
        world = dist.get_world_size() if dist.is_initialized() else 1
        idx = shard_indices(self.flux.shape[0], rank, world, self.epoch, self.shuffle, self.seed)
        for i in range(0, len(idx), self.bs):
            j = idx[i:i + self.bs]
            yield self.flux[j], self.error[j], self.labels[j]


def build(args, for_test=False):
    seed_everything(args.seed)
    config = load_config(args.config)
    if args.gpu is None:
        args.gpu = torch.cuda.device_count() if torch.cuda.is_available() else 0
    train = config.setdefault("train", {})
    train["gpus"] = args.gpu
    train["debug"] = args.debug
    train["save"] = False if for_test else bool(getattr(args, "save", False))  # pure evaluation never saves (test.py:41)
    module = ViTLModule(config=config)
    m = config["model"]
    bs = train.get("batch_size", 64)

    def spectra(n, seed, shuffle):
        return SyntheticSpectra(n, m["image_size"], bs, m["task_type"], module.model.config.num_labels, seed, shuffle)

    return config, module, spectra


def main(args):
    """`launch.sh run`: fit (optionally resuming from --ckpt: weights, optimizer, scheduler, epoch), then test."""
    config, module, spectra = build(args)
    n_eval = max(config["train"].get("batch_size", 64), args.synthetic // 8)
    trainer = Trainer(config["train"])
    hist = trainer.fit(module, spectra(args.synthetic, 1, not args.debug), spectra(n_eval, 2, False), ckpt_path=args.ckpt)
    test_logs = trainer.test(module, spectra(n_eval, 3, False))
    if trainer.rank == 0:
        print("[test] " + " ".join(f"{k}={v:.5g}" for k, v in sorted(test_logs.items())))
        if trainer.checkpointer is not None:
            print(f"[save] best {trainer.checkpointer.best_path} ({trainer.checkpointer.monitor}="
                  f"{trainer.checkpointer.best_score}); last {trainer.checkpointer.last_path}")
    return hist


if __name__ == "__main__":
    a = parse_args()
    from vit_amd.launch import launch_ranks, under_launcher

    if a.gpu and a.gpu > 1 and not under_launcher():  # one command -> N ranks (hardware_utils.py:86-95 'ddp')
        sys.exit(launch_ranks(a.gpu, os.path.abspath(__file__), sys.argv[1:]))
    main(a)
