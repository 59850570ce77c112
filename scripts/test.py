#!/usr/bin/env python3
"""`scripts/test.py` of the reference (scripts/test.py:15-48) on the MI355X path: evaluation ONLY.  Builds the module from
the config, loads `--ckpt` (a Lightning-layout checkpoint or a bare state_dict; 'none' = the freshly initialised weights),
applies `train.precision`, runs the test loop once and prints the metrics.  Nothing is trained and nothing is saved."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

import argparse

from scripts.run import build  # same config plumbing and synthetic data as `run`
from vit_amd.trainer import Trainer


def parse_args():
    p = argparse.ArgumentParser(description="ViT evaluation runner (MI355X path)")
    p.add_argument("-f", "--config", type=str, default="configs/baseline.yaml")
    p.add_argument("-w", "--wandb", type=int, default=0, help="accepted for CLI compatibility; W&B is out of scope")
    p.add_argument("-g", "--gpu", type=int, default=None)
    p.add_argument("--debug", type=int, default=0)
    p.add_argument("--ckpt", type=str, default="last", help="checkpoint path, or 'best' / 'last' under $CKPT_DIR, or 'none'")
    p.add_argument("--seed", type=int, default=42)
    p.add_argument("--synthetic", type=int, default=4096)
    return p.parse_args()


def main(args):
    config, module, spectra = build(args, for_test=True)
    trainer = Trainer(config["train"])
    ckpt = args.ckpt if args.ckpt not in (None, "", "none", "None") else None
    if ckpt in ("best", "last"):
        d = os.environ.get("CKPT_DIR", "./checkpoints")
        if ckpt == "last":
            ckpt = os.path.join(d, "last.ckpt")
        else:  # the single best-by-monitor file ModelCheckpoint(save_top_k=1) leaves next to last.ckpt
            cands = sorted(f for f in os.listdir(d) if f.startswith("epoch=") and f.endswith(".ckpt"))
            if not cands:
                raise FileNotFoundError(f"no best checkpoint under {d}")
            ckpt = os.path.join(d, cands[-1])
    print(f"[test] config={args.config} ckpt={ckpt or 'current'}")
    n_eval = max(config["train"].get("batch_size", 64), args.synthetic // 8)
    logs = trainer.test(module, spectra(n_eval, 3, False), ckpt_path=ckpt)
    if trainer.rank == 0:
        print("[test] " + " ".join(f"{k}={v:.5g}" for k, v in sorted(logs.items())))
    return logs


if __name__ == "__main__":
    main(parse_args())
