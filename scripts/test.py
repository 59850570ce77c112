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
    p.add_argument("--ckpt", type=str, default="best", help="checkpoint path, or 'best' / 'last' under $CKPT_DIR, or 'none' "
                                                            "(default 'best', as the reference's scripts/test.py:22)")
    p.add_argument("--seed", type=int, default=42)
    p.add_argument("--synthetic", type=int, default=None,
                   help="evaluate on seeded synthetic spectra instead of the config's data.test_path")
    return p.parse_args()


def resolve_checkpoint(which, ckpt_dir):
    """'last' -> <dir>/last.ckpt; 'best' -> the path ModelCheckpoint recorded in last.ckpt (`callbacks.checkpoint.
    best_model_path`: what the run itself called its best), else the `epoch=N-...ckpt` file with the highest NUMERIC epoch
    (a lexicographic sort would put epoch=9 after epoch=10); anything else is a path."""
    import re

    from vit_amd.trainer import load_checkpoint_file

    if which not in ("best", "last"):
        return which
    last = os.path.join(ckpt_dir, "last.ckpt")
    if which == "last":
        return last
    if os.path.exists(last):
        best = ((load_checkpoint_file(last).get("callbacks") or {}).get("checkpoint") or {}).get("best_model_path")
        if best and os.path.exists(best):
            return best
        if best and os.path.exists(os.path.join(ckpt_dir, os.path.basename(best))):  # the directory was moved
            return os.path.join(ckpt_dir, os.path.basename(best))
    cands = []
    for f in os.listdir(ckpt_dir) if os.path.isdir(ckpt_dir) else []:
        m = re.match(r"epoch=(\d+)-.*\.ckpt$", f)
        if m:
            cands.append((int(m.group(1)), f))
    if not cands:
        raise FileNotFoundError(f"no best checkpoint under {ckpt_dir}")
    return os.path.join(ckpt_dir, max(cands)[1])


def main(args):
    config, module, data = build(args, for_test=True)
    trainer = Trainer(config["train"])
    ckpt = args.ckpt if args.ckpt not in (None, "", "none", "None") else None
    if ckpt is not None:
        ckpt = resolve_checkpoint(ckpt, os.environ.get("CKPT_DIR", "./checkpoints"))
    print(f"[test] config={args.config} ckpt={ckpt or 'current'}")
    logs = trainer.test(module, data.test_loader(), ckpt_path=ckpt)
    if trainer.rank == 0:
        print("[test] " + " ".join(f"{k}={v:.5g}" for k, v in sorted(logs.items())))
    return logs


if __name__ == "__main__":
    main(parse_args())
