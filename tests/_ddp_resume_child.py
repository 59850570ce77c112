"""Rank process of tests/test_ddp_gpu.py::test_zero1_checkpoint_resume (started by vit_amd.launch.launch_ranks; NOT a test
module).  Two ranks share GPU 0 over gloo.  Each rank runs (a) an uninterrupted 3-epoch fit and (b) a 2-epoch fit with
`train.save`, then a fresh module / trainer resumed from last.ckpt for the third epoch, both under the exchange named on the
command line, and writes the final parameters and AdamW moments of both runs to <out>/rank{r}.pt."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import torch


def main(out_dir, exchange, placement="auto"):
    from vit_amd.data import SpecDataset, SpecLoader
    from vit_amd.module import ViTLModule
    from vit_amd.trainer import Trainer, seed_everything

    os.environ["CKPT_DIR"] = os.path.join(out_dir, "ck")

    def config(ep, save):
        return {
            "model": dict(name="vit", task_type="reg", image_size=4096, patch_size=32, hidden_size=32, num_hidden_layers=3,
                          num_attention_heads=2, stride_size=32, proj_fn="SW"),
            "train": dict(batch_size=8, ep=ep, precision="32", ddp_exchange=exchange, save=save),
            "loss": {"name": "mae"}, "opt": {"type": "AdamW", "lr": 1e-3}, "data": {"param": "log_g", "val_path": "x"},
            "noise": {"noise_level": 0},
        }

    g = torch.Generator().manual_seed(3)
    flux, err, lab = torch.randn(48, 4096, generator=g).abs(), 0.1 * torch.rand(48, 4096, generator=g), torch.rand(48, generator=g)
    # placement 'host': every rank stages its shard's batches through its own pinned buffers / copy stream / worker thread
    train = SpecLoader(SpecDataset(flux[:32], err[:32], lab[:32]), 8, shuffle=True, placement=placement)
    val = SpecLoader(SpecDataset(flux[32:], err[32:], lab[32:], stage="val"), 8, placement=placement)

    def run(ep, save, ckpt=None):
        seed_everything(42)
        cfg = config(ep, save)
        module = ViTLModule(config=cfg)
        trainer = Trainer(cfg["train"], device=torch.device("cuda", 0), verbose=False)
        trainer.fit(module, train, val, ckpt_path=ckpt)
        torch.cuda.synchronize()
        opt = trainer.optimizer
        opt.gather_sharded_state()
        n = module.model.engine.layout.n_trainable
        return {"params": module.model.engine.flat.detach().cpu().clone(), "m": opt._m[:n].detach().cpu().clone(),
                "v": opt._v[:n].detach().cpu().clone(), "step": int(opt._step), "mode": trainer.reducer.mode,
                "last": trainer.checkpointer.last_path if trainer.checkpointer else None}

    full = run(3, False)
    part = run(2, True)
    torch.distributed.barrier()  # rank 0 wrote last.ckpt
    resumed = run(3, False, ckpt=os.path.join(os.environ["CKPT_DIR"], "last.ckpt"))
    rank = torch.distributed.get_rank()
    torch.save({"full": full, "part": part, "resumed": resumed}, os.path.join(out_dir, f"rank{rank}.pt"))
    torch.distributed.barrier()
    torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2], *sys.argv[3:4])
