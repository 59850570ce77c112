"""Rank process of tests/test_ddp_gpu.py (started by vit_amd.launch.launch_ranks; NOT a test module).

Every rank builds the same C1 model on cuda:0 (the ranks SHARE the one GPU of the box; the collective runs over gloo, which
is the same Python code path as RCCL in vit_amd.ddp apart from the library underneath), takes its DistributedSampler share
of a fixed batch, and runs `Trainer.training_step` once with dropout off.  Rank r writes {flat gradient as it was handed to
the optimizer, updated parameters, loss} to <out>/rank{r}.pt."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import torch


def main(out_dir, precision, exchange, geom="C1", grad_dtype="fp32", max_bucket=0):
    from oracle import refvit  # checker-side helper: seeded weights / inputs only
    from vit_amd import ddp as ddp_mod
    from vit_amd.module import ViTLModule
    from vit_amd.trainer import Trainer, seed_everything

    seed_everything(42)
    rc = refvit.named_config(geom)  # "C3": ViT-B depth and widths -> 7.1 M-element (28 MB) layer buckets
    nb = 8
    config = {
        "model": dict(name="vit", task_type="reg", image_size=rc.image_size, patch_size=rc.patch_size,
                      hidden_size=rc.hidden_size, num_hidden_layers=rc.num_hidden_layers,
                      num_attention_heads=rc.num_attention_heads, stride_size=rc.stride_size, proj_fn="SW"),
        "train": dict(batch_size=nb, ep=1, precision=precision, ddp_exchange=exchange, ddp_grad_dtype=grad_dtype,
                      **({"ddp_max_bucket_elems": int(max_bucket)} if int(max_bucket) else {})),
        "loss": {"name": "mae"}, "opt": {"type": "AdamW", "lr": 1e-3}, "data": {"param": "log_g"}, "noise": {"noise_level": 0},
    }
    module = ViTLModule(config=config)
    # rank-dependent initial weights: the start-up broadcast must make every replica rank 0's
    module.model.load_state_dict(refvit.make_state_dict(rc, 100 + int(os.environ.get("RANK", "0"))))
    trainer = Trainer(config["train"], device=torch.device("cuda", 0), verbose=False)
    trainer._setup(module)
    module.eval()  # dropout off (masks are per-sample functions of (seed, row): a sharded batch would see other masks)
    flux, error, labels = refvit.make_inputs(rc, 8, 7)
    idx = ddp_mod.shard_indices(8, trainer.rank, trainer.world, epoch=0, shuffle=False)
    batch = tuple(t[idx].cuda() for t in (flux, error, labels))
    eng = module.model.engine
    seen = {}
    step0 = trainer.optimizer.step

    def spy_step(*a, **k):
        seen["grads"] = eng.grads.detach().cpu().clone()  # after reducer.finish(): what the optimizer consumes
        return step0(*a, **k)

    trainer.optimizer.step = spy_step
    loss = trainer.training_step(module, batch, 0)
    torch.cuda.synchronize()
    torch.save({"grads": seen["grads"], "params": eng.flat.detach().cpu().clone(), "loss": float(loss), "idx": idx,
                "n_trainable": eng.layout.n_trainable, "world": trainer.world, "backend": trainer.backend,
                "mode": trainer.reducer.mode if trainer.reducer else None,
                "calls": trainer.reducer.calls_per_step if trainer.reducer else 0,
                "bytes": trainer.reducer.bytes_per_step if trainer.reducer else 0,
                "overlap_dw": bool(eng.side_stream is not None),
                "grad_norm": float(trainer.optimizer.last_grad_norm.sqrt())},
               os.path.join(out_dir, f"rank{trainer.rank}.pt"))
    if torch.distributed.is_initialized():
        torch.distributed.barrier()
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2], sys.argv[3] if len(sys.argv) > 3 else "allreduce", *sys.argv[4:7])
