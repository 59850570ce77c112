"""Rank process of tests/test_ddp_gpu.py::test_fit_with_reserve_cus_auto_* (started by vit_amd.launch.launch_ranks or as one
VIT_DIST_SINGLE=1 process; NOT a test module).

Runs `Trainer.fit` twice in this process on the same seeded model and data: once with `train.ddp_reserve_cus: auto` (the
autotune times 4 x (1 + 3) real optimisation steps on the first batch before epoch 0 and must put every piece of training
state back), once with the value the autotune chose written into the config.  Writes both end states to <out>/rank{r}.pt.
Dropout is ON (its stream position is part of the state), the learning-rate scheduler steps per batch (one-cycle), the
train loader of run 1 is a ONE-SHOT generator in the second variant (the peeked batch must come back)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import torch


def run(reserve, one_shot, precision):
    from vit_amd.data import SpecDataset, SpecLoader
    from vit_amd.module import ViTLModule
    from vit_amd.trainer import Trainer, seed_everything

    seed_everything(42)
    config = {
        "model": dict(name="vit", task_type="reg", image_size=2048, patch_size=64, hidden_size=256, num_hidden_layers=2,
                      num_attention_heads=4, stride_size=64, proj_fn="SW"),
        "train": dict(batch_size=16, ep=1 if one_shot else 2, precision=precision, ddp_reserve_cus=reserve),
        "loss": {"name": "mae"}, "opt": {"type": "AdamW", "lr": 1e-3, "lr_sch": "onecycle"},
        "data": {"param": "log_g", "num_samples": 96}, "noise": {"noise_level": 0.2},
    }
    g = torch.Generator().manual_seed(3)
    ds = SpecDataset(torch.rand((96, 2048), generator=g), 0.1 * torch.rand((96, 2048), generator=g),
                     torch.rand((96,), generator=g), task="reg", stage="train", noise_level=0.2)
    module = ViTLModule(config=config)
    trainer = Trainer(config["train"], device=torch.device("cuda", 0), verbose=False)
    loader = SpecLoader(ds, 16, shuffle=True)
    if one_shot:
        loader.bind(trainer.device)
        loader = iter(loader)  # a generator: iter(x) is x, the autotune's peek would swallow its first batch
    hist = trainer.fit(module, loader)
    torch.cuda.synchronize()
    eng, opt = module.model.engine, trainer.optimizer
    return {"params": eng.flat.detach().cpu().clone(), "m": opt._m.detach().cpu().clone(), "v": opt._v.detach().cpu().clone(),
            "opt_step": int(opt._step), "global_step": int(trainer.global_step), "dropout_step": int(eng.step_counter),
            "lr": float(opt.param_groups[0]["lr"]), "reserve_cus": int(trainer.reserve_cus),
            "loss": float(hist[-1][f"{module.loss_name}_loss"]), "rng": torch.random.get_rng_state().clone()}


def main(out_dir, precision):
    from vit_amd import ddp as ddp_mod

    ddp_mod.init_distributed()
    out = {}
    for one_shot in (False, True):
        auto = run("auto", one_shot, precision)
        fixed = run(auto["reserve_cus"], one_shot, precision)
        out["one_shot" if one_shot else "loader"] = {"auto": auto, "fixed": fixed}
    rank = int(os.environ.get("RANK", "0"))
    torch.save(out, os.path.join(out_dir, f"rank{rank}.pt"))
    if torch.distributed.is_initialized():
        torch.distributed.barrier()
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2] if len(sys.argv) > 2 else "bf16-mixed")
