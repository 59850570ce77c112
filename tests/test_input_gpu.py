"""The input path of the training loop (reference: BaseDataModule.train_dataloader =
DataLoader(shuffle, pin_memory, persistent_workers) -> batch.to(device), src/basemodule.py:76-85, src/vit.py:83-92).

vit_amd.data.SpecLoader produces the batch ON the MI355X once a device is bound: either from a split uploaded once (a batch
= a row gather in HBM) or, for a host-resident split, through pinned staging buffers filled by a worker thread and a copy
stream that runs one to two batches ahead of the step.  Checked here: every placement yields the batches the plain host
iteration yields (same order, same rows, partial last batch, 4-tuples), tensors the step never reads stay on the host, and
`Trainer.fit` at the benchmarked geometry (C3) fed from host memory sustains the rate of the pre-staged benchmark loop."""
import time

import pytest
import torch

pytestmark = pytest.mark.gpu


def _ds(n, L, stage="train", noise=0.0, seed=0):
    from vit_amd.data import SpecDataset

    g = torch.Generator().manual_seed(seed)
    return SpecDataset(torch.rand((n, L), generator=g), 0.1 * torch.rand((n, L), generator=g), torch.rand((n,), generator=g),
                       task="reg", stage=stage, noise_level=noise)


@pytest.mark.parametrize("placement", ["device", "host", "auto"])
@pytest.mark.parametrize("stage,noise", [("train", 0.0), ("train", 0.3), ("val", 0.0), ("val", 0.3)])
def test_placements_yield_the_host_batches(dev, placement, stage, noise):
    from vit_amd.data import SpecLoader

    ds = _ds(203, 64, stage, noise)
    for epoch in (0, 1):
        ref = SpecLoader(ds, 32, shuffle=(stage == "train"), seed=5)
        got = SpecLoader(ds, 32, shuffle=(stage == "train"), seed=5, placement=placement).bind(dev)
        ref.set_epoch(epoch)
        got.set_epoch(epoch)
        n = 0
        for a, b in zip(ref, got):
            assert len(a) == len(b) == (4 if (stage == "val" and noise > 0) else 3)
            for i, (x, y) in enumerate(zip(a, b)):
                is_error = i == len(a) - 2
                if is_error and not (stage == "train" and noise > 0):
                    assert y is None  # never read by the step: not moved (dead weight when noise_level = 0)
                    continue
                assert y.is_cuda and torch.equal(x, y.cpu()), (epoch, n, i)
            n += 1
        assert n == 7 == len(got)
    assert got.resolved == ("host" if placement == "host" else "device")


def test_host_staging_runs_ahead_across_epochs(dev):
    """When an epoch has been consumed, the first batches of epoch + 1 are staged at once (the buffers and their events live
    across epochs).  The next iteration takes them if it asks for that epoch and discards them otherwise; either way the
    batches are the host iteration's."""
    from vit_amd.data import SpecLoader

    ds = _ds(10 * 16 + 5, 256, seed=9)
    ld = SpecLoader(ds, 16, shuffle=True, seed=2, placement="host").bind(dev)
    for epoch in (0, 1, 5, 5, 6):  # 0 -> 1 and 5 -> 6 use what was staged ahead; 1 -> 5 and 5 -> 5 discard it
        ld.set_epoch(epoch)
        ref = SpecLoader(ds, 16, shuffle=True, seed=2)
        ref.set_epoch(epoch)
        got = [(b[0].cpu(), b[2].cpu()) for b in ld]
        assert len(got) == 11
        for (f, l), r in zip(got, ref):
            assert torch.equal(f, r[0]) and torch.equal(l, r[2]), epoch
        assert (ld._ahead is not None) and ld._ahead[0][0] == epoch + 1
    ld.close()
    assert ld._ahead is None
    # an epoch abandoned half way stages nothing ahead
    it = iter(ld)
    next(it)
    it.close()
    assert ld._ahead is None


def test_host_staging_many_short_epochs(dev):
    """Sixty epochs of three batches each (epoch boundaries are where the worker thread, the slot events and the batches staged
    ahead change hands), with epochs abandoned after the first batch in between: contents as the host iteration gives them,
    and no thread left behind."""
    import threading

    from vit_amd.data import SpecLoader

    ds = _ds(3 * 8, 512, seed=21)
    ld = SpecLoader(ds, 8, shuffle=True, seed=4, placement="host").bind(dev)
    before = threading.active_count()
    for epoch in range(60):
        ld.set_epoch(epoch)
        ref = SpecLoader(ds, 8, shuffle=True, seed=4)
        ref.set_epoch(epoch)
        if epoch % 7 == 3:  # abandoned epoch: nothing is staged ahead, the next epoch starts cold
            it = iter(ld)
            first = next(it)
            assert torch.equal(first[0].cpu(), next(iter(ref))[0])
            it.close()
            continue
        for b, r in zip(ld, ref):
            assert torch.equal(b[0].cpu(), r[0]) and torch.equal(b[2].cpu(), r[2]), epoch
    ld.close()
    torch.cuda.synchronize()
    assert threading.active_count() <= before + 1


def test_host_staging_keeps_batches_valid_while_running_ahead(dev):
    """The stager refills a slot only after the consumer let go of it: hold each batch across the next two fetches (what the
    step does: labels are read again in backward) and compare afterwards."""
    from vit_amd.data import SpecLoader

    ds = _ds(40 * 16, 4096, seed=3)
    ld = SpecLoader(ds, 16, shuffle=True, seed=1, placement="host", prefetch=2).bind(dev)
    ref = list(SpecLoader(ds, 16, shuffle=True, seed=1))
    held = []
    for k, b in enumerate(ld):
        # some device work between fetches so that copies really run ahead of the consumer
        junk = torch.randn((1024, 1024), device=dev)
        (junk @ junk).sum()
        held.append((k, b[0].clone(), b[2].clone(), b[0], b[2]))
        if len(held) >= 2:
            kk, f0, l0, fv, lv = held.pop(0)
            # the VIEW handed out one fetch ago still holds its batch (its slot is only released at the next fetch)
            assert torch.equal(f0.cpu(), ref[kk][0]) and torch.equal(l0.cpu(), ref[kk][2])
    assert k == 39


def test_fit_from_host_memory_sustains_the_prestaged_rate(dev):
    """`Trainer.fit` at C3 (ViT-B/16 224^2 restated, B = 256, bf16-mixed, dropout on) for 72 steps from a
    HOST-resident split through the pinned / copy-stream staging must reach >= 0.97 x the images/s of the benchmark's loop
    over one batch pre-staged in HBM (same process, same kernels); the device-resident placement likewise.  Measured: 0.98-0.99 x
    from host memory (0.978 with 12-step epochs: what is left is paid once per epoch), 1.000 x from a resident split."""
    from vit_amd.data import SpecLoader
    from vit_amd.module import ViTLModule
    from vit_amd.trainer import Trainer, seed_everything

    L, B, n_batches, epochs = 50176, 256, 24, 3
    config = {"model": dict(name="vit", task_type="reg", image_size=L, patch_size=256, hidden_size=768, num_hidden_layers=12,
                            num_attention_heads=12, stride_size=256, proj_fn="SW"),
              "train": dict(batch_size=B, ep=epochs, precision="bf16-mixed"), "loss": {"name": "mae"},
              "opt": {"type": "AdamW", "lr": 1e-4}, "data": {"param": "log_g"}, "noise": {"noise_level": 0}}
    ds = _ds(B * n_batches, L, seed=11)  # 1.2 GB of flux (+ as much error, which must stay where it is)
    rates = {}
    for placement in ("host", "device"):
        seed_everything(42)
        module = ViTLModule(config=config)
        trainer = Trainer(config["train"], device=dev, verbose=False)
        loader = SpecLoader(ds, B, shuffle=True, placement=placement)
        hist = trainer.fit(module, loader)
        assert trainer.global_step == epochs * n_batches and loader.resolved == placement
        # epoch 0 allocates the arena and warms the kernels up; epochs 1..2 are 48 steady steps (each epoch ends with one
        # device sync for its logs: part of what fit() costs)
        dt = sum(h["epoch_time_s"] for h in hist[1:])
        rates[placement] = (epochs - 1) * n_batches * B / dt
        if placement == "host":
            # the pre-staged loop of bench.py on the same model: one resident batch, 48 steps
            batch = tuple(t[:B].to(dev) if t is not None else None for t in (ds.flux, None, ds.labels))
            module.train()
            for i in range(3):
                trainer.training_step(module, batch, i)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for i in range((epochs - 1) * n_batches):
                trainer.training_step(module, batch, i)
            torch.cuda.synchronize()
            rates["prestaged"] = (epochs - 1) * n_batches * B / (time.perf_counter() - t0)
        del module, trainer
    print(f"[fit C3] images/s: pre-staged loop {rates['prestaged']:.0f}, fit() from host memory {rates['host']:.0f} "
          f"({rates['host'] / rates['prestaged']:.3f} x), fit() from a device-resident split {rates['device']:.0f} "
          f"({rates['device'] / rates['prestaged']:.3f} x)")
    assert rates["host"] >= 0.97 * rates["prestaged"], rates
    assert rates["device"] >= 0.97 * rates["prestaged"], rates


def test_training_noise_needs_the_error_tensor(dev):
    """noise.noise_level > 0 in the module with a dataset built at noise 0: the loader did not ship `error`; the step says so
    instead of failing inside a kernel call, and ship_error=True fixes it."""
    from vit_amd.data import SpecLoader
    from vit_amd.module import ViTLModule

    cfg = {"model": dict(name="vit", task_type="reg", image_size=512, patch_size=32, hidden_size=32, num_hidden_layers=1,
                         num_attention_heads=2, stride_size=32, proj_fn="SW"),
           "train": dict(batch_size=8, ep=1), "loss": {"name": "mae"}, "opt": {"type": "AdamW", "lr": 1e-3},
           "data": {"param": "log_g"}, "noise": {"noise_level": 0.5}}
    m = ViTLModule(config=cfg).to(dev)
    m.train()
    ds = _ds(16, 512)
    b = next(iter(SpecLoader(ds, 8, placement="device").bind(dev)))
    with pytest.raises(ValueError, match="error"):
        m.training_step(b, 0)
    b = next(iter(SpecLoader(ds, 8, placement="device", ship_error=True).bind(dev)))
    assert torch.isfinite(m.training_step(b, 0))
