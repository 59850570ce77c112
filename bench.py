#!/usr/bin/env python3
"""Benchmark of the MI355X ViT training hot path (BASELINE.json metric: images/sec of a ViT-B/16 224^2 bf16 train step).

    python bench.py --gpus N --steps K --warmup W

N > 1 works both ways: under `python -m torch.distributed.run --nproc-per-node N bench.py --gpus N ...` (RANK / LOCAL_RANK /
WORLD_SIZE / MASTER_* in the environment: this process IS a rank), or from a plain shell, where this process only
launches N fresh rank processes of itself (it never touches the GPU), waits for them and exits with their status.

A "step" is one full optimisation step of the reference's `training_step` path on one synthetic batch that is already
resident in HBM: forward (dropout on) -> backward -> [RCCL gradient all-reduce, overlapped] -> global-norm clip 0.5 ->
AdamW.  Workload = SURVEY.md section 8 config C3: flux [256, 50176] f32 per GPU (a 224x224x1 image flattened),
patch 256 (=16^2) -> 196 tokens + CLS, hidden 768, 12 heads, 12 layers, MLP 3072, 85.8 M parameters, regression head,
MSE loss (what baseline.yaml's loss.name 'mae' resolves to), AdamW lr 1e-3 wd 0.  Weak scaling: 256 images per GPU
(default; `--global-batch 256` fixes the total instead: strong scaling, 256 / N images per GPU).

Rank 0 prints ONE JSON line.  Besides the driver's fields it carries
  roofline     -- the dominant kernel (by summed time) of the step: algorithmic FLOPs per launch / mean launch duration,
                  measured live with HIP events on the launch stream during the timed steps, against the dense bf16
                  MFMA peak (2.5 PFLOP/s, MI355X_MICROARCH.md); `step_frac` is the whole-step model-FLOPs utilisation.
  cpu_baseline -- the CPU oracle (oracle/refvit.py, plain fp32 torch: a "port") timed on this host's cores on a bounded
                  sample of the same workload (rank 0, N=1 only).
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PEAK_BF16_DENSE = 2.5e15  # FLOP/s, MI355X dense bf16 MFMA (MI355X_MICROARCH.md, chip-level parameters)
# What a registers-only MFMA loop SUSTAINS on random operands inside this chip's 1400 W cap (tools/micro/mfma_power.hip,
# profiles/r05_b_mfma_power.txt: 16x16x32 bf16, 2.17 GHz, 1320 W; constant operands reach 2.35 PFLOP/s at 2.4 GHz).  Reported
# beside the contract's `peak`, never instead of it.
SUSTAINED_BF16_DENSE_RANDOM = 2.03e15

WORKLOADS = {
    # name: (image_size, patch, hidden, layers, heads, per-GPU batch)
    "vit_b16_224": (50176, 256, 768, 12, 12, 256),
    "vit_l16_384": (147456, 256, 1024, 24, 16, 32),
    "vit_tiny16_32": (1024, 256, 192, 12, 3, 64),
    "baseline_yaml": (4096, 32, 32, 3, 2, 64),
}


def log(msg):
    print(f"[bench] {msg}", file=sys.stderr, flush=True)


def train_flop_per_image(L, P, D, layers, F):
    """2*3*[N*P*D + layers*(3TD^2 + 2T^2 D + TD^2 + 2TDF) + D^2 + D]."""
    N = L // P
    T = N + 1
    macs = N * P * D + layers * (3 * T * D * D + 2 * T * T * D + T * D * D + 2 * T * D * F) + D * D + D
    return 6.0 * macs


def launch_check(n_expected: int):
    """CPU-only rehearsal of the rank plumbing of `bench.py --gpus N` (tests/test_launch_cpu.py)."""
    import torch
    import torch.distributed as dist

    from vit_amd import ddp as ddp_mod

    rank, local, world = ddp_mod.init_distributed(backend="gloo")
    t = torch.tensor([float(rank)], dtype=torch.float64)
    if world > 1:
        dist.all_reduce(t)
    if rank == 0:
        print(json.dumps({"launch_check": True, "n_gpus": n_expected, "world": world, "rank_sum": float(t),
                          "backend": dist.get_backend() if world > 1 else None}), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    if world != n_expected:
        raise SystemExit(f"--gpus {n_expected} but WORLD_SIZE={world}")


def comm_probe(args, trainer, module, eng, red, batch, barrier, dev, world, median_ms):
    """N > 1 (or the single-rank rehearsal), after `value` is taken: what the gradient exchange costs and which of the
    N > 1 knobs pays, from ONE invocation (no scaling curve was ever measured for this path, so the first multi-GPU run has to
    answer these by itself).  Every figure is the median per-step hipEvent time of `probe_steps` steps, MAX over ranks:
      * the step with the exchange switched off (replicas drift apart, which no longer matters) -> exposed exchange time;
      * the bare all-reduce of the step's buckets with nothing else on the GPU -> algorithm / bus bandwidth over xGMI;
      * `reserve_cus` in {0, 8, 16, 32}: the ping-pong GEMMs and the pair-pipelined attention backward are persistent grids of
        one workgroup per CU, so an overlapping RCCL kernel has no CU to run on unless some are left free -- step time with
        and without the exchange at each setting;
      * the `zero1` schedule (reduce-scatter, sharded AdamW, parameter all-gather) at reserve 0 and at the best reserve;
      * strong scaling: a fixed global batch of 256 images (256 / N per GPU), SURVEY.md section 8d."""
    import torch

    from vit_amd import _cabi
    from vit_amd import ddp as ddp_mod

    dist = torch.distributed
    n_probe = max(2, min(args.steps, 10))

    def run(n, b=batch):
        barrier()
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(n + 1)]
        ev[0].record()
        for i in range(n):
            trainer.training_step(module, b, i)
            ev[i + 1].record()
        barrier()
        t = torch.tensor([ev[i].elapsed_time(ev[i + 1]) for i in range(n)], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.median())

    def set_exchange(r):
        trainer.reducer = r
        eng.grad_ready_cb = r.bucket_ready if r is not None else None
        if hasattr(trainer.optimizer, "attach_reducer"):
            trainer.optimizer.attach_reducer(r)

    out = {"probe_steps": n_probe}
    set_exchange(None)
    off0 = run(n_probe)
    set_exchange(red)
    out["ms_per_step_exchange_off"] = round(off0, 3)
    out["exposed_exchange_ms"] = round(median_ms - off0, 3)
    # the bare collective
    barrier()
    t3 = time.perf_counter()
    for _ in range(n_probe):
        for lo, hi in eng.layout.buckets():
            red.bucket_ready(lo, hi)
        red.finish()
    barrier()
    t = torch.tensor([time.perf_counter() - t3], dtype=torch.float64, device=dev)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    dt_ar = float(t[0])
    alg = red.bytes_per_step / (dt_ar / n_probe) / 1e9
    out.update({"bare_allreduce_ms": round(dt_ar / n_probe * 1e3, 3), "algbw_GBps": round(alg, 1),
                "busbw_GBps": round(alg * 2 * (world - 1) / world, 1)})
    # reserve_cus sweep, all-reduce schedule
    sweep, best = [], (median_ms, args.reserve_cus)
    for rc in (0, 8, 16, 32):
        trainer.set_reserve_cus(module, rc)
        on = run(n_probe)
        set_exchange(None)
        off = run(n_probe)
        set_exchange(red)
        sweep.append({"reserve_cus": rc, "ms_per_step": round(on, 3), "ms_per_step_exchange_off": round(off, 3),
                      "exposed_exchange_ms": round(on - off, 3)})
        if on < best[0]:
            best = (on, rc)
    off_by_rc = {x["reserve_cus"]: x["ms_per_step_exchange_off"] for x in sweep}
    out["reserve_cus_sweep"] = sweep
    out["best_reserve_cus"] = best[1]
    # the sharded schedule
    try:
        z = ddp_mod.make_reducer("zero1", eng, grad_dtype="fp32", max_bucket_elems=getattr(trainer, "max_bucket_elems", 64 << 20))
        zs = []
        for rc in sorted({0, best[1]}):
            trainer.set_reserve_cus(module, rc)
            set_exchange(z)
            on = run(n_probe)
            zs.append({"reserve_cus": rc, "ms_per_step": round(on, 3),
                       "exposed_exchange_ms": round(on - off_by_rc.get(rc, off0), 3)})
        out["zero1"] = zs
    except Exception as e:  # noqa: BLE001
        out["zero1_error"] = f"{type(e).__name__}: {e}"
    set_exchange(red)
    trainer.set_reserve_cus(module, args.reserve_cus)
    # strong scaling at a global batch of 256
    B = batch[0].shape[0]
    if not args.global_batch and 256 % world == 0 and 256 // world != B and 256 // world <= B:
        bs = 256 // world
        small = tuple(t[:bs].contiguous() for t in batch)
        for i in range(2):
            trainer.training_step(module, small, i)  # the arena of the new batch size
        ms = run(n_probe, small)
        out["strong_scaling"] = {"global_batch": 256, "per_gpu_batch": bs, "ms_per_step": round(ms, 3),
                                 "images_per_s": round(256 / (ms * 1e-3), 2)}
    return out


def build_run(workload, B, precision, dev, rank, use_graph=False, quiet=True):
    """Module + trainer (optimizer, exchange) + one synthetic batch resident in HBM for a named workload (SURVEY 8d inputs:
    generator seed 1234 + rank, flux ~ N(0,1), error = 0.1 |N(0,1)|, labels ~ U[0,1); model seed 42, scripts/run.py:22)."""
    import contextlib

    import torch

    from vit_amd.module import ViTLModule
    from vit_amd.trainer import Trainer, seed_everything

    L, P, D, layers, heads, _ = WORKLOADS[workload]
    config = {
        "model": dict(name="vit", task_type="reg", image_size=L, patch_size=P, hidden_size=D, num_hidden_layers=layers,
                      num_attention_heads=heads, stride_size=P, proj_fn="SW"),
        "train": dict(batch_size=B, ep=1, precision=precision, hip_graph=use_graph),
        "loss": {"name": "mae"},
        "opt": {"type": "AdamW", "lr": 1e-3},
        "data": {"param": "log_g"},
        "noise": {"noise_level": 0},
    }
    seed_everything(42)
    with contextlib.redirect_stdout(sys.stderr):  # the builders print like the reference's do; stdout carries ONE JSON line
        module = ViTLModule(config=config)
    trainer = Trainer(config["train"], device=dev, verbose=False)
    trainer._setup(module)
    module.train()
    g = torch.Generator(device="cpu").manual_seed(1234 + rank)
    flux = torch.randn((B, L), generator=g)
    error = 0.1 * torch.randn((B, L), generator=g).abs()
    labels = torch.rand((B,), generator=g)
    host = (flux, error, labels)
    batch = tuple(t.to(dev) for t in host)
    return module, trainer, batch, host


def median(xs):
    srt = sorted(xs)
    return srt[len(srt) // 2] if len(srt) % 2 else 0.5 * (srt[len(srt) // 2 - 1] + srt[len(srt) // 2])


def time_steps(trainer, module, batches, n, warmup):
    """Median per-step hipEvent time (ms) of `n` optimisation steps after `warmup` untimed ones; `batches` is an iterator."""
    import torch

    for i in range(warmup):
        trainer.training_step(module, next(batches), i)
    torch.cuda.synchronize()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(n + 1)]
    ev[0].record()
    for i in range(n):
        trainer.training_step(module, next(batches), i)
        ev[i + 1].record()
    torch.cuda.synchronize()
    return median([ev[i].elapsed_time(ev[i + 1]) for i in range(n)])


def secondary_workloads(dev, rank, precision):
    """BASELINE.json's other single-GPU configurations, timed in the same process after the headline so that they are
    driver-observed too: C5 = ViT-L/16 384^2 (B 32) and C2 = ViT-Tiny/16 32x32 (B 64; launch-bound: a
    correctness configuration).  Median per-step hipEvent time of the same full optimisation step."""
    import itertools

    out = {}
    for name, steps, warmup in (("vit_l16_384", 16, 4), ("vit_tiny16_32", 30, 10)):
        L, P, D, layers, heads, B = WORKLOADS[name]
        t0 = time.perf_counter()
        module, trainer, batch, _ = build_run(name, B, precision, dev, rank)
        ms = time_steps(trainer, module, itertools.repeat(batch), steps, warmup)
        flop_img = train_flop_per_image(L, P, D, layers, 4 * D)
        out[name] = {"value": round(B / (ms * 1e-3), 2), "unit": "images/s", "ms_per_step": round(ms, 3), "steps": steps,
                     "warmup": warmup, "batch": B, "train_gflop_per_image": round(flop_img / 1e9, 3),
                     "step_frac": round(B / (ms * 1e-3) * flop_img / PEAK_BF16_DENSE, 4),
                     "wall_s": round(time.perf_counter() - t0, 1)}
        log(f"secondary {name}: {out[name]}")
        del module, trainer, batch
    return out


def input_pipeline_probe(trainer, module, host, dev, B, steps, device_ms):
    """What `Trainer.fit` pays for its inputs at this workload (SURVEY 8a16 / 8d 'variant produced on host + H2D'): the same
    optimisation steps fed by vit_amd.data.SpecLoader from a HOST-resident split of 8 x B spectra -- (a) placement 'host': a
    worker thread gathers each shuffled batch into pinned memory and a copy stream moves it one to two batches ahead; (b)
    placement 'device': the split uploaded once, a batch = a row gather in HBM.  Reference: DataLoader(shuffle, pin_memory,
    persistent_workers) -> batch.to(device), src/basemodule.py:76-85.  `value` (inputs resident) is not touched by this."""
    import torch

    from vit_amd.data import SpecDataset, SpecLoader

    flux, error, labels = host
    reps = 8
    ds = SpecDataset(torch.cat([flux.roll(i, 0) for i in range(reps)]).abs_(), torch.cat([error] * reps),
                     torch.cat([labels] * reps), task="reg", stage="train")
    out = {"split_rows": len(ds), "split_mbytes": round(ds.flux.numel() * 4 / 1e6, 1), "steps": steps,
           "device_resident_batch_ms_per_step": round(device_ms, 3)}

    def stream(loader):
        e = 0
        while True:
            loader.set_epoch(e)
            yield from loader
            e += 1

    for mode in ("host", "device"):
        loader = SpecLoader(ds, B, shuffle=True, placement=mode).bind(dev)
        it = stream(loader)
        ms = time_steps(trainer, module, it, steps, 3)
        it.close()
        out[f"{mode}_ms_per_step"] = round(ms, 3)
        out[f"{mode}_images_per_s"] = round(B / (ms * 1e-3), 2)
        out[f"{mode}_vs_resident_batch"] = round(device_ms / ms, 4)
    out["h2d_bytes_per_step_host"] = int(B * (flux.shape[1] + 1) * 4)
    out["note"] = "error tensor not shipped (noise_level 0: SURVEY 8a16); the reference ships 2*B*L*4 bytes per step"
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=40)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--workload", default="vit_b16_224", choices=sorted(WORKLOADS))
    ap.add_argument("--batch", type=int, default=0, help="per-GPU batch (default: the workload's)")
    ap.add_argument("--global-batch", type=int, default=0,
                    help="STRONG scaling: this many images per step over ALL GPUs (per-GPU batch = global / N; SURVEY.md section "
                         "8d: 'also report fixed-global-256 strong scaling'); the JSON line then says \"scaling\": \"strong\"")
    ap.add_argument("--precision", default="bf16-mixed", choices=["bf16-mixed", "32"],
                    help="bf16-mixed = the BASELINE.json metric; 32 = the fp32-class mode (x3 GEMMs), for the record only")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-kernel-timing", action="store_true", help="skip the per-GEMM HIP-event brackets")
    ap.add_argument("--no-overlap", action="store_true", help="weight-gradient GEMMs on the main stream (no second HIP stream)")
    ap.add_argument("--overlap", action="store_true",
                    help="weight-gradient GEMMs on a second HIP stream whatever the shape (default: the engine decides from the tile "
                         "count -- one stream at the benchmarked B = 256, two where the GEMM grids leave >= 30 %% of the CUs idle)")
    ap.add_argument("--graph", action="store_true",
                    help="N = 1: replay the step as one captured hipGraph (vit_amd/graph.py) instead of launching its kernels. "
                         "Measured equal to eager launches within noise (the host runs ahead of the GPU either way) and 0.7 ms "
                         "slower when combined with the second stream, so it is opt-in and switches the second stream off")
    ap.add_argument("--no-graph", action="store_true", help="(default now; kept so older command lines still parse)")
    ap.add_argument("--reserve-cus", type=int, default=-1,
                    help="size the one-workgroup-per-CU kernels for this many fewer CUs (room for the collective's kernels when "
                         "the gradient exchange overlaps the backward; 0 = all CUs).  Default -1: 0 on one GPU; with N > 1 the "
                         "value is chosen BEFORE the warm-up by timing three steps at 0 / 8 / 16 / 32 (Trainer.autotune_reserve_cus; "
                         "no multi-GPU run of this path has been measured yet, so the first one tunes itself) and reported")
    ap.add_argument("--no-gc-freeze", action="store_true",
                    help="diagnostic: leave the start-up heap in the cyclic GC's young generations (shows the pause the freeze removes)")
    ap.add_argument("--no-comm-probe", action="store_true", help="N > 1: skip the exchange-off steps and the bare all-reduce timing")
    ap.add_argument("--no-secondary", action="store_true", help="skip the C5 / C2 lines timed after the headline (N = 1 only)")
    ap.add_argument("--input", default="device", choices=["device", "host"],
                    help="device (default; the contract's `value`): the batch is resident in HBM.  host: every timed step takes its "
                         "batch from a host-resident split through SpecLoader's pinned / copy-stream staging (PCIe-inclusive; "
                         "reported as such, never the headline).  The default run measures both after `value` (input_pipeline)")
    ap.add_argument("--no-input-probe", action="store_true", help="skip the host-staged / device-gather input measurements")
    ap.add_argument("--launch-check", action="store_true",
                    help="rank plumbing only (no GPU): every rank joins a gloo group, rank 0 prints {world, sum of ranks}")
    args = ap.parse_args()

    from vit_amd.launch import launch_ranks, under_launcher  # imports neither torch nor HIP

    if args.gpus > 1 and not under_launcher():
        # plain shell: this process only starts the N rank processes and never touches the GPU
        sys.exit(launch_ranks(args.gpus, os.path.abspath(__file__), sys.argv[1:]))
    if args.launch_check:
        return launch_check(args.gpus)

    # stdout carries ONE JSON line: libraries write there too (RCCL prints a version banner when its first communicator is
    # created), so file descriptor 1 is pointed at stderr for the whole run and the line goes to the saved descriptor at the end
    sys.stdout.flush()
    real_stdout = os.dup(1)
    os.dup2(2, 1)

    import torch

    from vit_amd import _cabi
    from vit_amd import ddp as ddp_mod
    from vit_amd import functional as vf

    rank, local, world = ddp_mod.init_distributed()
    # exchanging: N > 1, or VIT_DIST_SINGLE=1 (one rank, but the process group exists and every collective of the step runs:
    # the RCCL rehearsal a one-GPU box allows)
    exchanging = ddp_mod.exchange_active()
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run --nproc-per-node {args.gpus}")
    if os.environ.get("VIT_BENCH_SHARE_GPU"):  # rehearsal: all ranks on device 0 (use with VIT_DIST_BACKEND=gloo)
        local = 0
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)

    L, P, D, layers, heads, B = WORKLOADS[args.workload]
    if args.batch:
        B = args.batch
    if args.global_batch:
        if args.global_batch % world:
            raise SystemExit(f"--global-batch {args.global_batch} is not a multiple of the {world} ranks")
        B = args.global_batch // world
    autotune = args.reserve_cus < 0 and exchanging
    if args.reserve_cus < 0:
        args.reserve_cus = 0
    use_graph = not exchanging and args.graph and not args.no_graph
    module, trainer, batch, host = build_run(args.workload, B, args.precision, dev, rank, use_graph=use_graph)
    if args.reserve_cus:
        trainer.set_reserve_cus(module, args.reserve_cus)
    if args.no_overlap or use_graph:
        module.model.engine.overlap_dw = False
    elif args.overlap:
        module.model.engine.overlap_dw = True
    lib = _cabi.load()

    def barrier():
        torch.cuda.synchronize()
        if exchanging:
            torch.distributed.barrier()
        torch.cuda.synchronize()

    import itertools

    batch_iter = itertools.repeat(batch)
    if args.input == "host":
        # PCIe-inclusive variant: every step's batch is gathered from a host-resident split into pinned memory by the loader's
        # worker thread and copied on its copy stream (vit_amd/data.py: _Stager).  Never the headline (`value` = resident inputs).
        from vit_amd.data import SpecDataset, SpecLoader

        ds = SpecDataset(torch.cat([host[0].roll(i, 0) for i in range(8)]).abs_(), torch.cat([host[1]] * 8),
                         torch.cat([host[2]] * 8), task="reg", stage="train")
        loader = SpecLoader(ds, B, shuffle=True, placement="host").bind(dev)

        def stream_batches():
            e = 0
            while True:
                loader.set_epoch(e)
                yield from loader
                e += 1

        batch_iter = stream_batches()

    tuned = None
    if autotune:
        tuned = trainer.autotune_reserve_cus(module, batch, restore=False)  # these steps are warm-up here
        args.reserve_cus = trainer.reserve_cus
        if rank == 0:
            log(f"reserve_cus autotune (ms per step): {tuned} -> {args.reserve_cus}")
    if rank == 0:
        log(f"model on {dev}, {sum(p.numel() for p in module.parameters())} parameters; warm-up {args.warmup} steps")
    for i in range(args.warmup):
        try:
            trainer.training_step(module, next(batch_iter), i)
        except Exception as e:  # noqa: BLE001 - the harness must still produce its line: capture trouble -> eager launches
            if not (use_graph and i == 0):
                raise
            log(f"hipGraph capture failed ({type(e).__name__}: {e}); continuing with eager launches")
            use_graph = trainer.use_graph = False
            torch.cuda.synchronize()
            trainer.training_step(module, batch, i)
    torch.cuda.synchronize()
    if rank == 0:
        log(f"timing {args.steps} steps")
    # BASELINE.md section 2 / SURVEY.md section 8d: per-step hipEvent times, median.  One event record per step boundary on the
    # launch stream (the step ends on it: the optimizer kernels wait for the side stream and the collective) -- K + 1 records
    # in all, nothing else inside the region.  `value` is taken from the MEDIAN step; the region mean (wall clock between the
    # two barriers, the contract's bracket) is reported beside it as `mean_ms_per_step`.
    # Host-side transients are made visible instead of guessed at: `host_ms` = wall time the host spent ISSUING each step
    # (no sync inside), and every garbage collection that runs inside the region is recorded with its duration.  What r03's
    # driver line showed (region mean 36.8 ms against 35.0 for every step but one) was ONE such pause: a generation-2
    # collection of CPython's cyclic GC landing in a timed step (every step allocates a few thousand tracked objects; with the
    # ~10^6 objects that importing torch leaves on the heap a full collection takes 50-100 ms, during which nothing is
    # launched and the GPU drains).  The trainer's step loop runs with the start-up heap frozen (Trainer.freeze_heap(): one
    # collect, then gc.freeze(), so later collections scan only what the steps allocate); the bench does the same after warm-up.
    import gc

    if not args.no_gc_freeze:
        trainer.freeze_heap()
    gc_events, gc_t = [], [0.0]

    def gc_cb(phase, info):
        if phase == "start":
            gc_t[0] = time.perf_counter()
        else:
            gc_events.append({"generation": info["generation"], "ms": round((time.perf_counter() - gc_t[0]) * 1e3, 3),
                              "at_step": len(host_ms)})

    barrier()
    marks = [torch.cuda.Event(enable_timing=True) for _ in range(args.steps + 1)]
    host_ms = []
    gc.callbacks.append(gc_cb)
    t0 = time.perf_counter()
    marks[0].record()
    th = t0
    for i in range(args.steps):
        loss = trainer.training_step(module, next(batch_iter), i)
        marks[i + 1].record()
        tn = time.perf_counter()
        host_ms.append((tn - th) * 1e3)
        th = tn
    barrier()
    dt = time.perf_counter() - t0
    gc.callbacks.remove(gc_cb)
    step_ms = [marks[i].elapsed_time(marks[i + 1]) for i in range(args.steps)]
    two_streams = module.model.engine.side_stream is not None  # what the timed steps ran with (engine: overlap_dw "auto")
    # The per-GEMM HIP-event brackets (two event records around each of ~170 vit_gemm calls per step) are installed ONLY for
    # this repetition of the same steps (same state, same inputs): the clean region above ran the library's own vf.gemm, so
    # `value` carries no event record beyond the K + 1 step marks.  `roofline.instrumented_ms_per_step` is what the bracketed
    # steps took -- on ONE stream (each GEMM alone on the GPU: with the weight-gradient GEMMs on a second stream a bracket
    # would time two kernels sharing the CUs); `event_cost_ms_per_step` = bracketed - the same one-stream steps without brackets.
    records = []
    dt_inst, n_inst = None, 0
    if not args.no_kernel_timing:
        n_inst = min(args.steps, 20)
        orig_gemm = vf.gemm

        def timed_gemm(a, b, **kw):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            out = orig_gemm(a, b, **kw)
            e1.record()
            records.append((lib.vit_last_gemm_kernel().decode(), kw["M"], kw["N"], kw["K"], e0, e1))
            return out

        trainer.use_graph = False  # the per-GEMM event brackets need the eager launches
        overlap_was = module.model.engine.overlap_dw
        module.model.engine.overlap_dw = False
        vf.gemm = timed_gemm
        try:
            barrier()
            t1 = time.perf_counter()
            for i in range(n_inst):
                trainer.training_step(module, batch, i)
            barrier()
            dt_inst = time.perf_counter() - t1
        finally:
            vf.gemm = orig_gemm
        # the same one-stream steps WITHOUT the brackets: instrumented - this = what the ~340 event records per step cost
        one_stream_ms = None
        if not exchanging:
            try:
                one_stream_ms = time_steps(trainer, module, itertools.repeat(batch), min(n_inst, 10), 1)
            finally:
                module.model.engine.overlap_dw = overlap_was
        module.model.engine.overlap_dw = overlap_was
    if exchanging:  # MAX over ranks, of the region and of every step
        t = torch.tensor([dt] + step_ms, dtype=torch.float64, device=dev)
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
        dt, step_ms = float(t[0]), [float(x) for x in t[1:]]
    final_loss = float(loss.detach())
    srt = sorted(step_ms)
    median_ms = srt[len(srt) // 2] if len(srt) % 2 else 0.5 * (srt[len(srt) // 2 - 1] + srt[len(srt) // 2])
    if rank == 0:
        log("per-step ms (hipEvents on the launch stream): " + " ".join(f"{x:.2f}" for x in step_ms))
        log("host issue ms per step: " + " ".join(f"{x:.2f}" for x in host_ms))
        log(f"garbage collections inside the region: {gc_events}")
        log(f"median {median_ms:.3f} ms, min {srt[0]:.3f}, max {srt[-1]:.3f}, region mean {dt / args.steps * 1e3:.3f} ms")

    # ---- N > 1: what the gradient exchange costs.  (a) the same steps with the exchange switched off (replicas drift
    # apart, which no longer matters: `value` is already taken), (b) the bare all-reduce of the flat gradient buffer in
    # the step's own buckets with nothing else on the GPU -> algorithm / bus bandwidth over xGMI.
    comm = None
    if exchanging:
        dist = torch.distributed
        eng = module.model.engine
        red = trainer.reducer
        comm = {"backend": dist.get_backend(), "world_size": dist.get_world_size(), "exchange": red.mode,
                "collectives_per_step": red.calls_per_step, "bytes_per_step": red.bytes_per_step,
                "reserve_cus": args.reserve_cus,
                "reserve_cus_autotune_ms": {str(k): round(v, 3) for k, v in tuned.items()} if tuned else None}
        if not args.no_comm_probe:
            try:
                comm.update(comm_probe(args, trainer, module, eng, red, batch, barrier, dev, world, median_ms))
            except Exception as e:  # noqa: BLE001 - the probe is extra evidence; `value` above is already measured
                comm["probe_error"] = f"{type(e).__name__}: {e}"

    mean_ms_per_step = dt / args.steps * 1e3
    ms_per_step = median_ms
    value = world * B / (median_ms * 1e-3)
    F = 4 * D
    flop_img = train_flop_per_image(L, P, D, layers, F)

    roofline = None
    kernels = {}
    if not args.no_kernel_timing and rank == 0:
        agg = {}
        for var, M, N, K, e0, e1 in records:
            ms = e0.elapsed_time(e1)
            a = agg.setdefault(var, [0, 0.0, 0.0])
            a[0] += 1
            a[1] += ms
            a[2] += 2.0 * M * N * K
        for var, (n, ms, fl) in agg.items():
            kernels[var] = dict(launches=n, total_ms=round(ms, 3), mean_us=round(ms / n * 1e3, 2),
                                tflops=round(fl / (ms * 1e-3) / 1e12, 1))
        dom = max(agg.items(), key=lambda kv: kv[1][1])
        var, (n, ms, fl) = dom
        achieved = fl / (ms * 1e-3) / 1e12
        gemm_ms = sum(v[1] for v in agg.values())
        gemm_fl = sum(v[2] for v in agg.values())
        # HBM bytes per launch of that kernel: NOT measured in this run (PMC counters need rocprofv3 around the process);
        # it is the figure from the newest committed PMC passes of this same command, and `traffic_source` says which
        # A counter file is only quoted when it was collected from THIS library: the summary records the sha256 of the kernel
        # sources (`lib_stamp`, vit_amd.build.source_stamp()); a file from other kernels is refused and named as stale.
        traffic, traffic_source = None, None
        try:
            import glob

            from vit_amd.build import source_stamp
            for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "*pmc_traffic.json")))[-1:]:
                doc = json.load(open(path))
                ent = doc["kernels"].get(var)
                name = f"profiles/{os.path.basename(path)}"
                if doc.get("lib_stamp") != source_stamp():
                    traffic_source = (f"{name} is STALE (collected from other kernel sources: lib_stamp "
                                      f"{str(doc.get('lib_stamp'))[:12]} != {source_stamp()[:12]}); not quoted")
                elif ent and args.workload == "vit_b16_224" and B == 256:
                    traffic = ent["hbm_bytes_per_launch"]
                    traffic_source = f"{name} (rocprofv3 --pmc passes of this command at the same kernel sources, committed; constant, not re-measured here)"
        except Exception:  # noqa: BLE001 - the profile summary is optional evidence, never required to run
            traffic = None
        roofline = {
            "bound": "mfma", "kernel": var, "achieved": round(achieved, 2), "peak": PEAK_BF16_DENSE / 1e12,
            "unit": "TFLOP/s", "frac": round(achieved * 1e12 / PEAK_BF16_DENSE, 4), "traffic": traffic,
            "traffic_source": traffic_source,
            "flop_per_launch": fl / n, "mean_launch_us": round(ms / n * 1e3, 2), "launches_timed": n,
            "share_of_step_time": round(ms / (dt_inst * 1e3), 3),
            "all_gemm_tflops": round(gemm_fl / (gemm_ms * 1e-3) / 1e12, 1),
            "all_gemm_share_of_step_time": round(gemm_ms / (dt_inst * 1e3), 3),
            "instrumented_steps": n_inst, "instrumented_ms_per_step": round(dt_inst / n_inst * 1e3, 3),
            "one_stream_uninstrumented_ms_per_step": None if one_stream_ms is None else round(one_stream_ms, 3),
            "event_cost_ms_per_step": None if one_stream_ms is None else round(dt_inst / n_inst * 1e3 - one_stream_ms, 3),
            "step_tflops": round(value / world * flop_img / 1e12, 2),
            "step_frac": round(value / world * flop_img / PEAK_BF16_DENSE, 4),
            "power_limited_peak": {"value": SUSTAINED_BF16_DENSE_RANDOM / 1e12, "unit": "TFLOP/s",
                                   "frac": round(achieved * 1e12 / SUSTAINED_BF16_DENSE_RANDOM, 4),
                                   "step_frac": round(value / world * flop_img / SUSTAINED_BF16_DENSE_RANDOM, 4),
                                   "source": "profiles/r05_b_mfma_power.txt: registers-only 16x16x32 bf16 MFMA loop on random operands, "
                                             "2.17 GHz at 1320 W of the 1400 W cap (measured constant, not re-measured here)"},
        }

    # ---- N = 1: what fit() pays for its inputs, and the other single-GPU configurations, in this same driver-observed run
    input_pipeline, secondary = None, None
    if world == 1 and not exchanging:
        if not args.no_input_probe and args.input == "device":
            try:
                input_pipeline = input_pipeline_probe(trainer, module, host, dev, B, min(args.steps, 12), median_ms)
                log(f"input pipeline: {input_pipeline}")
            except Exception as e:  # noqa: BLE001 - extra evidence; `value` above is already measured
                input_pipeline = {"error": f"{type(e).__name__}: {e}"}
        if not args.no_secondary and args.workload == "vit_b16_224" and not args.batch:
            del batch, batch_iter
            module.model.engine._drop_arenas()
            try:
                secondary = secondary_workloads(dev, rank, args.precision)
            except Exception as e:  # noqa: BLE001
                secondary = {"error": f"{type(e).__name__}: {e}"}
    gc.unfreeze()

    cpu_baseline = None
    if rank == 0 and world == 1 and not exchanging and not args.no_cpu_baseline:
        cpu_baseline = run_cpu_baseline(args.workload, L, P, D, layers, heads)

    if rank == 0:
        out = {
            "metric": "images/sec ViT-B/16 224^2 bf16 train step" if args.workload == "vit_b16_224" else f"images/sec {args.workload} bf16 train step",
            "value": round(value, 2), "unit": "images/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(ms_per_step, 3), "mean_ms_per_step": round(mean_ms_per_step, 3),
            "timing": {"value_from": "median of per-step hipEvent times over the K timed steps (BASELINE.md section 2)",
                       "min_ms": round(srt[0], 3), "max_ms": round(srt[-1], 3),
                       "region_wall_ms": round(dt * 1e3, 3), "step_ms": [round(x, 3) for x in step_ms],
                       "host_issue_ms": [round(x, 2) for x in host_ms], "gc_in_region": gc_events,
                       "heap_frozen": not args.no_gc_freeze},
            "higher_is_better": True, "scaling": "strong" if args.global_batch else "weak",
            "vs_baseline": None,
            "input": "resident in HBM" if args.input == "device" else "host-resident split, pinned staging + copy stream per step (PCIe-inclusive)",
            "dtype": "bf16" if args.precision == "bf16-mixed" else "f32 (split-bf16 x3 MFMA)", "data": "synthetic",
            "config": {"workload": f"{args.workload}: flux[{B},{L}] f32/GPU, patch {P}, {L // P}+1 tokens, hidden {D}, "
                                   f"{heads} heads, {layers} layers, MLP {F}; fwd+bwd+clip0.5+AdamW, dropout 0.1 on",
                       "global_batch": B * world, "parallelism": f"dp{world}", "train_gflop_per_image": round(flop_img / 1e9, 2),
                       "final_loss": final_loss, "reserve_cus": args.reserve_cus,
                       "launch": "one hipGraph replay per step" if use_graph else
                       ("eager launches, weight-gradient GEMMs on a second HIP stream" if two_streams
                        else "eager launches, one stream")},
            "roofline": roofline, "cpu_baseline": cpu_baseline, "secondary": secondary, "input_pipeline": input_pipeline,
            "comm": comm, "kernels": kernels,
        }
        sys.stdout.flush()
        os.write(real_stdout, (json.dumps(out) + "\n").encode())
    if exchanging:
        torch.distributed.barrier()
        torch.distributed.destroy_process_group()


def run_cpu_baseline(workload, L, P, D, layers, heads):
    """The CPU oracle (plain fp32 torch restatement of the reference path; 'port') on this host's cores: same step
    (fwd, dropout on, bwd, clip 0.5, AdamW), bounded sample."""
    import torch

    from oracle import refvit

    # the GPU box gives this job a 16-core share of a much larger host: os.cpu_count() would oversubscribe it
    try:
        cores = len(os.sched_getaffinity(0))
    except AttributeError:
        cores = os.cpu_count() or 1
    cores = max(1, min(cores, int(os.environ.get("VIT_BENCH_CPU_THREADS", "16"))))
    torch.set_num_threads(cores)
    log(f"cpu baseline: {cores} threads, building oracle state")
    rc = refvit.RefConfig(image_size=L, patch_size=P, hidden_size=D, num_hidden_layers=layers,
                          num_attention_heads=heads, stride_size=P, loss_name="mae")
    Bc = 8 if D >= 512 else 64
    sd = refvit.make_state_dict(rc, 7)
    flux, _, labels = refvit.make_inputs(rc, Bc, 8)
    tr = refvit.RefTrainer(rc, sd, training=True)
    t0 = time.perf_counter()
    tr.step(flux, labels)  # warm-up
    log(f"cpu baseline: warm-up step {time.perf_counter() - t0:.1f} s")
    n = 3  # 1 warm-up + 3 timed steps
    t0 = time.perf_counter()
    for i in range(n):
        tr.step(flux, labels)
        log(f"cpu baseline: step {i + 1}/{n} at {time.perf_counter() - t0:.1f} s")
    dt = time.perf_counter() - t0
    return {"value": round(Bc * n / dt, 3), "unit": "images/s", "cores": torch.get_num_threads(), "kind": "port",
            "sample": f"{workload}: batch {Bc}, 1 warm-up + {n} timed steps, fp32, dropout on, oracle/refvit.py RefTrainer "
                      f"({dt / n:.2f} s/step)"}


if __name__ == "__main__":
    main()
