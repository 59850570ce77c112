#!/usr/bin/env python3
"""One-GPU prior for the 8-GPU `reserve_cus` autotune (VERDICT r4 #4 iii; tools, not product).

The ViT-B/16 step of bench.py with a STAND-IN collective in the place of the RCCL all-reduce: for every gradient bucket the
engine reports complete, `standin_kernel` (tools/micro/standin_collective.hip) is launched on a stream of its own -- k persistent
workgroups streaming the bucket through their CUs, paced so that the bucket takes what a ring all-reduce over xGMI would take at
the given bus bandwidth (bytes * 2 (n - 1) / n / busbw, n = 8) -- and joined before the optimizer, exactly where
vit_amd.ddp.GradAllReducer enqueues and joins the real thing.  Swept over k (channels) x reserve_cus x busbw: median step time,
and its distance to the step without any collective = what the overlap costs.  No link is involved; this measures only how the
one-workgroup-per-CU kernels of the backward share the chip with a resident communication kernel.

    python tools/standin_sweep.py [--steps 10] [--out gpurun_out/standin.json]
"""
import argparse
import ctypes
import itertools
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def build_standin():
    import torch

    src = os.path.join(ROOT, "tools", "micro", "standin_collective.hip")
    out = os.path.join(ROOT, "tools", "micro", "libstandin.so")
    if not os.path.exists(out) or os.path.getmtime(out) < os.path.getmtime(src):
        tl = os.path.join(os.path.dirname(torch.__file__), "lib")
        subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-shared", "-fPIC", "-o", out, src, f"-L{tl}",
                        "-lamdhip64", "-Wl,-rpath," + tl], check=True)
    lib = ctypes.CDLL(out)
    lib.standin_launch.argtypes = [ctypes.c_void_p, ctypes.c_long, ctypes.c_int, ctypes.c_double, ctypes.c_void_p]
    lib.standin_launch.restype = ctypes.c_int
    return lib


class StandIn:
    """Duck-typed reducer (bucket_ready / finish / mode) for Trainer.training_step."""
    mode = "standin"

    def __init__(self, lib, eng, wgs, busbw, world=8):
        import torch

        self.lib, self.eng, self.wgs, self.busbw, self.world = lib, eng, wgs, busbw, world
        self.stream = torch.cuda.Stream(device=eng.flat.device)
        self.events = []

    def bucket_ready(self, lo, hi):
        import torch

        if hi <= lo or self.wgs <= 0:
            return
        g = self.eng.grads[lo:hi]
        cur = torch.cuda.current_stream(g.device)  # the stream the bucket's last kernels were enqueued on
        self.stream.wait_stream(cur)
        secs = (hi - lo) * 4 * 2.0 * (self.world - 1) / self.world / (self.busbw * 1e9)
        rc = self.lib.standin_launch(g.data_ptr(), hi - lo, self.wgs, secs, self.stream.cuda_stream)
        assert rc == 0, rc
        ev = torch.cuda.Event()
        ev.record(self.stream)
        self.events.append(ev)

    def finish(self):
        import torch

        cur = torch.cuda.current_stream(self.eng.flat.device)
        for ev in self.events:
            cur.wait_event(ev)
        self.events = []


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--out", default=os.path.join(ROOT, "gpurun_out", "standin.json"))
    ap.add_argument("--no-overlap", action="store_true")
    a = ap.parse_args()
    import torch

    import bench

    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    lib = build_standin()
    module, trainer, batch, _ = bench.build_run("vit_b16_224", 256, "bf16-mixed", dev, 0)
    eng = module.model.engine
    eng.overlap_dw = not a.no_overlap
    rep = itertools.repeat(batch)

    def run(standin, reserve):
        trainer.set_reserve_cus(module, reserve)
        trainer.reducer = standin
        eng.grad_ready_cb = standin.bucket_ready if standin is not None else None
        return bench.time_steps(trainer, module, rep, a.steps, 2)

    run(None, 0)  # warm-up
    trainer.freeze_heap()
    res = {"workload": "vit_b16_224 B=256 bf16-mixed", "steps": a.steps, "second_stream": bool(eng.overlap_dw), "world_modelled": 8,
           "bucket_bytes": [4 * (hi - lo) for lo, hi in eng.layout.buckets()], "rows": []}
    base = {}
    for reserve in (0, 8, 16, 32):
        base[reserve] = run(None, reserve)
        print(f"no collective, reserve {reserve:2d}: {base[reserve]:.3f} ms", flush=True)
    res["no_collective_ms"] = base
    for busbw in (300.0, 150.0):
        for wgs in (8, 16, 32):
            for reserve in (0, 8, 16, 32):
                ms = run(StandIn(lib, eng, wgs, busbw), reserve)
                total_s = sum(res["bucket_bytes"]) * 1.75 / (busbw * 1e9)
                row = {"busbw_GBps": busbw, "channels": wgs, "reserve_cus": reserve, "ms_per_step": round(ms, 3),
                       "vs_no_collective_reserve0_ms": round(ms - base[0], 3), "vs_same_reserve_ms": round(ms - base[reserve], 3),
                       "collective_busy_ms_per_step": round(total_s * 1e3, 3)}
                res["rows"].append(row)
                print(row, flush=True)
    os.makedirs(os.path.dirname(a.out), exist_ok=True)
    json.dump(res, open(a.out, "w"), indent=1)


if __name__ == "__main__":
    main()
