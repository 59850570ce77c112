#!/usr/bin/env python3
"""Attention forward / backward at any (B, H, T, dh): us per call and TFLOP/s against the algorithmic 4 (fwd) / 10 (bwd:
five products) x B H T^2 dh FLOP -- the long-sequence regime of the reference's stride sweep (configs/sweep.yaml: S = 1 at
L = 4096 gives T = 4034) runs the TILED kernels (T > 592), the BASELINE shapes the resident ones.
usage: python tools/attn_shape_bench.py B,H,T,dh [B,H,T,dh ...] [--dropout 0.1] [--json out.json]"""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import vit_amd.functional as vf

args = [a for a in sys.argv[1:] if a[0].isdigit() and a.count(",") == 3]
opts = dict(zip(sys.argv[1:], sys.argv[2:]))
pd = float(opts.get("--dropout", 0.1))
dev = torch.device("cuda:0")
lib = vf._cabi.load() if hasattr(vf, "_cabi") else None


def t(fn, n=6):
    fn(); fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


rows = []
for spec in args:
    B, H, T, dh = map(int, spec.split(","))
    M, D = B * T, H * dh
    qkv = (torch.randn(M, 3 * D, device=dev) * 0.5).to(torch.bfloat16)
    dctx = (torch.randn(M, D, device=dev) * 0.5).to(torch.bfloat16)
    ctx = torch.empty(M, D, device=dev, dtype=torch.bfloat16); lo = torch.empty_like(ctx)
    lse = torch.empty(B * H, T, device=dev); dqkv = torch.empty_like(qkv); delta = torch.empty(B * H, T, device=dev)
    dp = (pd, 1, 2)
    sc = dh ** -0.5
    f = t(lambda: vf.attention_fwd(qkv, B, H, T, dh, sc, dropout=dp, ctx=ctx, lse=lse, ctx_lo=lo))
    b = t(lambda: vf.attention_bwd(qkv, ctx, dctx, lse, B, H, T, dh, sc, dropout=dp, dqkv=dqkv, delta=delta, ctx_lo=lo))
    ff, bf_ = 4.0 * B * H * T * T * dh, 10.0 * B * H * T * T * dh
    row = dict(B=B, H=H, T=T, dh=dh, dropout=pd, fwd_us=round(f, 1), bwd_us=round(b, 1), fwd_tflops=round(ff / f / 1e6, 1),
               bwd_tflops=round(bf_ / b / 1e6, 1), fwd_frac_of_2500=round(ff / f / 1e6 / 2500, 4), bwd_frac_of_2500=round(bf_ / b / 1e6 / 2500, 4))
    rows.append(row)
    print(f"B {B:4d} H {H:3d} T {T:5d} dh {dh:3d}: fwd {f:9.1f} us = {row['fwd_tflops']:6.1f} TFLOP/s ({100 * row['fwd_frac_of_2500']:.1f} %)   "
          f"bwd {b:9.1f} us = {row['bwd_tflops']:6.1f} TFLOP/s ({100 * row['bwd_frac_of_2500']:.1f} %)", flush=True)
if "--json" in opts:
    json.dump({"tool": "tools/attn_shape_bench.py", "peak_tflops": 2500, "rows": rows}, open(opts["--json"], "w"), indent=1)
