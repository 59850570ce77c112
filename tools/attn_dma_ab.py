#!/usr/bin/env python3
"""A/B of the resident two-kernel attention backward's prologue in ONE process: register-staged images (attn_bwd_dma 0) against
LDS-DMA in reading order with per-tile counted waits (1), interleaved rounds.  usage: python tools/attn_dma_ab.py [B,H,T,dh ...]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import vit_amd.functional as vf
from vit_amd import _cabi

dev = torch.device("cuda:0")
shapes = [tuple(map(int, a.split(","))) for a in sys.argv[1:]] or [(32, 16, 577, 64), (64, 12, 300, 64), (256, 12, 197, 64)]
_cabi.set_option("attn_bwd_fused", 0)  # the two-kernel path everywhere
for B, H, T, dh in shapes:
    M, D = B * T, H * dh
    qkv = (torch.randn(M, 3 * D, device=dev) * 0.5).to(torch.bfloat16)
    dctx = (torch.randn(M, D, device=dev) * 0.5).to(torch.bfloat16)
    ctx = torch.empty(M, D, device=dev, dtype=torch.bfloat16); lo = torch.empty_like(ctx)
    lse = torch.empty(B * H, T, device=dev); delta = torch.empty(B * H, T, device=dev)
    cs = torch.empty(3 * D, device=dev)
    dp = (0.1, 1, 2)
    vf.attention_fwd(qkv, B, H, T, dh, dh ** -0.5, dropout=dp, ctx=ctx, lse=lse, ctx_lo=lo)
    out, times = {}, {0: [], 1: []}
    for rnd in range(6):
        for mode in (0, 1):
            _cabi.set_option("attn_bwd_dma", mode)
            d = torch.empty_like(qkv)
            f = lambda: vf.attention_bwd(qkv, ctx, dctx, lse, B, H, T, dh, dh ** -0.5, dropout=dp, dqkv=d, delta=delta, colsum_out=cs, ctx_lo=lo)
            f(); torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(5):
                f()
            e1.record(); torch.cuda.synchronize()
            if rnd:
                times[mode].append(e0.elapsed_time(e1) / 5 * 1e3)
            out[mode] = (d.clone(), cs.clone(), delta.clone())
    same = all(torch.equal(a, b) for a, b in zip(out[0], out[1]))
    med = {m: sorted(v)[len(v) // 2] for m, v in times.items()}
    print(f"B {B} H {H} T {T} dh {dh}: register-staged {med[0]:7.1f} us   DMA in reading order {med[1]:7.1f} us   ({100 * (med[1] / med[0] - 1):+.1f} %)   "
          f"bit-identical: {same}", flush=True)
_cabi.set_option("attn_bwd_dma", 1)
_cabi.set_option("attn_bwd_fused", 4)
