#!/usr/bin/env python3
"""Timing decomposition of the fused / persistent attention backward by skipping pieces (vit_set_option("attn_debug"))."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import vit_amd.functional as vf
from vit_amd import _cabi
dev = torch.device("cuda:0")
B, H, T, dh = 256, 12, 197, 64
M, D = B * T, H * dh
qkv = (torch.randn(M, 3 * D, device=dev) * 0.5).to(torch.bfloat16)
dctx = (torch.randn(M, D, device=dev) * 0.5).to(torch.bfloat16)
ctx = torch.empty(M, D, device=dev, dtype=torch.bfloat16); lo = torch.empty_like(ctx)
lse = torch.empty(B * H, T, device=dev); dqkv = torch.empty_like(qkv); delta = torch.empty(B * H, T, device=dev)
cs = torch.empty(3 * D, device=dev)
def t(fn, n=10):
    fn(); fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
sc = dh ** -0.5
vf.attention_fwd(qkv, B, H, T, dh, sc, dropout=(0.1, 1, 2), ctx=ctx, lse=lse, ctx_lo=lo)
for fused in (1, 3):
    _cabi.set_option("attn_bwd_fused", fused)
    for dp in ((0.1, 1, 2), (0.0, 0, 0)):
        row = []
        for dbg in (0, 1, 2, 3, 4, 7, 15, 8):
            _cabi.set_option("attn_debug", dbg)
            row.append((dbg, t(lambda: vf.attention_bwd(qkv, ctx, dctx, lse, B, H, T, dh, sc, dropout=dp, dqkv=dqkv, delta=delta,
                                                        colsum_out=cs, ctx_lo=lo))))
        _cabi.set_option("attn_debug", 0)
        print(f"mode{fused} dropout {dp[0]}: " + "  ".join(f"dbg{d}={v:.0f}" for d, v in row), flush=True)
print("dbg bits: 1 = no phase B, 2 = no phase A, 4 = no Q/dO/K staging, 8 = no dK/dV stores")
_cabi.set_option("attn_bwd_fused", 4)
