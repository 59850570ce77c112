#!/usr/bin/env python3
"""Resident attention kernels at the ViT-B shape (B=256, H=12, T=197, dh=64), per attn_split setting."""
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
ctx = torch.empty(M, D, device=dev, dtype=torch.bfloat16)
lse = torch.empty(B * H, T, device=dev)
dqkv = torch.empty_like(qkv); delta = torch.empty(B * H, T, device=dev)
drop = (0.1, 1, 2)
def t(fn, n=5):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
for dp in ((0.1, 1, 2), (0.0, 0, 0)):
    _cabi.set_option("attn_split", 2)
    f = t(lambda: vf.attention_fwd(qkv, B, H, T, dh, dh ** -0.5, dropout=dp, ctx=ctx, lse=lse))
    b = t(lambda: vf.attention_bwd(qkv, ctx, dctx, lse, B, H, T, dh, dh ** -0.5, dropout=dp, dqkv=dqkv, delta=delta))
    print(f"dropout p={dp[0]}: fwd {f:6.1f} us  bwd {b:6.1f} us", flush=True)
ref = None
for rnd in range(1):
    for split in (1, 2):
        _cabi.set_option("attn_split", split)
        f = t(lambda: vf.attention_fwd(qkv, B, H, T, dh, dh ** -0.5, dropout=drop, ctx=ctx, lse=lse))
        b = t(lambda: vf.attention_bwd(qkv, ctx, dctx, lse, B, H, T, dh, dh ** -0.5, dropout=drop, dqkv=dqkv, delta=delta))
        if ref is None: ref = (ctx.clone(), dqkv.clone())
        assert torch.equal(ctx, ref[0]) and torch.equal(dqkv, ref[1]), split
        print(f"attn_split={split}: fwd {f:6.1f} us  bwd {b:6.1f} us", flush=True)
_cabi.set_option("attn_split", 2)
