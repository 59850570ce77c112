#!/usr/bin/env python3
"""A few launches of each attention kernel at the ViT-B shape, for rocprofv3 --pmc passes (tools/pmc_kernels.py)."""
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
dp = (0.1, 1, 2)
for _ in range(3):
    vf.attention_fwd(qkv, B, H, T, dh, dh ** -0.5, dropout=dp, ctx=ctx, lse=lse, ctx_lo=lo)
    for fused in (0, 4):
        _cabi.set_option("attn_bwd_fused", fused)
        vf.attention_bwd(qkv, ctx, dctx, lse, B, H, T, dh, dh ** -0.5, dropout=dp, dqkv=dqkv, delta=delta, colsum_out=cs, ctx_lo=lo)
torch.cuda.synchronize()
