#!/usr/bin/env python3
"""Time one attention-backward mode at the ViT-B shape (for A/B runs of variant builds through VIT_AMD_LIB).
usage: python tools/pipe_ab.py [mode=4] [dropout=0.1]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import vit_amd.functional as vf
from vit_amd import _cabi
mode = int(sys.argv[1]) if len(sys.argv) > 1 else 4
pd = float(sys.argv[2]) if len(sys.argv) > 2 else 0.1
dev = torch.device("cuda:0")
B, H, T, dh = 256, 12, 197, 64
M, D = B * T, H * dh
qkv = (torch.randn(M, 3 * D, device=dev) * 0.5).to(torch.bfloat16)
dctx = (torch.randn(M, D, device=dev) * 0.5).to(torch.bfloat16)
ctx = torch.empty(M, D, device=dev, dtype=torch.bfloat16); lo = torch.empty_like(ctx)
lse = torch.empty(B * H, T, device=dev); dqkv = torch.empty_like(qkv); delta = torch.empty(B * H, T, device=dev)
cs = torch.empty(3 * D, device=dev)
dp = (pd, 1, 2)
vf.attention_fwd(qkv, B, H, T, dh, dh ** -0.5, dropout=dp, ctx=ctx, lse=lse, ctx_lo=lo)
_cabi.set_option("attn_bwd_fused", mode)
f = lambda: vf.attention_bwd(qkv, ctx, dctx, lse, B, H, T, dh, dh ** -0.5, dropout=dp, dqkv=dqkv, delta=delta, colsum_out=cs, ctx_lo=lo)
f(); f(); torch.cuda.synchronize()
ts = []
for _ in range(3):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10): f()
    e1.record(); torch.cuda.synchronize()
    ts.append(e0.elapsed_time(e1) / 10 * 1e3)
print(f"{os.path.basename(os.environ.get('VIT_AMD_LIB', 'libvit_amd.so'))} mode {mode} dropout {pd}: " + " ".join(f"{t:.1f}" for t in ts) + " us")
