#!/usr/bin/env python3
"""Time the attention forward at the ViT-B shape (A/B runs of variant builds through VIT_AMD_LIB).
usage: python tools/fwd_ab.py [dropout=0.1] [B H T]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import vit_amd.functional as vf
pd = float(sys.argv[1]) if len(sys.argv) > 1 else 0.1
B, H, T = (int(v) for v in sys.argv[2:5]) if len(sys.argv) > 4 else (256, 12, 197)
dh = 64
dev = torch.device("cuda:0")
M, D = B * T, H * dh
qkv = (torch.randn(M, 3 * D, device=dev) * 0.5).to(torch.bfloat16)
ctx = torch.empty(M, D, device=dev, dtype=torch.bfloat16); lo = torch.empty_like(ctx)
lse = torch.empty(B * H, T, device=dev)
f = lambda: vf.attention_fwd(qkv, B, H, T, dh, dh ** -0.5, dropout=(pd, 1, 2), ctx=ctx, lse=lse, ctx_lo=lo)
f(); f(); torch.cuda.synchronize()
ts = []
for _ in range(3):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10): f()
    e1.record(); torch.cuda.synchronize()
    ts.append(e0.elapsed_time(e1) / 10 * 1e3)
print(f"{os.path.basename(os.environ.get('VIT_AMD_LIB', 'libvit_amd.so'))} fwd B={B} H={H} T={T} dropout {pd}: " + " ".join(f"{t:.1f}" for t in ts) + " us")
