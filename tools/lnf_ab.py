#!/usr/bin/env python3
"""Time layernorm_fwd_residual (x + bf16 delta -> xsum f32, y bf16) for A/B runs of variant builds (VIT_AMD_LIB).
usage: python tools/lnf_ab.py [M D]"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import vit_amd.functional as vf
dev = torch.device("cuda:0")
M, D = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (50432, 768)
g = torch.Generator(device="cpu").manual_seed(0)
x = torch.randn((M, D), generator=g).to(dev)
delta = torch.randn((M, D), generator=g).to(dev).to(torch.bfloat16)
gam, bet = torch.randn(D, generator=g).to(dev), torch.randn(D, generator=g).to(dev)
xsum = torch.empty_like(x); y = torch.empty((M, D), device=dev, dtype=torch.bfloat16)
mean = torch.empty(M, device=dev); rstd = torch.empty(M, device=dev)
run = lambda: vf.layernorm_fwd_residual(x, delta, xsum, gam, bet, 1e-12, out=y, mean=mean, rstd=rstd)
for _ in range(30): run()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(50): run()
e1.record(); torch.cuda.synchronize()
t = e0.elapsed_time(e1) / 50 * 1e3
print(os.path.basename(os.environ.get("VIT_AMD_LIB", "libvit_amd.so")), f"M={M} D={D} ln_fwd_residual: {t:.1f} us = {M * D * 12 / t / 1e6:.2f} TB/s", float(y.float().sum()))
