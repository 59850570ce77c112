#!/usr/bin/env python3
"""vit_adamw_step alone at the ViT-B / ViT-L parameter counts: us per call and HBM TB/s against 30 B per parameter."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import vit_amd.functional as vf
dev = torch.device("cuda:0")
for n in (85_254_912, 303_000_576):
    p = torch.randn(n, device=dev); g = torch.randn(n, device=dev) * 1e-3
    m = torch.zeros(n, device=dev); v = torch.zeros(n, device=dev)
    pb = torch.empty(n, device=dev, dtype=torch.bfloat16)
    sq = torch.ones(1, device=dev)
    f = lambda: vf.adamw_step(p, g, m, v, pb, lr=1e-3, step=3, sqnorm=sq, max_norm=0.5)
    f(); f(); torch.cuda.synchronize()
    ts = []
    for _ in range(5):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); f(); f(); f(); e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) / 3 * 1e3)
    t = sorted(ts)[2]
    print(f"n = {n / 1e6:.1f} M: {t:8.1f} us = {n * 30 / t / 1e6:.2f} TB/s", flush=True)
    del p, g, m, v, pb
