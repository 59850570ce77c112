#!/usr/bin/env python3
"""Attention kernels at the ViT-B shape (B=256, H=12, T=197, dh=64): forward, two-kernel backward, pair-pipelined backward;
with / without dropout and the context residual.  Prints us per call and the effective HBM rate against the algorithmic
bytes (fwd: qkv + ctx (+lo); bwd: qkv + ctx (+lo) + dctx + dqkv)."""
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
lo = torch.empty_like(ctx)
lse = torch.empty(B * H, T, device=dev)
dqkv = torch.empty_like(qkv); delta = torch.empty(B * H, T, device=dev)
cs = torch.empty(3 * D, device=dev)
def t(fn, n=10):
    fn(); fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
sc = dh ** -0.5
for dp in ((0.1, 1, 2), (0.0, 0, 0)):
    for use_lo in (True, False):
        l = lo if use_lo else None
        f = t(lambda: vf.attention_fwd(qkv, B, H, T, dh, sc, dropout=dp, ctx=ctx, lse=lse, ctx_lo=l))
        fb = (M * 3 * D + M * D * (2 if use_lo else 1)) * 2
        bb = (M * 3 * D * 2 + M * D * (3 if use_lo else 2)) * 2
        res = {}
        for fused in (0, 4):
            _cabi.set_option("attn_bwd_fused", fused)
            res[fused] = t(lambda: vf.attention_bwd(qkv, ctx, dctx, lse, B, H, T, dh, sc, dropout=dp, dqkv=dqkv, delta=delta,
                                                    colsum_out=cs, ctx_lo=l))
            res[fused, "d"] = dqkv.clone()
        err3 = float((res[0, "d"].float() - res[4, "d"].float()).norm() / res[0, "d"].float().norm())
        print(f"dropout {dp[0]} residual {int(use_lo)}: fwd {f:6.1f} us ({fb / f / 1e6:.2f} TB/s)  bwd two-kernel {res[0]:6.1f} us  "
              f"pipelined {res[4]:6.1f} us ({bb / res[4] / 1e6:.2f} TB/s)  rel diff {err3:.1e}", flush=True)
_cabi.set_option("attn_bwd_fused", 4)
