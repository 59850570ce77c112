#!/usr/bin/env python3
"""Diagnostic: the dW GEMM (K = all tokens, split-K) with its operands as stored (both read through the transposing LDS
path, 'TT') against the same product on pre-transposed copies ('NT'): does the layout or the long-K / split-K shape set
the rate?"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import vit_amd.functional as vf
dev = torch.device("cuda:0")
M, D, F = 50432, 768, 3072
g = torch.Generator(device="cpu").manual_seed(0)
R = lambda *s: (torch.randn(s, generator=g) * 0.5).to(dev).to(torch.bfloat16)
dy, x = R(M, F), R(M, D)            # dW fc1 [3072 x 768] = dy^T x
dyT, xT = dy.t().contiguous(), x.t().contiguous()
out = torch.empty(F, D, device=dev)
def t(fn):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / 5 * 1e3
fl = 2 * M * F * D
for rnd in range(2):
    tt = t(lambda: vf.gemm(dy, x, M=F, N=D, K=M, a_trans=True, b_trans=True, out=out, split_k=-1))
    ref = out.clone()
    nt = t(lambda: vf.gemm(dyT, xT, M=F, N=D, K=M, out=out, split_k=-1))
    err = float((out - ref).norm() / ref.norm())
    print(f"dW fc1: TT {tt:6.1f} us ({fl / tt / 1e6:5.0f} TF)   NT on transposed copies {nt:6.1f} us ({fl / nt / 1e6:5.0f} TF)   rel diff {err:.1e}", flush=True)
