#!/usr/bin/env python3
"""Experiment: de-phase the epilogue bursts of the ping-pong GEMM (gemm_debug bit 16, delay = debug>>8 x 8128 cycles)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import vit_amd.functional as vf
from vit_amd import _cabi
from vit_amd._cabi import ACT_DGELU, ACT_GELU
dev = torch.device("cuda:0")
M, D, F = 50432, 768, 3072
g = torch.Generator(device="cpu").manual_seed(0)
R = lambda *s: (torch.randn(s, generator=g) * 0.5).to(dev).to(torch.bfloat16)
x768, x3072 = R(M, D), R(M, F)
W1, W2, Wqkv = R(F, D), R(D, F), R(3 * D, D)
b3072, b2304 = torch.randn(F, device=dev), torch.randn(3 * D, device=dev)
o3072 = torch.empty(M, F, device=dev, dtype=torch.bfloat16)
o2304 = torch.empty(M, 3 * D, device=dev, dtype=torch.bfloat16)
aux = torch.empty(M, F, device=dev, dtype=torch.bfloat16)
cases = {
    "fwd fc1 gelu": (lambda: vf.gemm(x768, W1, M=M, N=F, K=D, out=o3072, bias=b3072, act=ACT_GELU, aux_out=aux), 2 * M * F * D),
    "dX fc2 dgelu": (lambda: vf.gemm(x768, W2, M=M, N=F, K=D, b_trans=True, out=o3072, act=ACT_DGELU, aux_in=aux), 2 * M * F * D),
    "fwd qkv": (lambda: vf.gemm(x768, Wqkv, M=M, N=3 * D, K=D, out=o2304, bias=b2304), 2 * M * 3 * D * D),
}
def t(fn):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / 5
for rnd in range(2):
    for delay in (0, 1, 2, 3, 4, 6):
        _cabi.set_option("gemm_debug", 0 if delay == 0 else (16 | (delay << 8)))
        print(f"delay {delay}: " + " | ".join(f"{n}: {t(fn)*1e3:6.1f} us" for n, (fn, fl) in cases.items()), flush=True)
_cabi.set_option("gemm_debug", 0)
