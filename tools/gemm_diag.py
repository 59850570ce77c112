#!/usr/bin/env python3
"""Where does the LDS-DMA GEMM core spend its time?  Times a few ViT-B GEMMs with the operand DMA disabled (pure
LDS-read + MFMA loop) and with the MFMAs disabled (pure operand delivery), per geometry.  Diagnostics only."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import vit_amd.functional as vf
from vit_amd import _cabi

dev = torch.device("cuda:0")
M, D, F = 50432, 768, 3072
g = torch.Generator(device="cpu").manual_seed(0)
R = lambda *s: (torch.randn(s, generator=g) * 0.5).to(dev).to(torch.bfloat16)
x768, x3072, Wqkv, W1 = R(M, D), R(M, F), R(3 * D, D), R(F, D)
o2304 = torch.empty(M, 3 * D, device=dev, dtype=torch.bfloat16)
o768 = torch.empty(M, D, device=dev, dtype=torch.bfloat16)
dW = torch.empty(D, F, device=dev)
cases = {
    "NT qkv  K=768 N=2304": (lambda: vf.gemm(x768, Wqkv, M=M, N=3 * D, K=D, out=o2304), 2 * M * 3 * D * D),
    "NN dXfc1 K=3072 N=768": (lambda: vf.gemm(x3072, W1, M=M, N=D, K=F, b_trans=True, out=o768), 2 * M * F * D),
    "TN dWfc2 K=50432": (lambda: vf.gemm(x768, x3072, M=D, N=F, K=M, a_trans=True, b_trans=True, out=dW, split_k=-1), 2 * M * F * D),
}
def t(fn):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / 5
for core in (2,):
    _cabi.set_option("gemm_core", core)
    for name, (fn, fl) in cases.items():
        row = []
        for dbg in (0, 4, 8, 12):
            _cabi.set_option("gemm_debug", dbg)
            ms = t(fn)
            row.append(f"dbg{dbg}: {ms*1e3:7.1f} us ({fl/(ms*1e-3)/1e12:6.0f} TF-equiv)")
        print(f"core{core} {name:24s} " + " | ".join(row))
_cabi.set_option("gemm_debug", 0); _cabi.set_option("gemm_core", 1)
