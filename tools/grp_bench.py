#!/usr/bin/env python3
"""A/B of the two-N-group tile order (vit_set_option gemm_ngroups) on the N = 3072 GEMMs of a ViT-B layer."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import vit_amd.functional as vf
from vit_amd import _cabi
dev = torch.device("cuda:0")
M, D, F = 50432, 768, 3072
g = torch.Generator().manual_seed(0)
R = lambda *s: (torch.randn(s, generator=g) * 0.5).to(dev).to(torch.bfloat16)
x, dy, W1, W2 = R(M, D), R(M, D), R(F, D), R(D, F)
b1 = torch.randn(F, device=dev)
out, aux = torch.empty(M, F, device=dev, dtype=torch.bfloat16), torch.empty(M, F, device=dev, dtype=torch.bfloat16)
dU = torch.empty(M, F, device=dev, dtype=torch.bfloat16); cs = torch.empty(F, device=dev)
def t(fn, n=20):
    fn(); fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
fc1 = lambda: vf.gemm(x, W1, M=M, N=F, K=D, out=out, bias=b1, act=vf.ACT_GELU_GRAD, aux_out=aux)
dxa = lambda: vf.gemm(dy, W2, M=M, N=F, K=D, b_trans=True, out=dU, act=vf.ACT_MUL_AUX, aux_in=aux, colsum_out=cs)
ref = {}
for rnd in range(2):
    for grp in (0, 1):
        _cabi.set_option("gemm_ngroups", grp)
        a, b = t(fc1), t(dxa)
        key = (out.clone(), aux.clone(), dU.clone(), cs.clone())
        if not ref: ref = key
        same = all(torch.equal(u, v) for u, v in zip(ref, key))
        fl = 2.0 * M * F * D
        print(f"ngroups={grp}: fc1+gelu' {a:6.1f} us ({fl / a / 1e6:.0f} TF)  dX*aux {b:6.1f} us ({fl / b / 1e6:.0f} TF)  identical={same}", flush=True)
_cabi.set_option("gemm_ngroups", 1)
