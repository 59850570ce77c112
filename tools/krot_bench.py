#!/usr/bin/env python3
"""A/B of the K-walk rotation (vit_set_option gemm_krot) on the GEMMs of a ViT-B layer (fwd, dX, dW)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import vit_amd.functional as vf
from vit_amd import _cabi
dev = torch.device("cuda:0")
M, D, F = 50432, 768, 3072
g = torch.Generator().manual_seed(0)
R = lambda *s: (torch.randn(s, generator=g) * 0.5).to(dev).to(torch.bfloat16)
x, dy, xF, W1, W2, Wq = R(M, D), R(M, D), R(M, F), R(F, D), R(D, F), R(3 * D, D)
x3 = R(M, 3 * D)
b1, bq, bD = torch.randn(F, device=dev), torch.randn(3 * D, device=dev), torch.randn(D, device=dev)
oF, aux, oD, oQ = (torch.empty(M, n, device=dev, dtype=torch.bfloat16) for n in (F, F, D, 3 * D))
dW2, dW1, dWo, dWq = (torch.empty(s, device=dev) for s in ((D, F), (F, D), (D, D), (3 * D, D)))
cs = torch.empty(F, device=dev)
def t(fn, n=20):
    fn(); fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
cases = {
    "fwd qkv": (lambda: vf.gemm(x, Wq, M=M, N=3 * D, K=D, out=oQ, bias=bq), 2.0 * M * 3 * D * D, oQ),
    "fwd fc1+gelu'": (lambda: vf.gemm(x, W1, M=M, N=F, K=D, out=oF, bias=b1, act=vf.ACT_GELU_GRAD, aux_out=aux), 2.0 * M * F * D, oF),
    "fwd fc2": (lambda: vf.gemm(xF, W2, M=M, N=D, K=F, out=oD, bias=bD, dropout=(0.1, 1, 2)), 2.0 * M * F * D, oD),
    "dX fc2*aux": (lambda: vf.gemm(dy, W2, M=M, N=F, K=D, b_trans=True, out=oF, act=vf.ACT_MUL_AUX, aux_in=aux, colsum_out=cs), 2.0 * M * F * D, oF),
    "dX fc1": (lambda: vf.gemm(xF, W1, M=M, N=D, K=F, b_trans=True, out=oD), 2.0 * M * F * D, oD),
    "dX qkv": (lambda: vf.gemm(x3, Wq, M=M, N=D, K=3 * D, b_trans=True, out=oD), 2.0 * M * 3 * D * D, oD),
    "dW fc2": (lambda: vf.gemm(dy, xF, M=D, N=F, K=M, a_trans=True, b_trans=True, out=dW2, split_k=-1), 2.0 * M * F * D, dW2),
    "dW fc1": (lambda: vf.gemm(xF, x, M=F, N=D, K=M, a_trans=True, b_trans=True, out=dW1, split_k=-1), 2.0 * M * F * D, dW1),
    "dW out": (lambda: vf.gemm(dy, x, M=D, N=D, K=M, a_trans=True, b_trans=True, out=dWo, split_k=-1), 2.0 * M * D * D, dWo),
    "dW qkv": (lambda: vf.gemm(x3, x, M=3 * D, N=D, K=M, a_trans=True, b_trans=True, out=dWq, split_k=-1), 2.0 * M * 3 * D * D, dWq),
}
modes = [int(a) for a in (sys.argv[1:] or ["0", "1"])]
for name, (fn, fl, out) in cases.items():
    row, ref = [], None
    for rnd in range(2):
        for kr in modes:
            _cabi.set_option("gemm_krot", kr)
            us = t(fn)
            if ref is None: ref = out.float().clone()
            err = float((out.float() - ref).norm() / ref.norm())
            row.append(f"krot={kr}: {us:6.1f} us ({fl / us / 1e6:5.0f} TF, d={err:.0e})")
    print(f"{name:14s} " + "  ".join(row), flush=True)
_cabi.set_option("gemm_krot", 0)
