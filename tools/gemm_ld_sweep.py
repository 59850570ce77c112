#!/usr/bin/env python3
"""Does the row stride of the operands matter (memory-channel camping)?  Same GEMMs, A (and B) stored with padded
leading dimensions.  Diagnostics only."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import vit_amd.functional as vf
from vit_amd import _cabi

dev = torch.device("cuda:0")
M, D, F = 50432, 768, 3072
g = torch.Generator(device="cpu").manual_seed(0)
def R(r, c, ld):
    t = torch.zeros((r, ld), dtype=torch.bfloat16, device=dev)
    t[:, :c] = (torch.randn((r, c), generator=g) * 0.5).to(dev).to(torch.bfloat16)
    return t
def t(fn):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / 5
for core in (0, 5):
    _cabi.set_option("gemm_core", core)
    for pad in (0, 32, 64, 128, 192, 256):
        # NT, K=768: qkv forward
        A = R(M, D, D + pad); W = R(3 * D, D, D + pad); out = torch.empty(M, 3 * D, device=dev, dtype=torch.bfloat16)
        t1 = t(lambda: vf.gemm(A, W, M=M, N=3 * D, K=D, lda=D + pad, ldb=D + pad, out=out))
        # NN, K=3072: dX of fc1 (A = dU [M,3072], B = W1 [3072,768] stored [K][N])
        A2 = R(M, F, F + pad); W1 = R(F, D, D + pad); o2 = torch.empty(M, D, device=dev, dtype=torch.bfloat16)
        t2 = t(lambda: vf.gemm(A2, W1, M=M, N=D, K=F, b_trans=True, lda=F + pad, ldb=D + pad, out=o2))
        # NT, K=3072: fc2 forward without epilogue extras
        W2 = R(D, F, F + pad)
        t3 = t(lambda: vf.gemm(A2, W2, M=M, N=D, K=F, lda=F + pad, ldb=F + pad, out=o2))
        f1, f2 = 2 * M * 3 * D * D, 2 * M * F * D
        print(f"core{core} pad {pad:4d}: NT qkv K768 {t1*1e3:7.1f} us {f1/t1/1e9:6.0f} TF | NN dXfc1 K3072 {t2*1e3:7.1f} us {f2/t2/1e9:6.0f} TF | NT fc2 K3072 {t3*1e3:7.1f} us {f2/t3/1e9:6.0f} TF")
        del A, W, A2, W1, W2
_cabi.set_option("gemm_core", 1)
