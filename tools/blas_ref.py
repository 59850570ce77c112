#!/usr/bin/env python3
"""Calibration only (never used by the product path): what torch.matmul (hipBLASLt / rocBLAS) reaches on the ViT-B GEMM
shapes on this GPU, bf16, random data -- the practical ceiling to compare tools/gemm_bench.py against."""
import torch
dev = torch.device("cuda:0")
M, D, F = 50432, 768, 3072
g = torch.Generator(device="cpu").manual_seed(0)
R = lambda *s: (torch.randn(s, generator=g) * 0.5).to(dev).to(torch.bfloat16)
def t(fn):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / 10
x768, x3072, x2304 = R(M, D), R(M, F), R(M, 3 * D)
Wqkv, Wo, W1, W2 = R(3 * D, D), R(D, D), R(F, D), R(D, F)
cases = {
    "NT qkv  [M,768]x[2304,768]^T": (lambda: x768 @ Wqkv.t(), 2 * M * 3 * D * D),
    "NT out  [M,768]x[768,768]^T": (lambda: x768 @ Wo.t(), 2 * M * D * D),
    "NT fc1  [M,768]x[3072,768]^T": (lambda: x768 @ W1.t(), 2 * M * F * D),
    "NT fc2  [M,3072]x[768,3072]^T": (lambda: x3072 @ W2.t(), 2 * M * F * D),
    "NN dXfc2 [M,768]x[768,3072]": (lambda: x768 @ W2, 2 * M * F * D),
    "NN dXfc1 [M,3072]x[3072,768]": (lambda: x3072 @ W1, 2 * M * F * D),
    "NN dXqkv [M,2304]x[2304,768]": (lambda: x2304 @ Wqkv, 2 * M * 3 * D * D),
    "TN dWfc2 [M,768]^T x [M,3072]": (lambda: x768.t() @ x3072, 2 * M * F * D),
    "TN dWfc1 [M,3072]^T x [M,768]": (lambda: x3072.t() @ x768, 2 * M * F * D),
    "TN dWqkv [M,2304]^T x [M,768]": (lambda: x2304.t() @ x768, 2 * M * 3 * D * D),
}
for name, (fn, fl) in cases.items():
    ms = t(fn)
    print(f"{name:34s} {ms*1e3:8.1f} us  {fl/(ms*1e-3)/1e12:7.0f} TF")
