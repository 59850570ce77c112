#!/usr/bin/env python3
"""Randomised correctness sweep of the ping-pong GEMM core over 256-aligned shapes: every operand layout, bf16 / f32 outputs,
split-K, bias, and K from one K-tile up (the stream cursor runs past the end of short walks).  Compares with an fp32 torch
product of the same bf16 operands.  python tools/gemm_sweep.py [--n 400] [--seed 0]"""
import argparse, os, random, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import vit_amd.functional as vf
from vit_amd import _cabi

ap = argparse.ArgumentParser()
ap.add_argument("--n", type=int, default=400)
ap.add_argument("--seed", type=int, default=0)
a = ap.parse_args()
rnd = random.Random(a.seed)
dev = torch.device("cuda:0")
lib = _cabi.load()
worst, bad, kernels = 0.0, [], {}
for it in range(a.n):
    M = 256 * rnd.choice([1, 2, 3, 4, 5, 8, 13, 40, 197])
    N = 256 * rnd.choice([1, 2, 3, 4, 9, 12])
    K = 64 * rnd.choice([1, 2, 3, 4, 5, 7, 12, 13, 36, 48])
    at, bt = rnd.random() < 0.5, rnd.random() < 0.5
    kind = rnd.choice(["f32", "bf16", "split", "bias"])
    if M * N > 256 * 197 * 3072 // 2 and kind == "f32":
        kind = "bf16"
    g = torch.Generator(device="cpu").manual_seed(it)
    A = (torch.randn((M, K), generator=g) * 0.5).to(dev).to(torch.bfloat16)
    B = (torch.randn((N, K), generator=g) * 0.5).to(dev).to(torch.bfloat16)
    ref = A.float() @ B.float().t()
    As = A.t().contiguous() if at else A
    Bs = B.t().contiguous() if bt else B
    kw = dict(M=M, N=N, K=K, a_trans=at, b_trans=bt)
    if kind == "f32":
        out = vf.gemm(As, Bs, out_dtype=torch.float32, **kw); tol = 2e-5
    elif kind == "bf16":
        out = vf.gemm(As, Bs, out_dtype=torch.bfloat16, **kw).float(); tol = 4e-3
    elif kind == "split":
        out = vf.gemm(As, Bs, out_dtype=torch.float32, split_k=-1, **kw); tol = 2e-5
    else:
        bias = torch.randn(N, generator=g).to(dev)
        out = vf.gemm(As, Bs, out_dtype=torch.bfloat16, bias=bias, **kw).float(); ref = ref + bias; tol = 4e-3
    name = lib.vit_last_gemm_kernel().decode()
    kernels[name] = kernels.get(name, 0) + 1
    err = float((out - ref).norm() / ref.norm())
    worst = max(worst, err / tol)
    if not (err < tol):
        bad.append((M, N, K, at, bt, kind, name, err))
print("kernels:", kernels)
print(f"{a.n} cases, worst error / tolerance = {worst:.3f}, failures: {len(bad)}")
for b in bad[:20]:
    print("  FAIL", b)
sys.exit(1 if bad else 0)
