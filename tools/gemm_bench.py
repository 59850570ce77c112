#!/usr/bin/env python3
"""GEMM microbenchmark over the ViT-B/16 (C3) shapes: every GEMM of one layer's forward + backward, each GEMM core
(0 = generic 128x128 register-staged, 5 = the 256x256x64 ping-pong LDS-DMA core), interleaved rounds in ONE process, random data.
Usage: python tools/gemm_bench.py [--rounds 5] [--cores 0,2,3]"""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import vit_amd.functional as vf
from vit_amd import _cabi
from vit_amd._cabi import ACT_DGELU, ACT_GELU


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rounds", type=int, default=5)
    ap.add_argument("--cores", default="0,5")
    ap.add_argument("--M", type=int, default=50432)
    ap.add_argument("--only", default="", help="substring filter on the case names, e.g. 'fc1' or 'dX  fc2'")
    args = ap.parse_args()
    cores = [c for c in args.cores.split(",")]  # "5" or "5s8" / "5s10": ping-pong core with an 8- / 10-slot ring
    def select(c):  # "5", "5s10" (10-slot ring), "5b0" (always 256 workgroups), "5h0" (no half-tile tail launch)
        core, _, half = c.partition("h")
        core, _, bal = core.partition("b")
        core, _, slots = core.partition("s")
        _cabi.set_option("gemm_core", int(core))
        _cabi.set_option("gemm_pp_slots", 8)  # the 10-slot ring is gone
        _cabi.set_option("gemm_balance_wgs", int(bal) if bal else 1)
        _cabi.set_option("gemm_half_tail", int(half) if half else 0)
    dev = torch.device("cuda:0")
    M, D, F = args.M, 768, 3072
    g = torch.Generator(device="cpu").manual_seed(0)
    R = lambda *s: (torch.randn(s, generator=g) * 0.5).to(dev).to(torch.bfloat16)
    x768, x3072, x2304 = R(M, D), R(M, F), R(M, 3 * D)
    Wqkv, Wo, W1, W2 = R(3 * D, D), R(D, D), R(F, D), R(D, F)
    bias3072, bias768, bias2304 = torch.randn(F, device=dev), torch.randn(D, device=dev), torch.randn(3 * D, device=dev)
    res = torch.randn(M, D, device=dev)
    o768f, o768, o3072, o2304 = torch.empty(M, D, device=dev), torch.empty(M, D, device=dev, dtype=torch.bfloat16), \
        torch.empty(M, F, device=dev, dtype=torch.bfloat16), torch.empty(M, 3 * D, device=dev, dtype=torch.bfloat16)
    aux = torch.empty(M, F, device=dev, dtype=torch.bfloat16)
    dW1, dW2, dWo, dWqkv = (torch.empty(s, device=dev) for s in ((F, D), (D, F), (D, D), (3 * D, D)))
    drop = (0.1, 1, 2)
    cases = {
        "fwd qkv   [M,768]x[2304,768]^T +b": (lambda: vf.gemm(x768, Wqkv, M=M, N=3 * D, K=D, out=o2304, bias=bias2304), 2 * M * 3 * D * D),
        "fwd out   [M,768]x[768,768]^T +b,drop,res": (lambda: vf.gemm(x768, Wo, M=M, N=D, K=D, out=o768f, bias=bias768, dropout=drop, residual=res), 2 * M * D * D),
        "fwd fc1   [M,768]x[3072,768]^T +b,gelu": (lambda: vf.gemm(x768, W1, M=M, N=F, K=D, out=o3072, bias=bias3072, act=ACT_GELU, aux_out=aux), 2 * M * F * D),
        "fwd fc2   [M,3072]x[768,3072]^T +b,drop,res": (lambda: vf.gemm(x3072, W2, M=M, N=D, K=F, out=o768f, bias=bias768, dropout=drop, residual=res), 2 * M * F * D),
        "fwd out'  [M,768]x[768,768]^T +b,drop -> bf16": (lambda: vf.gemm(x768, Wo, M=M, N=D, K=D, out=o768, bias=bias768, dropout=drop), 2 * M * D * D),
        "fwd fc2'  [M,3072]x[768,3072]^T +b,drop -> bf16": (lambda: vf.gemm(x3072, W2, M=M, N=D, K=F, out=o768, bias=bias768, dropout=drop), 2 * M * F * D),
        "dX  fc2   [M,768]x[768,3072] *dgelu": (lambda: vf.gemm(x768, W2, M=M, N=F, K=D, b_trans=True, out=o3072, act=ACT_DGELU, aux_in=aux), 2 * M * F * D),
        "dX  fc1   [M,3072]x[3072,768]": (lambda: vf.gemm(x3072, W1, M=M, N=D, K=F, b_trans=True, out=o768), 2 * M * F * D),
        "dX  out   [M,768]x[768,768]": (lambda: vf.gemm(x768, Wo, M=M, N=D, K=D, b_trans=True, out=o768), 2 * M * D * D),
        "dX  qkv   [M,2304]x[2304,768]": (lambda: vf.gemm(x2304, Wqkv, M=M, N=D, K=3 * D, b_trans=True, out=o768), 2 * M * 3 * D * D),
        "dW  fc2   [M,768]^T x [M,3072]": (lambda: vf.gemm(x768, x3072, M=D, N=F, K=M, a_trans=True, b_trans=True, out=dW2, split_k=-1), 2 * M * F * D),
        "dW  fc1   [M,3072]^T x [M,768]": (lambda: vf.gemm(x3072, x768, M=F, N=D, K=M, a_trans=True, b_trans=True, out=dW1, split_k=-1), 2 * M * F * D),
        "dW  out   [M,768]^T x [M,768]": (lambda: vf.gemm(x768, x768, M=D, N=D, K=M, a_trans=True, b_trans=True, out=dWo, split_k=-1), 2 * M * D * D),
        "dW  qkv   [M,2304]^T x [M,768]": (lambda: vf.gemm(x2304, x768, M=3 * D, N=D, K=M, a_trans=True, b_trans=True, out=dWqkv, split_k=-1), 2 * M * 3 * D * D),
    }
    if args.only:
        cases = {k: v for k, v in cases.items() if any(o in k for o in args.only.split(","))}
    times = {k: {c: [] for c in cores} for k in cases}
    for r in range(args.rounds + 1):
        for name, (fn, fl) in cases.items():
            for c in cores:
                select(c)
                fn()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(3):
                    fn()
                e1.record()
                torch.cuda.synchronize()
                if r:
                    times[name][c].append(e0.elapsed_time(e1) / 3)
    _cabi.set_option("gemm_core", 1)
    _cabi.set_option("gemm_pp_slots", 8)
    _cabi.set_option("gemm_balance_wgs", 1)
    _cabi.set_option("gemm_half_tail", 0)
    tot = {c: 0.0 for c in cores}
    print(f"{'case':48s} " + " ".join(f"core{c}: us / TF".rjust(20) for c in cores))
    for name, (fn, fl) in cases.items():
        row = []
        for c in cores:
            t = sorted(times[name][c])[len(times[name][c]) // 2]
            tot[c] += t
            row.append(f"{t * 1e3:9.1f} / {fl / (t * 1e-3) / 1e12:6.0f}".rjust(20))
        print(f"{name:48s} " + " ".join(row))
    print(f"{'one layer, all 12 GEMMs (ms)':48s} " + " ".join(f"{tot[c]:20.3f}" for c in cores))


if __name__ == "__main__":
    main()
