#!/usr/bin/env python3
"""Clock and package power of each kernel class of the step, run alone in a loop for a few seconds (tools, not product):
which kernels pull the chip below its 2.4 GHz and to its 1 400 W cap?  rocm-smi is sampled from a thread while the loop runs.
    python tools/kernel_power.py [--seconds 5]"""
import argparse, os, re, subprocess, sys, threading, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import vit_amd.functional as vf
from vit_amd._cabi import ACT_GELU_GRAD

ap = argparse.ArgumentParser(); ap.add_argument("--seconds", type=float, default=5.0); a = ap.parse_args()
dev = torch.device("cuda:0")
B, H, T, dh = 256, 12, 197, 64
M, D, F = B * T, 768, 3072
g = torch.Generator().manual_seed(0)
R = lambda *s: (torch.randn(s, generator=g) * 0.5).to(dev).to(torch.bfloat16)
x768, x3072, x2304 = R(M, D), R(M, F), R(M, 3 * D)
Wqkv, W1 = R(3 * D, D), R(F, D)
b2304, b3072 = torch.randn(3 * D, device=dev), torch.randn(F, device=dev)
o2304, o3072, aux = torch.empty(M, 3 * D, device=dev, dtype=torch.bfloat16), torch.empty(M, F, device=dev, dtype=torch.bfloat16), torch.empty(M, F, device=dev, dtype=torch.bfloat16)
dW2 = torch.empty(D, F, device=dev)
ctx, lo, lse = torch.empty(M, D, device=dev, dtype=torch.bfloat16), torch.empty(M, D, device=dev, dtype=torch.bfloat16), torch.empty(B * H, T, device=dev)
dqkv, delta = torch.empty_like(x2304), torch.empty(B * H, T, device=dev)
xf, xs = torch.randn(M, D, device=dev), torch.empty(M, D, device=dev)
gam, bet = torch.ones(D, device=dev), torch.zeros(D, device=dev)
h16, mean, rstd = torch.empty(M, D, device=dev, dtype=torch.bfloat16), torch.empty(M, device=dev), torch.empty(M, device=dev)
n = 85_250_000
p, gr, m1, v1, sh = (torch.randn(n, device=dev) * 0.01 for _ in range(1)).__next__(), torch.randn(n, device=dev) * 0.01, torch.zeros(n, device=dev), torch.zeros(n, device=dev), torch.empty(n, device=dev, dtype=torch.bfloat16)
sc, drop = dh ** -0.5, (0.1, 1, 2)
vf.attention_fwd(x2304, B, H, T, dh, sc, dropout=drop, ctx=ctx, lse=lse, ctx_lo=lo)
cases = {
    "dW GEMM (fc2)": (lambda: vf.gemm(x768, x3072, M=D, N=F, K=M, a_trans=True, b_trans=True, out=dW2, split_k=-1), 2.0 * M * F * D),
    "QKV GEMM": (lambda: vf.gemm(x768, Wqkv, M=M, N=3 * D, K=D, out=o2304, bias=b2304), 2.0 * M * 3 * D * D),
    "FC1+GELU GEMM": (lambda: vf.gemm(x768, W1, M=M, N=F, K=D, out=o3072, bias=b3072, act=ACT_GELU_GRAD, aux_out=aux), 2.0 * M * F * D),
    "attention fwd": (lambda: vf.attention_fwd(x2304, B, H, T, dh, sc, dropout=drop, ctx=ctx, lse=lse, ctx_lo=lo), 4.0 * B * H * T * T * dh),
    "attention bwd": (lambda: vf.attention_bwd(x2304, ctx, x768, lse, B, H, T, dh, sc, dropout=drop, dqkv=dqkv, delta=delta, ctx_lo=lo), 10.0 * B * H * T * T * dh),
    "LayerNorm fwd+res": (lambda: vf.layernorm_fwd_residual(xf, x768, xs, gam, bet, 1e-12, out=h16, mean=mean, rstd=rstd), 0.0),
    "AdamW": (lambda: vf.adamw_step(p, gr, m1, v1, sh, lr=1e-3, beta1=0.9, beta2=0.999, eps=1e-8, weight_decay=0.0, step=3, sqnorm=None, max_norm=0.0), 0.0),
}
samples = []
def sampler(stop):
    while not stop[0]:
        t = subprocess.run(["rocm-smi", "--showclocks", "--showpower"], capture_output=True, text=True).stdout
        s = re.findall(r"sclk clock level: \d+: \((\d+)Mhz\)", t); w = re.findall(r"Package Power \(W\): ([0-9.]+)", t)
        if s and w: samples.append((time.time(), int(s[0]), float(w[0])))
        time.sleep(0.5)
for name, (fn, flop) in cases.items():
    fn(); torch.cuda.synchronize()
    samples.clear(); stop = [False]; th = threading.Thread(target=sampler, args=(stop,)); th.start()
    t0 = time.time(); e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    cnt = 0; e0.record()
    while time.time() - t0 < a.seconds:
        for _ in range(50): fn()
        cnt += 50; torch.cuda.synchronize()
    e1.record(); torch.cuda.synchronize(); stop[0] = True; th.join()
    us = e0.elapsed_time(e1) / cnt * 1e3
    ss = [s for s in samples if s[0] - t0 > 1.5]
    clk = sum(s[1] for s in ss) / max(1, len(ss)); pw = sum(s[2] for s in ss) / max(1, len(ss))
    print(f"{name:20s} {us:8.1f} us  {flop / us / 1e6 if flop else 0:7.1f} TFLOP/s   sclk {clk:5.0f} MHz   power {pw:6.0f} W   ({len(ss)} samples)", flush=True)
    time.sleep(1.0)
