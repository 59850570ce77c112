#!/usr/bin/env python3
"""The fused QKV projection at the ViT-B shape with pos_encoding_type 'rope': rotation in the GEMM epilogue (vit_gemm with the
rope fields, ping-pong core) against product + vit_rope_qk pass, interleaved in one process."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import vit_amd.functional as vf

dev = torch.device("cuda:0")
M, D, H, dh, T = 50432, 768, 12, 64, 197
x = (torch.randn(M, D, device=dev) * 0.5).to(torch.bfloat16)
W = (torch.randn(3 * D, D, device=dev) * 0.05).to(torch.bfloat16)
b = torch.randn(3 * D, device=dev)
inv = 1.0 / (10000.0 ** (torch.arange(0, dh, 2).float() / dh))
fr = torch.outer(torch.arange(T).float(), inv)
cos, sin = fr.cos().contiguous().to(dev), fr.sin().contiguous().to(dev)
out = torch.empty(M, 3 * D, device=dev, dtype=torch.bfloat16)
plain = lambda: vf.gemm(x, W, M=M, N=3 * D, K=D, out=out, bias=b)
fused = lambda: vf.gemm(x, W, M=M, N=3 * D, K=D, out=out, bias=b, rope=(cos, sin, T, dh, 2 * D))
def two():
    vf.gemm(x, W, M=M, N=3 * D, K=D, out=out, bias=b)
    vf.rope_qk(out, cos, sin, T, H, dh)
def t(fn):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / 5 * 1e3
res = {"plain": [], "fused": [], "two": []}
for r in range(5):
    for k, f in (("plain", plain), ("fused", fused), ("two", two)):
        v = t(f)
        if r:
            res[k].append(v)
med = {k: sorted(v)[len(v) // 2] for k, v in res.items()}
print(f"QKV projection [50432, 768] x [2304, 768]^T + bias: plain {med['plain']:.1f} us; + rope in the epilogue {med['fused']:.1f} us; "
      f"product then vit_rope_qk {med['two']:.1f} us")
