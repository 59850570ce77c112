#!/usr/bin/env python3
"""Where does the ping-pong GEMM (gemm3_kernel) spend a barrier interval?  Runs the GEMM shapes of a ViT-B layer on the
DIAGNOSTIC twin of the library (`python -m vit_amd.build --diag`, loaded through VIT_AMD_LIB) with pieces switched off:
  16 no operand DMA (LDS-DMA issue; the counted vmcnt waits stay)   32 no fragment reads (ds_read_b128 / tr reads)
  64 no MFMAs                                                      128 no epilogue
Results are meaningless with any bit set; only times are read.  A piece's cost is bounded from both sides: the time that
disappears when it alone is removed, and the time of the kernel that has only that piece left."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
if len(sys.argv) > 2:  # pp_diag.py epi <tag>: a compile-time variant build (python -m vit_amd.build --defs ... --tag <tag>)
    os.environ["VIT_AMD_LIB"] = os.path.join(ROOT, "vit_amd", "lib", f"libvit_amd_{sys.argv[2]}.so" if sys.argv[2] != "prod" else "libvit_amd.so")
os.environ.setdefault("VIT_AMD_LIB", os.path.join(ROOT, "vit_amd", "lib", "libvit_amd_diag.so"))
import torch
import vit_amd.functional as vf
from vit_amd import _cabi

dev = torch.device("cuda:0")
M, D, F = 50432, 768, 3072
g = torch.Generator(device="cpu").manual_seed(0)
R = lambda *s: (torch.randn(s, generator=g) * 0.5).to(dev).to(torch.bfloat16)
x768, x3072, dy768, Wqkv, W1 = R(M, D), R(M, F), R(M, D), R(3 * D, D), R(F, D)
bias = torch.zeros(3 * D, device=dev)
o2304 = torch.empty(M, 3 * D, device=dev, dtype=torch.bfloat16)
o768 = torch.empty(M, D, device=dev, dtype=torch.bfloat16)
dWfc = torch.empty(D, F, device=dev)
dWo = torch.empty(D, D, device=dev)
cases = {
    "NT qkv fwd   N=2304 K=768 ": (lambda: vf.gemm(x768, Wqkv, M=M, N=3 * D, K=D, bias=bias, out=o2304), 2 * M * 3 * D * D),
    "NN dX fc1    N=768 K=3072 ": (lambda: vf.gemm(x3072, W1, M=M, N=D, K=F, b_trans=True, out=o768), 2 * M * F * D),
    "TT dW fc2    768x3072     ": (lambda: vf.gemm(dy768, x3072, M=D, N=F, K=M, a_trans=True, b_trans=True, out=dWfc, split_k=-1), 2 * M * F * D),
    "TT dW out    768x768      ": (lambda: vf.gemm(dy768, x768, M=D, N=D, K=M, a_trans=True, b_trans=True, out=dWo, split_k=-1), 2 * M * D * D),
}


def t(fn, n=30):
    for _ in range(30):  # clocks up (a cold GPU runs the first milliseconds at idle clocks)
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


if len(sys.argv) > 1 and sys.argv[1] == "epi":
    masks = [(0, sys.argv[2] if len(sys.argv) > 2 else "full")]
else:
  masks = [(0, "full"), (16, "-dma"), (32, "-reads"), (64, "-mfma"), (128, "-epi"), (16 | 32, "mfma+epi"), (32 | 64, "dma+epi"),
         (16 | 64, "reads+epi"), (16 | 32 | 64, "barriers+epi"), (16 | 32 | 64 | 128, "barriers")]
W2 = R(D, F)
o3072 = torch.empty(M, F, device=dev, dtype=torch.bfloat16)
aux = torch.empty(M, F, device=dev, dtype=torch.bfloat16)
b3072 = torch.zeros(F, device=dev)
cases["NT fc1 fwd +gelu,aux N=3072"] = (lambda: vf.gemm(x768, W1, M=M, N=F, K=D, bias=b3072, act=_cabi.ACT_GELU_GRAD, aux_out=aux, out=o3072), 2 * M * F * D)
cases["NN dX fc2 *aux N=3072 K=768"] = (lambda: vf.gemm(dy768, W2, M=M, N=F, K=D, b_trans=True, act=_cabi.ACT_MUL_AUX, aux_in=aux, out=o3072), 2 * M * F * D)
print("kernel:", end=" ")
for name, (fn, fl) in cases.items():
    _cabi.load().vit_debug_pp_diag(0)
    fn()
    print(_cabi.load().vit_last_gemm_kernel().decode(), end="; ")
print()
for name, (fn, fl) in cases.items():
    row = []
    for m, label in masks:
        _cabi.load().vit_debug_pp_diag(m)
        us = t(fn)
        row.append(f"{label} {us:6.1f}")
    full = float(row[0].split()[-1])
    print(f"{name} [{fl / full / 1e6:5.0f} TF] " + " | ".join(row), flush=True)
_cabi.load().vit_debug_pp_diag(0)
