#!/usr/bin/env python3
"""In-kernel stamps of the pair-pipelined attention backward (build: python -m vit_amd.build --defs -DVIT_PIPE_STAMP --tag pst;
run with VIT_AMD_LIB=vit_amd/lib/libvit_amd_pst.so).  Prints, per wave of workgroup 0, the cycles per iteration spent in
each section: top (B + head-end epilogue), DMA issue, A, counted wait, D, barrier.  Shares, not lengths (a stamp costs ~40+)."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import vit_amd.functional as vf
from vit_amd import _cabi
dev = torch.device("cuda:0")
B, H, T, dh = 256, 12, 197, 64
M, D = B * T, H * dh
qkv = (torch.randn(M, 3 * D, device=dev) * 0.5).to(torch.bfloat16)
dctx = (torch.randn(M, D, device=dev) * 0.5).to(torch.bfloat16)
ctx = torch.empty(M, D, device=dev, dtype=torch.bfloat16); lo = torch.empty_like(ctx)
lse = torch.empty(B * H, T, device=dev); dqkv = torch.empty_like(qkv); delta = torch.empty(B * H, T, device=dev)
cs = torch.empty(3 * D, device=dev)
dp = (0.1, 1, 2)
vf.attention_fwd(qkv, B, H, T, dh, dh ** -0.5, dropout=dp, ctx=ctx, lse=lse, ctx_lo=lo)
_cabi.set_option("attn_bwd_fused", 4)
for _ in range(3):
    vf.attention_bwd(qkv, ctx, dctx, lse, B, H, T, dh, dh ** -0.5, dropout=dp, dqkv=dqkv, delta=delta, colsum_out=cs, ctx_lo=lo)
torch.cuda.synchronize()
lib = _cabi.load()
buf = (ctypes.c_ulonglong * 64)()
lib.vit_debug_pipe_stamps.argtypes = [ctypes.c_void_p]
assert lib.vit_debug_pipe_stamps(buf) == 0
names = ["E", "issue", "A", "wait", "D", "barrier", "B"]
print("cycles per iteration (workgroup 0):  " + "  ".join(f"{n:>9s}" for n in names) + "      total")
for w in range(8):
    it = buf[w * 8 + 7]
    v = [buf[w * 8 + k] / it for k in range(7)]
    print(f"wave {w}:                              " + "  ".join(f"{x:9.0f}" for x in v) + f"  {sum(v):9.0f}")
