#!/usr/bin/env python3
"""In-kernel stamps of the resident attention forward (build: python -m vit_amd.build --defs -DVIT_FWD_STAMP --tag fst; run with
VIT_AMD_LIB=vit_amd/lib/libvit_amd_fst.so): the mean cycles a WAVE spends in each section of its workgroup's lifetime --
K / V staging, barrier, Q fragments, key loop, normalise + stores.  usage: python tools/fwd_stamps.py [B H T]"""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import vit_amd.functional as vf
from vit_amd import _cabi
dev = torch.device("cuda:0")
B, H, T = (int(v) for v in sys.argv[1:4]) if len(sys.argv) > 3 else (256, 12, 197)
dh = 64
M, D = B * T, H * dh
qkv = (torch.randn(M, 3 * D, device=dev) * 0.5).to(torch.bfloat16)
ctx = torch.empty(M, D, device=dev, dtype=torch.bfloat16); lo = torch.empty_like(ctx)
lse = torch.empty(B * H, T, device=dev)
dp = (0.1, 1, 2)
f = lambda: vf.attention_fwd(qkv, B, H, T, dh, dh ** -0.5, dropout=dp, ctx=ctx, lse=lse, ctx_lo=lo)
for _ in range(3): f()
torch.cuda.synchronize()
lib = _cabi.load()
lib.vit_debug_fwd_stamps.argtypes = [ctypes.c_void_p, ctypes.c_int]
import numpy as np
NW = 1 << 16
buf = np.zeros(NW * 8, dtype=np.uint64)
assert lib.vit_debug_fwd_stamps(None, 1) == 0
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record(); f(); e1.record(); torch.cuda.synchronize()
assert lib.vit_debug_fwd_stamps(buf.ctypes.data_as(ctypes.c_void_p), 0) == 0
r = buf.reshape(NW, 8)
r = r[r[:, 7] == 1].astype(np.float64)
names = ["K/V staged", "barrier", "Q frags", "key loop", "normalise+stores"]
print(f"B={B} H={H} T={T}: {e0.elapsed_time(e1) * 1e3:.1f} us (stamped build); waves recorded: {len(r)}")
tot = r[:, :5].sum(1)
for k, nm in enumerate(names):
    c = r[:, k]
    print(f"  {nm:18s} mean {c.mean():8.0f}  p10 {np.percentile(c, 10):8.0f}  median {np.median(c):8.0f}  p90 {np.percentile(c, 90):8.0f} cycles ({100 * c.sum() / tot.sum():4.1f} %)")
print(f"  {'wave lifetime':18s} mean {tot.mean():8.0f}  p10 {np.percentile(tot, 10):8.0f}  median {np.median(tot):8.0f}  p90 {np.percentile(tot, 90):8.0f}")
span = r[:, 6].max() - r[:, 5].min()
print(f"  first start -> last end: {span:.0f} ticks; sum of wave lifetimes / (1024 SIMDs x span) = {tot.sum() / (1024 * span):.2f} waves resident per SIMD")
