#!/usr/bin/env python3
"""In-kernel stamps of the ping-pong GEMM (cdna_hip_programming.md, "In-kernel stamps"): where does a barrier interval
of gemm3_kernel go?  Runs on the stamp builds of the library (`python -m vit_amd.build --stamps 1|2`):
    python tools/pp_stamps.py 1      # 4 stamps per phase: LOAD segment | barrier + lgkmcnt | MFMA segment | closing barrier
    python tools/pp_stamps.py 2      # + inside the LOAD segment: fragment reads landed | LDS-DMA issue | counted vmcnt wait
Read SHARES, not lengths: a stamp costs ~40 cycles and its lgkmcnt(0) fences forbid overlaps the real kernel has."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
level = int(sys.argv[1]) if len(sys.argv) > 1 else 1
tag = sys.argv[2] if len(sys.argv) > 2 else f"stamp{level}"
os.environ["VIT_AMD_LIB"] = os.path.join(ROOT, "vit_amd", "lib", f"libvit_amd_{tag}.so")
import ctypes
import torch
import vit_amd.functional as vf
from vit_amd import _cabi

lib = _cabi.load()
lib.vit_debug_pp_stamps.argtypes = [ctypes.c_void_p, ctypes.c_int]
dev = torch.device("cuda:0")
M, D, F = 50432, 768, 3072
g = torch.Generator(device="cpu").manual_seed(0)
R = lambda *s: (torch.randn(s, generator=g) * 0.5).to(dev).to(torch.bfloat16)
x768, x3072, dy768, Wqkv, W1 = R(M, D), R(M, F), R(M, D), R(3 * D, D), R(F, D)
bias = torch.zeros(3 * D, device=dev)
o2304 = torch.empty(M, 3 * D, device=dev, dtype=torch.bfloat16)
o768 = torch.empty(M, D, device=dev, dtype=torch.bfloat16)
dWfc = torch.empty(D, F, device=dev)
cases = {
    "NT qkv fwd N=2304 K=768": lambda: vf.gemm(x768, Wqkv, M=M, N=3 * D, K=D, bias=bias, out=o2304),
    "NN dX fc1  N=768 K=3072": lambda: vf.gemm(x3072, W1, M=M, N=D, K=F, b_trans=True, out=o768),
    "TT dW fc2  768x3072    ": lambda: vf.gemm(dy768, x3072, M=D, N=F, K=M, a_trans=True, b_trans=True, out=dWfc, split_k=-1),
}
names3 = ["reads landed P1 (12)", "P2 (4)", "P3 (8)", "P4 (0)", "DMA issue + vmcnt", "barrier+lgkm", "MFMA seg + closing barrier + epilogue"]
names4 = ["LOAD segment P1 (+loop tail)", "P2", "P3", "P4", "barrier+lgkm", "MFMA segment", "closing barrier + epilogue"]
names5 = ["P1 LOAD segment incl. loop tail (per K-tile)", "everything else (per K-tile)", "", "", "", "", ""]
names = names5 if level == 5 else names4 if level == 4 else names3 if level == 3 else ["reads landed", "DMA issue", "vmcnt wait" if level >= 2 else "LOAD segment", "barrier+lgkm", "MFMA segment",
         "closing barrier", "epilogue"]
buf = torch.zeros(64, dtype=torch.int64, device=dev)
for name, fn in cases.items():
    for _ in range(20):
        fn()
    torch.cuda.synchronize()
    for block in (0, 100):
        buf.zero_()
        lib.vit_debug_pp_stamps(ctypes.c_void_p(buf.data_ptr()), block)
        fn()
        torch.cuda.synchronize()
        lib.vit_debug_pp_stamps(None, 0)
        b = buf.cpu().view(8, 8)
        for w in (0, 4):
            tot = float(b[w, :7].sum())
            nph = max(1, int(b[w, 7]))
            per = (lambda k: nph / 4 if (level in (3, 4) and k < 4) or level == 5 else nph)
            parts = " | ".join(f"{names[k]} {float(b[w, k]) / per(k):6.1f} ({100 * float(b[w, k]) / tot:4.1f}%)" for k in range(7)
                              if (level >= 2 or k >= 2) and names[k])
            print(f"{name} wg {block:3d} wave {w}: {nph:4d} phases, {tot / nph:6.1f} ticks/phase: {parts}", flush=True)
