"""Build libvit_amd.so (the C-ABI library of include/vit_amd.h) for gfx950 with hipcc, in-tree.

    python -m vit_amd.build [--force]

Objects go to vit_amd/csrc/_build/, the library to vit_amd/lib/libvit_amd.so (both git-ignored; the .so travels to the
GPU box with the repo snapshot).  The link step names torch's bundled libamdhip64.so (it has no SONAME), so that inside
a Python process the library binds to the SAME HIP runtime that owns torch's device pointers and streams instead of
pulling a second runtime from /opt/rocm.  hipcc cross-compiles without a GPU.
"""
from __future__ import annotations

import hashlib
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
CSRC = os.path.join(HERE, "csrc")
OBJ = os.path.join(CSRC, "_build")
LIBDIR = os.path.join(HERE, "lib")
LIB = os.path.join(LIBDIR, "libvit_amd.so")
ARCH = "gfx950"
SOURCES = ["api.hip", "gemm.hip", "gemm2.hip", "layernorm.hip", "attention.hip", "elementwise.hip"]
HEADERS = [os.path.join(CSRC, "common.h"), os.path.join(ROOT, "include", "vit_amd.h")]


def _hipcc() -> str:
    for c in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if c and (os.path.isabs(c) and os.path.exists(c) or not os.path.isabs(c)):
            return c
    raise RuntimeError("hipcc not found")


def _stamp(paths) -> str:
    h = hashlib.sha256()
    for p in paths:
        with open(p, "rb") as f:
            h.update(f.read())
    return h.hexdigest()


def source_stamp() -> str:
    """sha256 over the library's sources (what `build()` compares against to decide whether to rebuild): profile summaries
    record it (tools/pmc_traffic.py, tools/pmc_kernels.py) so that a reader can tell which kernels a counter set belongs to."""
    return _stamp([os.path.join(CSRC, s) for s in SOURCES] + HEADERS + [os.path.abspath(__file__)])


def _torch_libdir() -> str:
    import torch

    return os.path.join(os.path.dirname(torch.__file__), "lib")


def build(force: bool = False, verbose: bool = True, diag: bool = False, stamps: int = 0, defs=(), tag: str = "") -> str:
    """`diag=True` builds the DIAGNOSTIC twin libvit_amd_diag.so (-DVIT_PP_DIAG: the ping-pong GEMM can switch off its
    operand DMA / fragment reads / MFMAs / epilogue for timing, tools/pp_diag.py); it is loaded only when VIT_AMD_LIB names
    it and is never what the package or the tests use."""
    tag = ("_" + tag) if tag else (f"_stamp{stamps}" if stamps else ("_diag" if diag else ""))
    OBJ = os.path.join(CSRC, "_build" + tag)
    LIB = os.path.join(LIBDIR, f"libvit_amd{tag}.so")
    os.makedirs(OBJ, exist_ok=True)
    os.makedirs(LIBDIR, exist_ok=True)
    srcs = [os.path.join(CSRC, s) for s in SOURCES]
    stamp_file = os.path.join(OBJ, "stamp.txt")
    stamp = _stamp(srcs + HEADERS + [os.path.abspath(__file__)])
    if not force and os.path.exists(LIB) and os.path.exists(stamp_file) and open(stamp_file).read() == stamp:
        return LIB
    hipcc = _hipcc()
    flags = [f"--offload-arch={ARCH}", "-O3", "-fPIC", "-std=c++17", "-Wno-comment", "-Wno-inline-asm", "-DNDEBUG"]
    if stamps:  # in-kernel s_memtime stamps in the ping-pong GEMM (1 = per segment, 2 = also inside the LOAD segment)
        flags.append(f"-DVIT_PP_STAMP={int(stamps)}")
    elif diag:
        flags.append("-DVIT_PP_DIAG")
    flags.extend(defs)  # experiment builds: --defs "-DX -DY=1" --tag name -> libvit_amd_name.so

    def cc(src):
        obj = os.path.join(OBJ, os.path.basename(src).replace(".hip", ".o"))
        cmd = [hipcc, *flags, "-c", src, "-o", obj]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"hipcc failed on {src}:\n{r.stdout}\n{r.stderr}")
        return obj

    with ThreadPoolExecutor(max_workers=min(len(srcs), os.cpu_count() or 4)) as ex:
        objs = list(ex.map(cc, srcs))
    tl = _torch_libdir()
    link = ["g++", "-shared", "-o", LIB, *objs, f"-L{tl}", "-lamdhip64", "-Wl,--no-undefined", "-lstdc++", "-lm",
            "-Wl,-rpath," + tl, "-Wl,-rpath,/opt/rocm/lib"]
    r = subprocess.run(link, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError(f"link failed:\n{r.stdout}\n{r.stderr}")
    with open(stamp_file, "w") as f:
        f.write(stamp)
    if verbose:
        print(f"[vit_amd.build] built {LIB}")
    return LIB


if __name__ == "__main__":
    build(force="--force" in sys.argv, diag="--diag" in sys.argv,
          stamps=int(sys.argv[sys.argv.index("--stamps") + 1]) if "--stamps" in sys.argv else 0,
          defs=sys.argv[sys.argv.index("--defs") + 1].split() if "--defs" in sys.argv else (),
          tag=sys.argv[sys.argv.index("--tag") + 1] if "--tag" in sys.argv else "")
