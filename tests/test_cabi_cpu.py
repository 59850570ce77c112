"""CPU suite: the C-ABI library loads and exports every symbol include/vit_amd.h declares (no compute without a GPU)."""
import ctypes
import os

import pytest


def test_library_builds_and_exports_every_declared_symbol():
    from vit_amd import _cabi
    from vit_amd import build as vb

    lib_path = vb.build(force=False, verbose=False)
    assert os.path.exists(lib_path)
    lib = _cabi.load()
    declared = _cabi.declared_symbols()
    assert len(declared) >= 24
    for sym in declared:
        assert hasattr(lib, sym), f"{sym} declared in include/vit_amd.h but not exported"
        assert sym in _cabi._PROTOS, f"{sym} has no ctypes prototype"
    assert set(_cabi._PROTOS) == set(declared)
    assert lib.vit_version() == 100


def test_gemm_desc_layout_matches_header():
    """The ctypes mirror of vit_gemm_desc must have the C struct's size (x86-64 SysV layout)."""
    import subprocess
    import tempfile

    from vit_amd import _cabi

    src = '#include <stdio.h>\n#include "vit_amd.h"\nint main(){printf("%zu", sizeof(vit_gemm_desc));return 0;}\n'
    inc = os.path.dirname(_cabi.HEADER_PATH)
    with tempfile.TemporaryDirectory() as d:
        c = os.path.join(d, "s.c")
        open(c, "w").write(src)
        exe = os.path.join(d, "s")
        subprocess.run(["gcc", "-I", inc, c, "-o", exe], check=True)
        size = int(subprocess.run([exe], capture_output=True, text=True, check=True).stdout)
    assert ctypes.sizeof(_cabi.GemmDesc) == size


def test_error_path_without_gpu():
    """Argument errors are reported through the status code + vit_last_error, not by crashing."""
    from vit_amd import _cabi

    lib = _cabi.load()
    d = _cabi.GemmDesc()
    rc = lib.vit_gemm(None, ctypes.byref(d), None)
    assert rc == -1
    assert b"null operand" in lib.vit_last_error()
    rc = lib.vit_layernorm_fwd(None, None, None, None, None, 1, None, None, 4, 32, 1e-12, None)
    assert rc == -1


@pytest.fixture(scope="module")
def kernel_isa(tmp_path_factory):
    """ISA text of attention.hip and gemm2.hip as hipcc emits it for gfx950 (cross-compiled: no GPU needed)."""
    import shutil
    import subprocess

    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        pytest.skip("hipcc not available")
    root = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "vit_amd", "csrc")
    d = tmp_path_factory.mktemp("isa")
    out = {}
    for f in ("attention", "gemm2"):
        o = d / (f + ".s")
        subprocess.run([hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-S", "--cuda-device-only", os.path.join(root, f + ".hip"),
                        "-o", str(o)], check=True, capture_output=True, cwd=str(d))
        out[f] = o.read_text()
    return out


def _kernel_bodies(text, pattern):
    import re
    for name in re.findall(r"^(" + pattern + r"):", text, flags=re.M):
        a = text.index(name + ":")
        yield name, [l.strip() for l in text[a:text.index("s_endpgm", a)].split("\n")]


def test_no_drain_inside_the_dma_pipelines(kernel_isa):
    """r03's largest finding, pinned: no `s_waitcnt vmcnt(0)` and no scratch access may sit inside the loops that keep LDS-DMA in
    flight -- the K loop of every ping-pong GEMM instantiation (between its first and last MFMA) and the pair loop of the
    pair-pipelined attention backward (whose only vmcnt(0) is case 0 of its counted wait).  The compiler wrote such waits
    itself when the DMA went through its builtin, and writes one for every scratch reload."""
    n = 0
    for name, body in _kernel_bodies(kernel_isa["gemm2"], r"_ZN3vit12gemm3_kernelI\w+"):
        mf = [i for i, l in enumerate(body) if l.startswith("v_mfma")]
        loop = body[mf[0]:mf[-1] + 1]
        assert not [l for l in loop if "vmcnt(0)" in l], name
        assert not [l for l in loop if "scratch_" in l], name
        n += 1
    assert n >= 8, n
    m = 0
    for name, body in _kernel_bodies(kernel_isa["attention"], r"_ZN3vit20attn_bwd_pipe_kernelI\w+"):
        assert sum("vmcnt(0)" in l for l in body) <= 1, name
        assert not [l for l in body if "scratch_" in l], name
        m += 1
    assert m >= 2, m


def test_untracked_q_loads_are_not_read_before_their_wait(kernel_isa):
    """The resident attention forward requests its Q rows with inline-asm loads the compiler does not track (so that it
    does not drain the K / V LDS-DMA with them); the counted s_waitcnt that covers them is hand-placed.  Nothing may read
    or move those destination registers between the loads and that wait -- checked here on the ISA hipcc emits for both
    forms of the kernel (a compiler that copied the registers early would produce garbage only the GPU tests could see)."""
    import re

    text = kernel_isa["attention"]

    def regs(line):
        r = set()
        for m in re.finditer(r"\bv\[(\d+):(\d+)\]", line):
            r.update(range(int(m.group(1)), int(m.group(2)) + 1))
        for m in re.finditer(r"\bv(\d+)\b", line):
            r.add(int(m.group(1)))
        return r

    kernels = list(_kernel_bodies(text, r"_ZN3vit19attn_fwd_res_kernelILi64ELi2ELb1E\w+"))
    assert len(kernels) >= 2, [k for k, _ in kernels]  # the generic DMA form and the compile-time ViT-B form
    for name, body in kernels:
        loads = [i for i, l in enumerate(body) if l.startswith("global_load_dwordx4") and body[i - 1].startswith(";;#ASMSTART")]
        assert len(loads) == 4, (name, len(loads))
        dst = set()
        for i in loads:
            dst |= regs(body[i].split(",")[0])
        wait = next(i for i, l in enumerate(body) if i > loads[-1] and "s_waitcnt vmcnt" in l)
        for i in range(loads[-1] + 1, wait):
            l = body[i]
            if not l or l.startswith(";") or l.startswith(".") or "global_load_lds" in l:
                continue
            assert not (regs(l) & dst), (name, i, l)


def test_untracked_own_row_loads_of_the_resident_backward(kernel_isa):
    """The DMA forms of the two resident backward kernels (r04: images requested in reading order, per-tile counted waits) read the
    wave's own rows -- Q / dO / O / O_lo / lse, or K / V -- with inline-asm loads the compiler does not track, the oldest operations
    in the vmcnt order.  Between each of those loads and the first counted wait nothing may read, copy or overwrite its
    destination registers (the data has not landed): checked on the emitted ISA, as for the forward's Q loads."""
    import re

    text = kernel_isa["attention"]

    def regs(line):
        r = set()
        for m in re.finditer(r"\bv\[(\d+):(\d+)\]", line):
            r.update(range(int(m.group(1)), int(m.group(2)) + 1))
        for m in re.finditer(r"\bv(\d+)\b", line):
            r.add(int(m.group(1)))
        return r

    kernels = list(_kernel_bodies(text, r"_ZN3vit2[23]attn_bwd_d(?:q|kv)_res_kernelILi64ELi2ELb1E\w+"))
    assert len(kernels) == 2, [k for k, _ in kernels]
    for name, body in kernels:
        loads = [i for i, l in enumerate(body) if l.startswith("global_load_dword") and "lds" not in l and body[i - 1].startswith(";;#ASMSTART")]
        assert len(loads) in (8, 18), (name, len(loads))  # dK/dV: K, V rows; dQ: Q, dO, O, O_lo rows + lse
        wait = next(i for i, l in enumerate(body) if i > loads[-1] and "s_waitcnt vmcnt" in l)
        for li in loads:
            dst = regs(body[li].split(",")[0])
            for i in range(li + 1, wait):
                l = body[i]
                if not l or l.startswith(";") or l.startswith(".") or "global_load_lds" in l:
                    continue
                assert not (regs(l) & dst), (name, li, i, l)
        assert not [l for l in body if "scratch_" in l], name
