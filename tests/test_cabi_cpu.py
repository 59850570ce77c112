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
