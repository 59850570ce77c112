"""ctypes binding of libvit_amd.so (include/vit_amd.h).  Fails loudly when the library is missing or a call errors:
there is no CPU / eager fallback anywhere in vit_amd."""
from __future__ import annotations

import ctypes as C
import os
import re
import threading

_HERE = os.path.dirname(os.path.abspath(__file__))
# VIT_AMD_LIB: another build of the same library (the diagnostic twin of `python -m vit_amd.build --diag`, tools/pp_diag.py)
LIB_PATH = os.environ.get("VIT_AMD_LIB") or os.path.join(_HERE, "lib", "libvit_amd.so")
HEADER_PATH = os.path.join(os.path.dirname(_HERE), "include", "vit_amd.h")

VIT_OK = 0
VIT_F32, VIT_BF16 = 0, 1
ACT_NONE, ACT_GELU, ACT_DGELU, ACT_GELU_GRAD, ACT_MUL_AUX = 0, 1, 2, 3, 4
LOSS_MSE, LOSS_L1, LOSS_CE = 0, 1, 2


class VitError(RuntimeError):
    pass


class GemmDesc(C.Structure):
    _fields_ = [
        ("M", C.c_int), ("N", C.c_int), ("K", C.c_int),
        ("a_trans", C.c_int), ("b_trans", C.c_int),
        ("ab_dtype", C.c_int),
        ("A", C.c_void_p), ("lda", C.c_int64),
        ("B", C.c_void_p), ("ldb", C.c_int64),
        ("C", C.c_void_p), ("ldc", C.c_int64), ("c_dtype", C.c_int),
        ("alpha", C.c_float),
        ("bias", C.c_void_p),
        ("act", C.c_int),
        ("aux_out", C.c_void_p),
        ("aux_in", C.c_void_p),
        ("ldaux", C.c_int64),
        ("dropout_p", C.c_float), ("seed", C.c_uint64), ("site", C.c_uint64),
        ("residual", C.c_void_p), ("ldres", C.c_int64),
        ("rows_per_batch", C.c_int), ("out_batch_rows", C.c_int), ("out_row_offset", C.c_int),
        ("split_k", C.c_int),
        ("accumulate", C.c_int),
        ("colsum_out", C.c_void_p),
        ("rope_cos", C.c_void_p), ("rope_sin", C.c_void_p),
        ("rope_T", C.c_int), ("rope_dh", C.c_int), ("rope_cols", C.c_int),
    ]


_P, _I, _F, _U64, _I64, _SZ = C.c_void_p, C.c_int, C.c_float, C.c_uint64, C.c_int64, C.c_size_t

# name -> argtypes (restype is int unless listed in _RESTYPES)
_PROTOS = {
    "vit_version": [],
    "vit_last_error": [],
    "vit_create": [C.POINTER(_P), _I],
    "vit_destroy": [_P],
    "vit_set_workspace": [_P, _P, _SZ],
    "vit_set_option": [C.c_char_p, _I],
    "vit_handle_set_option": [_P, C.c_char_p, _I],
    "vit_step_state_bind": [_P, _P],
    "vit_step_advance": [_P, _U64, _F, _F, _P],
    "vit_adamw_step_dyn": [_P, _P, _P, _P, _P, _P, _I64, _F, _F, _F, _F, _P, _F, _P],
    "vit_gemm": [_P, C.POINTER(GemmDesc), _P],
    "vit_last_gemm_kernel": [],
    "vit_linear_fwd": [_P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _P, _F, _U64, _U64, _P, _P],
    "vit_linear_bwd_dx": [_P, _P, _P, _P, _I, _I, _I, _I, _P, _P],
    "vit_linear_bwd_dw": [_P, _P, _P, _P, _I, _I, _I, _I, _P],
    "vit_layernorm_fwd": [_P, _P, _P, _P, _P, _I, _P, _P, _I, _I, _F, _P],
    "vit_layernorm_fwd_residual": [_P, _P, _P, _I, _P, _P, _P, _P, _I, _P, _P, _I, _I, _F, _P],
    "vit_layernorm_bwd": [_P, _P, _I, _P, _P, _P, _P, _P, _P, _P, _P, _I, _I, _P],
    "vit_layernorm_bwd_fused": [_P, _P, _I, _P, _P, _P, _P, _P, _P, _P, _P, _I, _I, _P, _I, _P, _F, _U64, _U64, _P],
    "vit_attention_fwd": [_P, _P, _P, _P, _I, _I, _I, _I, _I, _F, _F, _U64, _U64, _P],
    "vit_attention_bwd": [_P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _F, _F, _U64, _U64, _P],
    "vit_attention_bwd_colsum": [_P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _F, _F, _U64, _U64, _P, _P],
    "vit_attention_fwd_lo": [_P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _F, _F, _U64, _U64, _P],
    "vit_attention_bwd_lo": [_P, _P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _F, _F, _U64, _U64, _P, _P],
    "vit_attention_probs": [_P, _P, _P, _I, _I, _I, _I, _I, _F, _P],
    "vit_unfold_cast": [_P, _P, _P, _I, _I, _I, _I, _I, _I, _P],
    "vit_fold_add": [_P, _P, _P, _I, _I, _I, _I, _I, _P],
    "vit_add_noise": [_P, _P, _P, _P, C.c_long, _F, _U64, _P],
    "vit_rope_qk": [_P, _P, _I, _P, _P, C.c_long, _I, _I, _I, C.c_long, _I, _P],
    "vit_embed_finish": [_P, _P, _P, _P, _I, _I, _I, _F, _U64, _U64, _P],
    "vit_embed_finish_bwd": [_P, _P, _P, _I, _P, _P, _I, _I, _I, _F, _U64, _U64, _I, _P],
    "vit_dropout_bwd_cast": [_P, _P, _P, _I, _I, _I, _F, _U64, _U64, _P],
    "vit_colsum": [_P, _P, _I, _I64, _P, _I, _I, _I, _P],
    "vit_cast_f32_bf16": [_P, _P, _P, _I64, _P],
    "vit_cast_bf16_f32": [_P, _P, _P, _I64, _F, _P],
    "vit_head_loss_fwd": [_P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _P],
    "vit_head_loss_bwd": [_P, _P, _P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _P],
    "vit_grad_sqnorm": [_P, _P, _I64, _P, _P],
    "vit_grad_sqnorm_acc": [_P, _P, _I64, _P, _P],
    "vit_adamw_step": [_P, _P, _P, _P, _P, _P, _I64, _F, _F, _F, _F, _F, _I, _P, _F, _P],
}
_RESTYPES = {"vit_last_error": C.c_char_p, "vit_last_gemm_kernel": C.c_char_p}

_lib = None
_lock = threading.Lock()


def declared_symbols(header_path: str = HEADER_PATH):
    """Every function the C header declares (used by the CPU test that checks the library exports them all)."""
    text = open(header_path).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(vit_[a-z0-9_]+)\s*\(", text)))


def load():
    """dlopen the library (after torch, so both share torch's HIP runtime) and set prototypes."""
    global _lib
    if _lib is not None:
        return _lib
    with _lock:
        if _lib is not None:
            return _lib
        import torch  # noqa: F401  (must be first: see vit_amd/build.py)

        if not os.path.exists(LIB_PATH):
            raise VitError(
                f"{LIB_PATH} is missing: run `python -m vit_amd.build` (or __graft_entry__.build()). "
                "vit_amd has no fallback path; the HIP library is required.")
        lib = C.CDLL(LIB_PATH)
        for name, argtypes in _PROTOS.items():
            fn = getattr(lib, name)
            fn.argtypes = argtypes
            fn.restype = _RESTYPES.get(name, C.c_int)
        _lib = lib
        # VIT_OPTIONS="name=value,name=value": kernel-selection knobs of vit_set_option for A/B runs of unmodified scripts
        for item in filter(None, os.environ.get("VIT_OPTIONS", "").split(",")):
            name, _, value = item.partition("=")
            rc = lib.vit_set_option(name.strip().encode(), int(value))
            if rc != VIT_OK:
                raise VitError(f"VIT_OPTIONS: vit_set_option({name!r}, {value}) failed: {lib.vit_last_error().decode()}")
        return lib


def set_option(name: str, value: int):
    check(load().vit_set_option(name.encode(), int(value)), f"vit_set_option({name})")


def check(rc: int, what: str = ""):
    if rc != VIT_OK:
        msg = load().vit_last_error()
        raise VitError(f"{what or 'vit call'} failed (status {rc}): {msg.decode() if msg else ''}")


class Handle:
    """vit_handle + a torch-owned workspace on one device."""

    def __init__(self, device_index: int, workspace_bytes: int = 256 << 20):
        import torch

        self.lib = load()
        self.device_index = device_index
        h = C.c_void_p()
        check(self.lib.vit_create(C.byref(h), device_index), "vit_create")
        self.h = h
        self._ws = None
        self.set_workspace(workspace_bytes)

    def set_workspace(self, nbytes: int):
        import torch

        self._ws = torch.empty(nbytes, dtype=torch.uint8, device=f"cuda:{self.device_index}")
        check(self.lib.vit_set_workspace(self.h, self._ws.data_ptr(), nbytes), "vit_set_workspace")
        self.workspace_bytes = nbytes

    def set_option(self, name: str, value: int):
        """Launch geometry of the calls made through THIS handle (vit_handle_set_option): 'reserve_cus'."""
        check(self.lib.vit_handle_set_option(self.h, name.encode(), int(value)), f"vit_handle_set_option({name})")

    def ensure_workspace(self, nbytes: int):
        if nbytes > self.workspace_bytes:
            self.set_workspace(int(nbytes * 1.25))

    def __del__(self):
        try:
            if getattr(self, "h", None):
                self.lib.vit_destroy(self.h)
                self.h = None
        except Exception:
            pass


_handles = {}


def handle_for(device) -> Handle:
    import torch

    idx = device.index if device.index is not None else torch.cuda.current_device()
    if idx not in _handles:
        _handles[idx] = Handle(idx)
    return _handles[idx]
