"""ctypes binding of include/mudpt.h (libmudpt_hip.so).  No fallback: a missing library is an error."""
from __future__ import annotations

import ctypes as C
import os
import re

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("MUDPT_LIB") or os.path.join(HERE, "lib", "libmudpt_hip.so")  # MUDPT_LIB: A/B runs of two builds on one box
HEADER_PATH = os.path.join(os.path.dirname(HERE), "include", "mudpt.h")

BF16, F16, F32 = 0, 1, 2  # F32: the parity mode (include/mudpt.h MUDPT_F32)
VARIANT_MUDPT, VARIANT_COCOOP = 0, 1
ABI_VERSION = 6
EPI_STORE, EPI_GELU, EPI_RESIDUAL, EPI_GELU_BWD, EPI_PATCH, EPI_STORE_F32 = range(6)


class MudptError(RuntimeError):
    pass


class Config(C.Structure):
    _fields_ = [(n, C.c_int32) for n in (
        "image_size", "patch", "v_width", "v_layers", "v_heads", "t_width", "t_layers", "t_heads", "ctx_len",
        "embed_dim", "n_ctx", "depth", "n_cls", "max_batch", "dtype", "variant")]


_vp, _i32, _f32, _sz = C.c_void_p, C.c_int32, C.c_float, C.c_size_t

# name -> (restype, argtypes); must list every function include/mudpt.h declares (tests check that)
SIGNATURES = {
    "mudpt_abi_version": (_i32, []),
    "mudpt_last_error": (C.c_char_p, []),
    "mudpt_create": (_i32, [C.POINTER(Config), C.POINTER(_vp)]),
    "mudpt_destroy": (_i32, [_vp]),
    "mudpt_set_weight": (_i32, [_vp, C.c_char_p, _vp, _sz]),
    "mudpt_set_class_prompts": (_i32, [_vp, _vp, _vp]),
    "mudpt_param_count": (_i32, [_vp]),
    "mudpt_param_numel": (_sz, [_vp]),
    "mudpt_param_info": (_i32, [_vp, _i32, C.POINTER(C.c_char_p), C.POINTER(_sz), C.POINTER(_sz), C.POINTER(_i32),
                                C.POINTER(C.c_int64 * 3)]),
    "mudpt_bind_params": (_i32, [_vp, _vp, _vp]),
    "mudpt_forward": (_i32, [_vp, _vp, _i32, _vp, _vp]),
    "mudpt_forward_ex": (_i32, [_vp, _vp, _i32, _vp, _i32, _vp]),
    "mudpt_forward_backward": (_i32, [_vp, _vp, _vp, _i32, _f32, _vp, _vp, _vp]),
    "mudpt_sgd_step": (_i32, [_vp, _f32, _f32, _f32, _f32, _i32, _vp]),
    "mudpt_sgd_reset": (_i32, [_vp]),
    "mudpt_allreduce_grads": (_i32, [_vp, _vp, _vp]),
    "mudpt_text_layout": (_i32, [_vp, C.POINTER(_i32), C.POINTER(_i32), C.POINTER(_i32)]),
    "mudpt_set_class_shard": (_i32, [_vp, _i32, _i32]),
    "mudpt_cp_buffers": (_i32, [_vp, C.POINTER(_vp), C.POINTER(_vp), C.POINTER(_sz)]),
    "mudpt_cp_forward": (_i32, [_vp, _vp, _i32, _i32, _vp]),
    "mudpt_cp_head": (_i32, [_vp, _vp, _i32, _f32, _vp, _vp, _i32, _vp]),
    "mudpt_cp_backward": (_i32, [_vp, _i32, _vp]),
    "mudpt_debug_read": (_i32, [_vp, C.c_char_p, _i32, _vp, _sz, C.POINTER(_sz)]),
    "mudpt_set_loss_scale": (_i32, [_vp, _f32]),
    "mudpt_model_set": (_i32, [_vp, C.c_char_p, _i32]),
    "mudpt_profile_enable": (_i32, [_vp, _i32]),
    "mudpt_profile_read": (_i32, [_vp, C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(C.c_int64)]),
    "mudpt_profile_read_classes": (_i32, [_vp, C.POINTER(C.c_double * 5), C.POINTER(C.c_double * 5), C.POINTER(C.c_int64 * 5), C.POINTER(C.c_double)]),
    "mudpt_gemm": (_i32, [_i32, _i32, _i32, _i32, _i32, _vp, _i32, _vp, _i32, _vp, _vp, _i32, _vp, _i32, _vp, _i32,
                          _i32, _i32, _vp, _i32, _vp]),
    "mudpt_gemm_split": (_i32, [_i32, _i32, _i32, _i32, _i32, _vp, _vp, _i32, _i32, _vp, _vp, _i32, _i32, _vp, _vp, _i32, _vp, _vp, _i32, _i32, _vp, _i32, _i32, _vp]),
    "mudpt_e4m3_from_f32": (_i32, [_vp, _vp, _sz, _i32]),
    "mudpt_layernorm_fwd_split": (_i32, [_i32, _vp, _i32, _vp, _vp, _vp, _vp, _i32, _i32, _i32, _i32, _vp]),
    "mudpt_layernorm_fwd": (_i32, [_i32, _vp, _i32, _vp, _vp, _vp, _vp, _i32, _i32, _vp, _vp, _i32, _i32, _vp]),
    "mudpt_layernorm_bwd": (_i32, [_i32, _vp, _i32, _i32, _vp, _i32, _vp, _vp, _vp, _vp, _vp, _i32, _vp, _i32, _vp,
                                   _i32, _i32, _i32, _vp]),
    "mudpt_attention_padded_len": (_i32, [_i32]),
    "mudpt_attention_fwd": (_i32, [_i32, _vp, _vp, _vp, _i32, _i32, _i32, _i32, _vp]),
    "mudpt_attention_fwd_exact": (_i32, [_vp, _vp, _vp, _vp, _i32, _i32, _vp, _i32, _i32, _i32, _i32, _vp]),
    "mudpt_attention_bwd": (_i32, [_i32, _vp, _vp, _vp, _vp, _vp, _vp, _i32, _i32, _i32, _i32, _vp]),
    "mudpt_attention_fwd_single": (_i32, [_i32, _vp, _vp, _vp, _vp, _vp, _i32, _i32, _i32, _i32, _vp]),
    "mudpt_attention_bwd_single": (_i32, [_i32, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i32, _i32, _i32, _i32, _vp]),
    "mudpt_layernorm_fwd_fused": (_i32, [_i32, _vp, _i32, _vp, _vp, _i32, _vp, _i32, _i32, _i32, _vp, _i32, _vp, _vp, _vp, _i32, _i32,
                                         _vp, _vp, _i32, _i32, _vp]),
    "mudpt_head": (_i32, [_vp, _vp, _vp, _f32, _f32, _i32, _i32, _i32, _vp, _vp, _vp, _vp, _vp]),
    "mudpt_reduce_rows": (_i32, [_i32, _vp, _vp, _i32, _i32, _i32, _i32, _i32, _vp, _i32, _i32, _f32, _vp]),
    "mudpt_cocoop_dbias": (_i32, [_i32, _vp, _vp, _vp, _i32, _i32, _i32, _i32, _i32, _f32, _vp]),
    "mudpt_sgemm": (_i32, [_i32, _i32, _i32, _i32, _i32, _f32, _vp, _i32, _vp, _i32, _f32, _vp, _i32, _vp, _vp]),
}

_lib = None


def declared_functions(header: str = HEADER_PATH):
    """Function names declared in include/mudpt.h."""
    text = re.sub(r"/\*.*?\*/", "", open(header).read(), flags=re.S)
    return sorted(set(re.findall(r"\b(mudpt_[a-z0-9_]+)\s*\(", text)))


def load():
    """Load libmudpt_hip.so (built by mudpt_amd.build / __graft_entry__.build()); never falls back."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise MudptError(f"{LIB_PATH} is missing: run `python -m mudpt_amd.build` (hipcc, gfx950) first; "
                         "there is no CPU fallback for the MuDPT path")
    # torch ships its own libamdhip64; importing it first makes the library bind to that same HIP runtime
    # instance (same SONAME), which it must share with torch's allocator and streams.  Loaded the other way
    # round the process ends up with two runtimes and the library sees "no ROCm-capable device".
    import torch  # noqa: F401
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)
        fn.restype, fn.argtypes = res, args
    _lib = lib
    return lib


def check(rc: int, what: str = ""):
    if rc != 0:
        msg = load().mudpt_last_error().decode(errors="replace")
        kind = {1: "bad argument", 2: "HIP error", 3: "bad state"}.get(rc, f"error {rc}")
        if rc == 1:
            raise AssertionError(f"mudpt {what}: {msg}")  # the reference asserts on bad cfg (trainers/mudpt.py:52,55,190)
        raise MudptError(f"mudpt {what}: {kind}: {msg}")


CP_VISION, CP_TEXT = 1, 2  # mudpt_cp_backward parts
FWD_REUSE_TEXT, FWD_TRAINING = 1, 2  # mudpt_forward_ex / mudpt_cp_forward flags


class _DeviceMemory:
    """fp32 device memory the library owns, exposed through the CUDA array interface (no copy, no ownership)."""

    def __init__(self, address: int, numel: int):
        self.__cuda_array_interface__ = {"shape": (numel,), "typestr": "<f4", "data": (address, False), "version": 2}


def device_view(address: int, numel: int, device):
    """torch view (fp32, flat) of library-owned device memory: the operand of a torch.distributed collective."""
    import torch
    t = torch.as_tensor(_DeviceMemory(address, numel), device=device)
    if t.data_ptr() != address:
        raise MudptError("device_view: torch copied the buffer instead of aliasing it")
    return t


def ptr(t):
    """Device / host pointer of a torch tensor (or None)."""
    return None if t is None else C.c_void_p(t.data_ptr())
