"""ctypes binding of libjyutvoice_hip.so (include/jyutvoice_hip.h).

The library is the product; there is no Python/CPU fallback.  `load()` raises if the shared object is
missing or cannot be loaded, and every wrapper raises `JvError` with the library's own message when a
call fails.
"""
from __future__ import annotations

import ctypes as C
import os

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("JYUTVOICE_HIP_LIB") or os.path.join(HERE, "libjyutvoice_hip.so")

JV_MODEL_TTS = 0
JV_MODEL_HIFT = 1
JV_MODEL_PROMPT = 2

ACT = {"none": 0, "relu": 1, "gelu": 2, "mish": 3, "elu": 4, "silu": 5}
PRO = {"none": 0, "snake": 1, "lrelu": 2}


class JvError(RuntimeError):
    def __init__(self, code: int, msg: str):
        super().__init__(f"libjyutvoice_hip error {code}: {msg}")
        self.code = code
        self.msg = msg


_p = C.c_void_p
_f = C.c_float
_i = C.c_int
_i64 = C.c_int64

# name -> (restype, argtypes); must list every symbol include/jyutvoice_hip.h declares (tests/test_abi.py)
SIGNATURES = {
    "jv_create": (_i, [C.POINTER(_p), _i, _i, _i, _i]),
    "jv_destroy": (None, [_p]),
    "jv_reserve": (_i, [_p, _i, _i, _i]),
    "jv_usable": (_i, [_p]),
    "jv_last_error": (C.c_char_p, []),
    "jv_num_tensors": (_i, [_p]),
    "jv_tensor_name": (C.c_char_p, [_p, _i]),
    "jv_tensor_model": (_i, [_p, _i]),
    "jv_tensor_ndim": (_i, [_p, _i]),
    "jv_tensor_dim": (_i64, [_p, _i, _i]),
    "jv_load_tensor": (_i, [_p, C.c_char_p, _p, C.POINTER(_i64), _i, _i, _p]),
    "jv_load_noise": (_i, [_p, _p, _i64, _i, _p]),
    "jv_finalize": (_i, [_p, _i, _p]),
    "jv_flow_estimator_step": (_i, [_p, _p, _p, _p, _p, _p, _p, _i, _i, _p, _p]),
    "jv_flow_estimator_masked": (_i, [_p, _p, _p, _p, _p, _p, _p, _i, _i, _p, _p]),
    "jv_flow_set_streaming": (_i, [_p, _i]),
    "jv_flow_set_graph": (_i, [_p, _i]),
    "jv_flow_set_contraction": (_i, [_p, _i]),
    "jv_flow_contraction_info": (_i, [_p, _p, _i]),
    "jv_cfm_solve": (_i, [_p, _p, _p, _p, _p, _i, _i, _i, _f, _p, _p, _p]),
    "jv_encoder_fwd": (_i, [_p, _p, _p, _p, _p, _p, _p, _p, _i, _i, _p, _p, _p, _p, _p]),
    "jv_load_mel_basis": (_i, [_p, _p, _i64, _i, _p]),
    "jv_mel_spectrogram": (_i, [_p, _p, _i, _i, _p, _p]),
    "jv_prompt_encoder_fwd": (_i, [_p, _p, _p, _i, _i, _p, _p]),
    "jv_length_regulate": (_i, [_p, _p, _p, _p, _i, _i, _f, _p, _p, _i, _p, _p, _p]),
    "jv_hift_f0": (_i, [_p, _p, _p, _i, _i, _p, _p]),
    "jv_hift_source": (_i, [_p, _p, _p, _p, _i, _i, _p, _p]),
    "jv_hift_source_seeded": (_i, [_p, _p, _p, C.c_uint64, C.c_uint32, _i, _i, _p, _p]),
    "jv_hift_decode": (_i, [_p, _p, _p, _p, _i, _i, _p, _p]),
    "jv_op_conv_gemm": (_i, [_p, _i64, _i, _i, _i, _i, _i, _p, _i, _p, _i, _i, _p, _f, _p, _p, _f, _p, _p, _p, _p]),
    "jv_op_attention": (_i, [_p, _p, _i, _i, _i, _i, _p, _p]),
    "jv_h3_scale_for_bound": (_f, [_f]),
    "jv_op_conv_h3_measured": (_i, [_p, _i64, _i, _i, _i, _i, _i, _p, _i, _p, _i, _i, _p, _f, _p, _p, _p, _f, _p, _p, _p]),
    "jv_op_attention_h3": (_i, [_p, _p, _i, _i, _i, _i, _f, _f, _f, _p, _p]),
    "jv_op_linear_h3": (_i, [_p, _i64, _i, _i, _p, _i, _p, _i, _p, _f, _i, _p, _p]),
    "jv_op_attention_planes": (_i, [_p, _i64, _p, _i, _i, _i, _i, _f, _f, _f, _i, _f, _p, _p, _p]),
    "jv_op_hiftconv": (_i, [_p, _i64, _i, _i, _i, _p, _p, _p, _p, _p, _p, _f, _i, _p, _f, _p, _p, _p]),
    "jv_op_rowconv": (_i, [_p, _i64, _i, _i, _p, _p, _p, _p, _i, _p, _p, _p, _p, _p, _p, _p]),
    "jv_op_rowgemm": (_i, [_p, _i64, _i, _i, _p, _i, _p, _i, _p, _p, _p, _f, _f, _i, _p, _p, _p, _p]),
    "jv_op_layernorm": (_i, [_p, _p, _p, _f, _i64, _i, _p, _p]),
    "jv_profile_enable": (_i, [_i]),
    "jv_profile_report": (_i, [C.c_char_p, _i64]),
}

_lib = None


def load() -> C.CDLL:
    """dlopen the in-tree library (building it is `python -m jyutvoice_amd.build` / __graft_entry__.build())."""
    global _lib
    if _lib is not None:
        return _lib
    # PyTorch-ROCm bundles its own libamdhip64.so.7; it must be the first HIP runtime in the process so that this
    # library binds to the same runtime instance (device pointers and streams are shared with torch).
    import torch  # noqa: F401
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f"{LIB_PATH} is missing: build it with `python -m jyutvoice_amd.build` (needs hipcc). "
            "jyutvoice_amd has no CPU fallback.")
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)   # AttributeError here = header/library drift
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def check(rc: int) -> None:
    if rc != 0:
        msg = load().jv_last_error()
        raise JvError(rc, msg.decode("utf-8", "replace") if msg else "?")
