"""ctypes binding of libmhla_hip.so (C ABI: include/mhla_hip.h).  No CPU fallback: if the
library is missing the product path raises."""
import ctypes
import os
from ctypes import POINTER, Structure, c_char_p, c_float, c_int, c_int32, c_int64, c_size_t, c_uint, c_void_p

_HERE = os.path.dirname(os.path.abspath(__file__))
# MHLA_LIB_PATH: A/B comparison of two builds of the same library (tools); the default is the in-tree build
LIB_PATH = os.environ.get("MHLA_LIB_PATH") or os.path.join(_HERE, "lib", "libmhla_hip.so")

ABI_VERSION = 9
F32, BF16, F16 = 0, 1, 2
FLAG_RELU_EPS = 1
FLAG_FORCE_GENERIC = 2
FLAG_NO_SMALLN = 4
FLAG_BF16_SUMMARIES = 8   # opt-in reduced precision of the block-mixing operator on 16-bit tensors (mhla_hip.h)
FLAG_NO_BWD_STATE = 16      # forward without a backward to come: skip the state only the backward reads
FLAG_FP32_GRADE_SUMMARIES = 32   # 16-bit tensors: block summaries with >= 16 significand bits (default: 11, fp16 payload x row multiplier)
CAUSAL_FORCE_GENERIC = 1
CAUSAL_BF16_SUMMARIES = 2
CAUSAL_FP32_GRADE_SUMMARIES = 4   # chunk summaries as bf16 hi + lo pairs (default: h16, 11 significand bits in 2 bytes)


class View(Structure):
    """mhla_view / mhla_mview: token-major [B, N, H, D], element strides, D contiguous."""
    _fields_ = [("ptr", c_void_p), ("sb", c_int64), ("sn", c_int64), ("sh", c_int64)]


NULL_VIEW = View(None, 0, 0, 0)

# name -> (restype, argtypes); every symbol include/mhla_hip.h declares
SIGNATURES = {
    "mhla_abi_version": (c_int, []),
    "mhla_build_flags": (c_char_p, []),
    "mhla_last_error": (c_char_p, []),
    "mhla_set_option": (c_int, [c_char_p, c_int]),
    "mhla_prof_enable": (None, [c_int]),
    "mhla_prof_report": (c_int, [c_char_p, c_size_t]),
    "mhla_debug_set_trace": (None, [c_void_p]),
    "mhla_blockmix_fwd_ws_bytes": (c_size_t, [c_int] * 7 + [c_uint]),
    "mhla_blockmix_bwd_ws_bytes": (c_size_t, [c_int] * 7 + [c_uint]),
    "mhla_blockmix_fwd_keeps_state": (c_int, [c_int] * 7 + [c_uint]),
    "mhla_blockmix_fwd": (c_int, [View, View, View, View, View, c_void_p, c_int, View, c_void_p, c_void_p, c_size_t,
                                  c_int, c_int, c_int, c_int, c_int, c_int, c_float, c_uint, c_void_p]),
    "mhla_blockmix_rope_fwd": (c_int, [View, View, View, c_int, c_void_p, c_int, c_void_p, c_void_p, c_int64, View, c_void_p,
                                       c_void_p, c_size_t, c_int, c_int, c_int, c_int, c_int, c_int, c_float, c_uint,
                                       c_void_p]),
    "mhla_blockmix_wan_fwd": (c_int, [View, View, View, c_int, c_void_p, c_int, c_void_p, c_void_p, c_int64, c_void_p, c_float,
                                      View, View, c_int, c_void_p, c_void_p, c_size_t, c_int, c_int, c_int, c_int, c_int,
                                      c_int, c_float, c_uint, c_void_p]),
    "mhla_blockmix_wan_pro_ok": (c_int, [c_int, c_int, c_int, c_int, c_uint]),
    "mhla_blockmix_wan_pro_fwd": (c_int, [View, View, View, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_void_p, c_int, c_void_p, c_void_p,
                                          c_int64, c_void_p, c_float, View, View, c_int, c_void_p, c_void_p, c_size_t, c_int, c_int, c_int,
                                          c_int, c_int, c_int, c_float, c_uint, c_void_p]),
    "mhla_describe_dispatch": (c_int, [c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_uint, c_char_p, c_size_t]),
    "mhla_causal_describe_dispatch": (c_int, [c_int, c_int, c_int, c_int, c_int, c_uint, c_char_p, c_size_t]),
    "mhla_rms_rstd": (c_int, [c_void_p, c_int64, c_void_p, c_int64, c_int, c_float, c_int, c_void_p]),
    "mhla_blockmix_bwd": (c_int, [View, View, View, View, View, c_void_p, c_int, View, View, View, View, View, View,
                                  View, c_void_p, c_void_p, c_void_p, c_size_t, c_void_p, c_int, c_int, c_int, c_int,
                                  c_int, c_int, c_float, c_uint, c_void_p]),
    "mhla_blockmix_rope_bwd": (c_int, [View, View, View, c_int, c_void_p, c_int, c_void_p, c_void_p, c_int64, View, View, View, View,
                                       View, c_void_p, c_void_p, c_void_p, c_size_t, c_void_p, c_int, c_int, c_int, c_int, c_int,
                                       c_int, c_float, c_uint, c_void_p]),
    "mhla_blockmix_bwd_status": (c_int, [c_void_p, c_size_t] + [c_int] * 7 + [c_uint, c_void_p]),
    "mhla_causal_fwd_ws_bytes": (c_size_t, [c_int] * 7 + [c_uint]),
    "mhla_causal_bwd_ws_bytes": (c_size_t, [c_int] * 7 + [c_uint]),
    "mhla_causal_normgate_fusable": (c_int, [c_int] * 5 + [c_uint]),
    "mhla_causal_fwd": (c_int, [View, View, View, c_void_p, c_int, View, c_void_p, c_size_t, c_int, c_int, c_int,
                                c_int, c_int, c_int, c_float, c_int, c_uint, c_void_p]),
    "mhla_causal_normgate_fwd": (c_int, [View, View, View, c_void_p, c_int, View, View, c_void_p, c_float, View, c_void_p,
                                         c_size_t, c_int, c_int, c_int, c_int, c_int, c_int, c_float, c_int, c_uint, c_void_p]),
    "mhla_causal_bwd": (c_int, [View, View, View, c_void_p, c_int, View, View, View, View, c_void_p, c_int, c_void_p,
                                c_size_t, c_void_p, c_int, c_int, c_int, c_int, c_int, c_int, c_float, c_int, c_uint, c_void_p]),
    "mhla_featmap_rotary": (c_int, [View, View, c_void_p, c_void_p, c_int64, c_int64, View, c_int, c_int, c_int, c_int, c_int,
                                    c_int, c_int, c_void_p]),
    "mhla_lepe2d": (c_int, [c_void_p, c_int64, c_int64, c_void_p, c_void_p, c_void_p, c_int64, c_int64, c_void_p, c_int64,
                            c_int64, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_void_p]),
    "mhla_lepe2d_wgrad_ws_bytes": (c_size_t, [c_int, c_int]),
    "mhla_lepe2d_wgrad": (c_int, [c_void_p, c_int64, c_int64, c_void_p, c_int64, c_int64, c_void_p, c_void_p, c_size_t,
                                  c_int, c_int, c_int, c_int, c_int, c_int, c_void_p]),
    "mhla_lepe3d": (c_int, [c_void_p, c_int64, c_int64, c_void_p, c_void_p, c_void_p, c_int64, c_int64, c_void_p, c_int64,
                            c_int64, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_void_p]),
    "mhla_lepe3d_wgrad_ws_bytes": (c_size_t, [c_int]),
    "mhla_lepe3d_wgrad": (c_int, [c_void_p, c_int64, c_int64, c_void_p, c_int64, c_int64, c_void_p, c_void_p, c_size_t,
                                  c_int, c_int, c_int, c_int, c_int, c_int, c_void_p]),
    "mhla_qk_prologue_rope": (c_int, [c_void_p, c_int64, c_void_p, c_void_p, c_int64, c_void_p, c_int64, c_void_p, c_void_p,
                                      c_int64, c_int, c_int, c_int64, c_int, c_int, c_float, c_float, c_int, c_void_p]),
    "mhla_qk_prologue_dw_rows": (c_int64, [c_int64]),
    "mhla_qk_prologue_bwd": (c_int, [c_void_p, c_int64, c_void_p, c_void_p, c_int64, c_void_p, c_int64, c_void_p, c_void_p,
                                     c_int64, c_int, c_int, c_void_p, c_int64, c_void_p, c_int64, c_int, c_int, c_float,
                                     c_int, c_void_p]),
    "mhla_qk_prologue": (c_int, [c_void_p, c_int64, c_void_p, c_void_p, c_int64, c_int64, c_int, c_int, c_float, c_float,
                                 c_int, c_void_p]),
    "mhla_rmsnorm_gate_fwd": (c_int, [c_void_p, c_int64, c_void_p, c_int64, c_void_p, c_void_p, c_int64, c_void_p,
                                      c_int64, c_int, c_float, c_int, c_void_p]),
    "mhla_rmsnorm_gate_dw_rows": (c_int64, [c_int64]),
    "mhla_rmsnorm_gate_bwd": (c_int, [c_void_p, c_int64, c_void_p, c_int64, c_void_p, c_void_p, c_int64, c_void_p,
                                      c_int64, c_void_p, c_int64, c_void_p, c_int64, c_int, c_float, c_int, c_void_p]),
}

_lib = None


class MhlaLibraryError(RuntimeError):
    pass


def load():
    """Load libmhla_hip.so (once).  Raises MhlaLibraryError when it is not built -- never falls back."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise MhlaLibraryError(
            f"{LIB_PATH} is missing: build it with `python -m mhla_amd.build` (hipcc --offload-arch=gfx950). "
            "mhla_amd has no CPU / eager fallback.")
    lib = ctypes.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)   # AttributeError if a declared symbol is not exported
        fn.restype = res
        fn.argtypes = args
    v = lib.mhla_abi_version()
    if v != ABI_VERSION:
        raise MhlaLibraryError(f"libmhla_hip.so ABI version {v} != expected {ABI_VERSION}; rebuild")
    # The shipped build has no packed-fp32 VALU instructions in device code; a library compiled with
    # another flag set (an older build, a hand-run hipcc command) is refused instead of silently reused.  Variant libraries
    # given through MHLA_LIB_PATH (tools/build_variant.sh) are taken as they are.
    flags = lib.mhla_build_flags().decode()
    if "MHLA_LIB_PATH" not in os.environ and os.environ.get("MHLA_PACKED_FP32") != "1" and "no-packed-fp32" not in flags:
        raise MhlaLibraryError(f"libmhla_hip.so was built with flags [{flags}], expected a no-packed-fp32 build: "
                               "rebuild with `python -m mhla_amd.build --force`")
    _lib = lib
    return lib


def check(rc: int, what: str):
    if rc != 0:
        msg = load().mhla_last_error().decode(errors="replace")
        raise RuntimeError(f"{what} failed (code {rc}): {msg}")
