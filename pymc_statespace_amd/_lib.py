"""ctypes binding of libkf_hip.so -- the C-ABI declared in include/kf.h.

There is deliberately NO fallback: if the HIP library is missing or does not load,
importing the product fails loudly (a silent CPU path would void every parity claim).
"""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# KF_LIB_PATH: developer override for A/B experiments with instrumented builds of the same ABI
LIB_PATH = os.environ.get("KF_LIB_PATH") or os.path.join(_HERE, "libkf_hip.so")

KF_ABI_VERSION = 7

KF_FILTER = {"standard": 0, "cholesky": 1, "univariate": 2, "single": 3}
KF_FLAG_MVN_CONST_EXACT = 1 << 0
KF_FLAG_Y_PER_DRAW = 1 << 1
KF_FLAG_H_DIAGONAL = 1 << 11
KF_FLAG_EXACT_RECURSION = 1 << 12
KF_FLAG_STEADY_STATE = 1 << 13
KF_FLAG_TV_SMOOTHER_REFERENCE_PAIRING = 1 << 14
KF_FLAG_TV = {"T": 1 << 4, "Z": 1 << 5, "R": 1 << 6, "H": 1 << 7, "Q": 1 << 8, "c": 1 << 9, "d": 1 << 10}

KF_WS_FORWARD, KF_WS_BACKWARD, KF_WS_SMOOTHER, KF_WS_ALL = 1, 2, 4, 7
KF_WS_VALUE_AND_GRAD = 8
KF_WS_SMOOTHER_FROM_FILTER = 16
KF_OK, KF_ERR_ARG, KF_ERR_HIP, KF_ERR_WORKSPACE, KF_ERR_UNSUPPORTED = 0, -1, -2, -3, -4

# every symbol include/kf.h declares (tests check the built library exports them all)
EXPORTS = (
    "kf_abi_version", "kf_create", "kf_destroy", "kf_last_error", "kf_set_option", "kf_get_option", "kf_set_lanes_per_draw",
    "kf_set_forward_kernel", "kf_forward_kernel_name", "kf_forward_kernel_name_ws", "kf_last_forward_kernel",
    "kf_workspace_bytes", "kf_workspace_bytes_for", "kf_filter_forward", "kf_filter_backward", "kf_filter_backward_tapes", "kf_filter_value_and_grad", "kf_smoother", "kf_smoother_from_filter",
    "kf_assemble", "kf_lyapunov", "kf_conditional_simulation", "kf_dare", "kf_dare_backward", "kf_simulate_statespace",
)


class KfDims(C.Structure):
    _fields_ = [
        ("B", C.c_int64), ("T", C.c_int32), ("p", C.c_int32), ("m", C.c_int32), ("r", C.c_int32),
        ("filter", C.c_int32), ("flags", C.c_uint32),
    ]


class KfLibraryMissing(ImportError):
    pass


_lib = None


def load():
    """Load libkf_hip.so (built by ``__graft_entry__.build()`` / ``make -C csrc``)."""
    global _lib
    if _lib is not None:
        return _lib
    # torch ships its own libamdhip64; load it FIRST so libkf_hip.so binds to the same HIP
    # runtime the tensors live in (two runtimes in one process do not share devices/streams)
    import torch  # noqa: F401

    if not os.path.exists(LIB_PATH):
        raise KfLibraryMissing(
            f"{LIB_PATH} not found: the HIP extension is not built. Run "
            "`python -c 'import __graft_entry__ as g; g.build()'` (or `make -C "
            "pymc_statespace_amd/csrc`). There is no CPU fallback."
        )
    lib = C.CDLL(LIB_PATH)
    vp, dp, ip = C.c_void_p, C.c_void_p, C.c_void_p  # device pointers travel as integers
    lib.kf_abi_version.restype = C.c_int
    lib.kf_create.argtypes = [C.POINTER(C.c_void_p), C.c_int]
    lib.kf_create.restype = C.c_int
    lib.kf_destroy.argtypes = [C.c_void_p]
    lib.kf_destroy.restype = None
    lib.kf_last_error.argtypes = [C.c_void_p]
    lib.kf_last_error.restype = C.c_char_p
    lib.kf_set_lanes_per_draw.argtypes = [C.c_void_p, C.c_int]
    lib.kf_set_lanes_per_draw.restype = C.c_int
    lib.kf_set_forward_kernel.argtypes = [C.c_void_p, C.c_int]
    lib.kf_set_forward_kernel.restype = C.c_int
    lib.kf_forward_kernel_name.argtypes = [C.c_void_p, C.POINTER(KfDims)]
    lib.kf_forward_kernel_name.restype = C.c_char_p
    lib.kf_workspace_bytes.argtypes = [C.c_void_p, C.POINTER(KfDims), C.POINTER(C.c_size_t)]
    lib.kf_workspace_bytes.restype = C.c_int
    lib.kf_forward_kernel_name_ws.argtypes = [C.c_void_p, C.POINTER(KfDims), C.c_size_t]
    lib.kf_forward_kernel_name_ws.restype = C.c_char_p
    lib.kf_last_forward_kernel.argtypes = [C.c_void_p]
    lib.kf_last_forward_kernel.restype = C.c_char_p
    lib.kf_set_option.argtypes = [C.c_void_p, C.c_char_p, C.c_char_p]
    lib.kf_set_option.restype = C.c_int
    lib.kf_get_option.argtypes = [C.c_void_p, C.c_char_p]
    lib.kf_get_option.restype = C.c_char_p
    lib.kf_workspace_bytes_for.argtypes = [C.c_void_p, C.POINTER(KfDims), C.c_int, C.POINTER(C.c_size_t)]
    lib.kf_workspace_bytes_for.restype = C.c_int
    lib.kf_filter_forward.argtypes = [vp, C.POINTER(KfDims)] + [dp] * 10 + [dp] * 6 + [ip, vp,
                                                                                     C.c_size_t, vp]
    lib.kf_filter_forward.restype = C.c_int
    lib.kf_filter_backward.argtypes = [vp, C.POINTER(KfDims)] + [dp] * 10 + [dp] * 4 + [dp] * 9 + [
        vp, C.c_size_t, vp]
    lib.kf_filter_backward.restype = C.c_int
    lib.kf_filter_backward_tapes.argtypes = [vp, C.POINTER(KfDims)] + [dp] * 10 + [dp] * 8 + [dp] * 9 + [
        vp, C.c_size_t, vp]
    lib.kf_filter_backward_tapes.restype = C.c_int
    lib.kf_filter_value_and_grad.argtypes = [vp, C.POINTER(KfDims)] + [dp] * 10 + [dp, dp, ip] + [dp] * 9 + [
        vp, C.c_size_t, vp]
    lib.kf_filter_value_and_grad.restype = C.c_int
    lib.kf_smoother.argtypes = [vp, C.POINTER(KfDims)] + [dp] * 5 + [dp] * 2 + [vp, C.c_size_t, vp]
    lib.kf_smoother.restype = C.c_int
    lib.kf_smoother_from_filter.argtypes = [vp, C.POINTER(KfDims)] + [dp] * 7 + [dp] * 2 + [vp, C.c_size_t, vp]
    lib.kf_smoother_from_filter.restype = C.c_int
    lib.kf_assemble.argtypes = [vp, C.c_int64, C.c_int32, dp, C.c_int64, dp, C.c_int32, dp, dp, dp, vp]
    lib.kf_assemble.restype = C.c_int
    lib.kf_lyapunov.argtypes = [vp, C.c_int64, C.c_int32, C.c_int32, dp, dp, dp, dp, dp, ip, vp]
    lib.kf_lyapunov.restype = C.c_int
    lib.kf_conditional_simulation.argtypes = [vp, C.c_int64, C.c_int32, C.c_int32, C.c_int32, dp, dp, dp,
                                              C.c_double, dp, ip, vp]
    lib.kf_conditional_simulation.restype = C.c_int
    lib.kf_dare.argtypes = [vp, C.c_int64, C.c_int32, C.c_int32, C.c_int32, dp, dp, dp, dp, dp, dp, dp, ip,
                            C.c_int32, C.c_double, vp]
    lib.kf_dare.restype = C.c_int
    lib.kf_dare_backward.argtypes = [vp, C.c_int64, C.c_int32, C.c_int32, C.c_int32, dp, dp, dp, dp, dp, dp, dp,
                                     dp, dp, dp, dp, dp, C.c_int32, ip, vp]
    lib.kf_dare_backward.restype = C.c_int
    lib.kf_simulate_statespace.argtypes = [vp, C.c_int64, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32,
                                           dp, dp, dp, dp, dp, dp, dp, dp, dp, dp, ip, vp]
    lib.kf_simulate_statespace.restype = C.c_int
    if lib.kf_abi_version() != KF_ABI_VERSION:
        raise KfLibraryMissing(f"{LIB_PATH}: ABI version {lib.kf_abi_version()} != {KF_ABI_VERSION}")
    _lib = lib
    return lib
