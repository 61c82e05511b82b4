"""ctypes binding of include/cdhip.h (the C-ABI of the HIP library) and its build.

There is no CPU fallback: if ``libcdhip.so`` is missing or no GPU is present the
calls fail loudly.  ``build()`` cross-compiles for gfx950 with hipcc and works
without a GPU.
"""
from __future__ import annotations

import ctypes as C
import os
import re
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(_HERE, "csrc")
# CDHIP_SO selects another build of the same library (kernel experiments); default is in-tree.
SO_PATH = os.environ.get("CDHIP_SO") or os.path.join(CSRC, "libcdhip.so")
HEADER = os.path.join(os.path.dirname(_HERE), "include", "cdhip.h")


def _sources():
    """Everything under csrc/ the library is compiled from (globbed: a new header cannot be forgotten)."""
    return sorted(f for f in os.listdir(CSRC) if f.endswith((".hip", ".hpp", ".h")))


CDH_OK, CDH_DIM_MISMATCH, CDH_BAD_ARG, CDH_DOMAIN, CDH_HIP_ERROR, CDH_RCCL_ERROR, CDH_OOM = range(7)
CDH_F64, CDH_F32 = 0, 1
CDH_LS, CDH_SQRT, CDH_WLS = 0, 1, 2
CDH_SWEEP_COORD, CDH_SWEEP_BLOCK = 0, 1


class DimensionMismatch(Exception):
    """Julia's DimensionMismatch (coordinate_descent.jl:13,15; cd_differentiable_function.jl:53)."""


class ArgumentError(Exception):
    """Julia's ArgumentError."""


class DomainError(Exception):
    """Julia's DomainError (sqrt of a negative in the sqrt-lasso update)."""


class HipError(RuntimeError):
    """HIP / RCCL / out-of-memory failure inside the library."""


class cdh_options(C.Structure):
    _fields_ = [("maxIter", C.c_int64), ("optTol", C.c_double), ("randomize", C.c_int32),
                ("warmStart", C.c_int32), ("numSteps", C.c_int64), ("seed", C.c_uint64)]


class cdh_stats(C.Structure):
    _fields_ = [("passes", C.c_int64), ("full_passes", C.c_int64), ("visits", C.c_int64),
                ("converged", C.c_int32), ("domain_error", C.c_int32), ("maxH", C.c_double),
                ("lambda_max", C.c_double)]


# int32_t (*cdh_host_allreduce_fn)(void *user, double *inout, int64_t count)
HOST_ALLREDUCE_FN = C.CFUNCTYPE(C.c_int32, C.c_void_p, C.POINTER(C.c_double), C.c_int64)


def needs_build() -> bool:
    if not os.path.exists(SO_PATH):
        return True
    t = os.path.getmtime(SO_PATH)
    deps = [os.path.join(CSRC, s) for s in _sources()] + [HEADER]
    return any(os.path.getmtime(d) > t for d in deps)


def build(force: bool = False) -> str:
    """hipcc --offload-arch=gfx950 csrc/cdhip.hip -> csrc/libcdhip.so (in-tree)."""
    if not force and not needs_build():
        return SO_PATH
    cmd = ["hipcc", "-O3", "-std=c++17", "--offload-arch=gfx950", "-fPIC", "-shared", "-Wall",
           "-o", SO_PATH, os.path.join(CSRC, "cdhip.hip"), "-ldl"]
    subprocess.run(cmd, check=True)
    return SO_PATH


def declared_symbols() -> list[str]:
    """Every function include/cdhip.h declares."""
    txt = open(HEADER).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(cdh_[a-z_A-Z0-9]+)\s*\(", txt)))


_lib = None


def lib():
    """Load libcdhip.so (never builds implicitly on a box without hipcc sources newer than it)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(SO_PATH):
        raise RuntimeError(
            f"{SO_PATH} is missing: run `python -c 'import __graft_entry__ as g; g.build()'` "
            "(hipcc --offload-arch=gfx950). There is no CPU fallback.")
    L = C.CDLL(SO_PATH)
    vp, i64, i32, f64 = C.c_void_p, C.c_int64, C.c_int32, C.c_double
    P = C.POINTER
    sig = {
        "cdh_create": [P(vp), i32, i32, i64, i64, i64, i64, i32],
        "cdh_destroy": [vp],
        "cdh_device_count": [P(i32)],
        "cdh_synchronize": [vp],
        "cdh_set_X_cols": [vp, i64, i64, vp, i64],
        "cdh_get_X_cols": [vp, i64, i64, vp, i64],
        "cdh_set_y": [vp, vp],
        "cdh_get_y": [vp, vp],
        "cdh_set_obs_weights": [vp, vp],
        "cdh_set_loss": [vp, i32],
        "cdh_generate": [vp, C.c_uint64, i64, f64, vp],
        "cdh_set_penalty": [vp, f64, vp, i64],
        "cdh_num_coordinates": [vp, P(i64)],
        "cdh_initialize": [vp, i64, i64, vp, vp],
        "cdh_set_iterate": [vp, i64, i64, vp, vp],
        "cdh_gradient": [vp, i64, P(f64)],
        "cdh_descend": [vp, i64, P(f64)],
        "cdh_lambda_max": [vp, P(f64)],
        "cdh_pass": [vp, i64, vp, P(f64)],
        "cdh_solve": [vp, P(cdh_options), P(cdh_stats)],
        "cdh_coordinate_descent": [vp, P(cdh_options), P(cdh_stats)],
        "cdh_get_beta": [vp, vp],
        "cdh_get_support": [vp, vp, P(i64)],
        "cdh_get_residual": [vp, vp],
        "cdh_col_rms": [vp, vp],
        "cdh_xt_r": [vp, vp],
        "cdh_gram": [vp, i64, vp, vp, vp, P(f64)],
        "cdh_xt_r_cols": [vp, i64, vp, vp],
        "cdh_resid_std": [vp, P(f64), P(f64)],
        "cdh_set_reuse_residual": [vp, i32],
        "cdh_resid_moments": [vp, P(f64), P(f64)],
        "cdh_objective": [vp, P(f64)],
        "cdh_set_sweep_mode": [vp, i32, i32],
        "cdh_set_use_graph": [vp, i32],
        "cdh_set_gradient_cache": [vp, i32],
        "cdh_cache_stats": [vp, P(i64)],
        "cdh_get_gradient_cache": [vp, P(i32)],
        "cdh_set_device_loop": [vp, i32],
        "cdh_device_loop_stats": [vp, P(i64)],
        "cdh_device_loop_table": [vp, P(i64)],
        "cdh_set_onchip_solve": [vp, i32],
        "cdh_onchip_stats": [vp, P(i64)],
        "cdh_onchip_last": [vp, P(i64)],
        "cdh_cache_drift": [vp, i32, P(f64)],
        "cdh_cache_gram_column": [vp, i64, vp, P(f64)],
        "cdh_set_screening": [vp, i32],
        "cdh_comm_unique_id": [vp],
        "cdh_comm_init": [vp, vp, i32, i32],
        "cdh_comm_drop": [vp],
        "cdh_p2p_local_handle": [vp, vp],
        "cdh_p2p_connect": [vp, vp, i32, i32],
        "cdh_p2p_enable": [vp, i32],
        "cdh_exchange_probe": [vp, vp, i64],
        "cdh_set_host_exchange": [vp, HOST_ALLREDUCE_FN, vp, i32, i32],
        "cdh_exchange_stats": [vp, P(i64), P(i64), P(i64), P(i32)],
        "cdh_exchange_latency": [vp, i64, i32, P(f64)],
        "cdh_profile_begin": [vp],
        "cdh_profile_end": [vp, P(f64), P(i64), P(f64)],
    }
    for name, args in sig.items():
        fn = getattr(L, name)
        fn.argtypes = args
        fn.restype = i32
    L.cdh_last_error.argtypes = [vp]
    L.cdh_last_error.restype = C.c_char_p
    _lib = L
    return L


def check(status: int, handle=None):
    """Map a cdh_status to the Julia exception type the reference would throw."""
    if status == CDH_OK:
        return
    msg = lib().cdh_last_error(handle)
    msg = msg.decode() if msg else ""
    if status == CDH_DIM_MISMATCH:
        raise DimensionMismatch(msg)
    if status == CDH_BAD_ARG:
        raise ArgumentError(msg)
    if status == CDH_DOMAIN:
        raise DomainError(msg)
    raise HipError(f"cdh status {status}: {msg}")
