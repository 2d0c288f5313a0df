"""ctypes binding of libgdrf_hip.so (declarations: include/gdrf_hip.h).

The product path has no CPU fallback: if the shared library is missing the import of any
compute entry point fails loudly with the build command.
"""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "csrc", "libgdrf_hip.so")

# every symbol include/gdrf_hip.h declares: (restype, argtypes)
_vp, _i64, _dbl, _int = C.c_void_p, C.c_int64, C.c_double, C.c_int
SIGNATURES = {
    "gdrf_last_error": (C.c_char_p, []),
    "gdrf_version": (_int, []),
    "gdrf_ctx_create": (_int, [C.POINTER(_vp), _int, _i64, _int, _int, _int, _int, _int, _int]),
    "gdrf_ctx_create_ex": (_int, [C.POINTER(_vp), _int, _i64, _int, _int, _int, _int, _int, _int, _int]),
    "gdrf_stores_t": (_int, [_vp]),
    "gdrf_set_mfma_mode": (_int, [_vp, _int]),
    "gdrf_get_mfma_mode": (_int, [_vp]),
    "gdrf_set_hyper_backward": (_int, [_vp, _int]),
    "gdrf_get_hyper_backward": (_int, [_vp]),
    "gdrf_set_whiten": (_int, [_vp, _int]),
    "gdrf_set_mean": (_int, [_vp, _vp, _i64, _i64]),
    "gdrf_set_learn_inducing": (_int, [_vp, _int]),
    "gdrf_inducing_layout": (_int, [_vp, C.POINTER(_i64)]),
    "gdrf_ctx_destroy": (None, [_vp]),
    "gdrf_param_layout": (_int, [_vp, C.POINTER(_i64)]),
    "gdrf_red_layout": (_int, [_vp, C.POINTER(_i64)]),
    "gdrf_payload_pack": (_int, [_vp, _vp, _vp, _vp]),
    "gdrf_payload_unpack": (_int, [_vp, _vp, _vp, _vp]),
    "gdrf_set_allreduce": (_int, [_vp, _vp, _vp]),
    "gdrf_payload_allreduce": (_int, [_vp, _vp, _vp, _vp]),
    "gdrf_set_dirichlet": (_int, [_vp, C.POINTER(_dbl)]),
    "gdrf_knm": (_int, [_vp, _vp, _i64, _vp, _vp, _vp, _i64, _vp]),
    "gdrf_fill_eps": (_int, [_vp, C.c_uint64, C.c_uint32, _i64, _i64, _vp, _vp]),
    "gdrf_ll_const": (_int, [_vp, _vp, _i64, C.POINTER(_dbl), _vp]),
    "gdrf_ll_const_dev": (_int, [_vp, _vp, _i64, _vp, _vp]),
    "gdrf_probe": (_int, [_vp, _vp, _vp, C.POINTER(_dbl), _int, C.POINTER(_int), _vp]),
    "gdrf_probe_launch": (_int, [_vp, _vp, _vp, C.POINTER(_dbl), _int, _vp]),
    "gdrf_probe_read": (_int, [_vp, _int, C.POINTER(_int), _vp]),
    "gdrf_factorize": (_int, [_vp, _vp, _vp, _dbl, _vp]),
    "gdrf_factorize_mode": (_int, [_vp, _vp, _vp, _dbl, _vp, _int]),
    "gdrf_step_local": (_int, [_vp, _vp, _vp, _vp, _i64, _vp, _vp, _vp, _vp, _vp]),
    "gdrf_step_local_link": (_int, [_vp, _vp, _vp, _vp, _i64, _vp, _vp, _vp, _vp, _vp, _int, _vp, _i64]),
    "gdrf_step_local2": (_int, [_vp, _vp, _vp, _vp, _vp, _i64, _vp, _vp, _vp, _vp, _vp]),
    "gdrf_set_mean_guide": (_int, [_vp, _vp, _i64, _i64]),
    "gdrf_step_finish": (_int, [_vp, _vp, _vp, _vp, _vp, _dbl, _dbl, _vp, _vp, _vp]),
    "gdrf_adam": (_int, [_vp, _int, _vp, _vp, _vp, _vp, _i64, _dbl, _dbl, _dbl, _dbl, _dbl, _dbl, _vp]),
    "gdrf_predict": (_int, [_vp, _vp, _i64, _vp, _vp, _vp, _int, _vp, _vp, _vp]),
    "gdrf_chol_failed": (_int, [_vp, C.POINTER(_int), _vp]),
    "gdrf_ws_ptr": (_int, [_vp, _int, C.POINTER(_vp), C.POINTER(_i64)]),
    "gdrf_ws_elem_size": (_int, [_vp, _int]),
    "gdrf_ws_copy": (_int, [_vp, _int, _vp, _i64, _vp]),
    "gdrf_set_timing": (_int, [_vp, _int]),
    "gdrf_get_timing": (_int, [_vp, C.POINTER(_dbl), C.POINTER(_i64), _int]),
}

ALLREDUCE_FN = C.CFUNCTYPE(_int, _vp, _i64, _int, _vp, _vp)       # gdrf_allreduce_fn of include/gdrf_hip.h

_lib = None


class GdrfHipError(RuntimeError):
    pass


def load():
    """Load the shared library (once) and declare every signature."""
    global _lib
    if _lib is not None:
        return _lib
    # torch must be imported first: libgdrf_hip.so needs libamdhip64.so.7, and the process must end up with ONE HIP
    # runtime (the one PyTorch-ROCm bundles and has already loaded), not a second copy from /opt/rocm.
    import torch  # noqa: F401
    if not os.path.exists(LIB_PATH):
        raise GdrfHipError(
            f"{LIB_PATH} is missing: the HIP extension is not built. Run "
            "`python -c 'import __graft_entry__ as g; g.build()'` or `make -C gdrf_amd/csrc` (needs hipcc, gfx950)."
        )
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)          # AttributeError if a declared symbol is not exported
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def check(rc: int, what: str):
    if rc != 0:
        msg = load().gdrf_last_error()
        raise GdrfHipError(f"{what} failed (code {rc}): {msg.decode() if msg else ''}")
