"""ctypes binding of libepi_gfx950.so (the C ABI declared in include/epi.h).

The library is the product: there is no CPU fallback.  Importing this module
succeeds without the shared object so that host-only code can run, but any
attempt to use the device path without it raises EpiLibraryError.
"""
from __future__ import annotations

import ctypes as C
import os

EPI_MAX_H = 64
EPI_MAX_D = 7

EPI_F_NEG_I = 1
EPI_F_NEG_NT = 2
EPI_F_BAD_DENOM = 4
EPI_F_ZERO_W = 8
EPI_F_COARSE_W = 16
COARSE_W_BOUND = 1e-3      # EPI_F_COARSE_W: n_total / (fixed-point weight total) above this

_HERE = os.path.dirname(os.path.abspath(__file__))
# EPI_GFX950_LIB points at another build of the same library (kernel A/B measurements); there is still no fallback.
LIB_PATH = os.environ.get("EPI_GFX950_LIB") or os.path.join(_HERE, "csrc", "libepi_gfx950.so")


class EpiLibraryError(RuntimeError):
    pass


class EpiError(RuntimeError):
    def __init__(self, code: int, msg: str):
        super().__init__(f"libepi_gfx950 error {code}: {msg}")
        self.code = code


class epi_params(C.Structure):
    _fields_ = [
        ("H", C.c_int32), ("D", C.c_int32), ("N", C.c_int64),
        ("inf_loss", C.c_double), ("test_loss", C.c_double), ("testrate_loss", C.c_double),
        ("q", C.c_double * EPI_MAX_H), ("p", C.c_double * EPI_MAX_H),
        ("k", C.c_double * EPI_MAX_H), ("theta", C.c_double * EPI_MAX_H),
    ]


_i64p = C.POINTER(C.c_int64)
_u64p = C.POINTER(C.c_uint64)
_f64p = C.POINTER(C.c_double)
_u32p = C.POINTER(C.c_uint32)
_vp = C.c_void_p

# name -> (restype, argtypes); every symbol include/epi.h declares
SIGNATURES = {
    "epi_ctx_create": (C.c_int, [C.c_int, C.POINTER(_vp)]),
    "epi_ctx_destroy": (C.c_int, [_vp]),
    "epi_params_set": (C.c_int, [_vp, C.POINTER(epi_params)]),
    "epi_last_error": (C.c_char_p, []),
    "epi_state_rows": (C.c_int, [C.c_int, C.c_int]),
    "epi_belief_create": (C.c_int, [_vp, C.c_int64, C.c_int64, C.POINTER(_vp)]),
    "epi_belief_destroy": (C.c_int, [_vp]),
    "epi_belief_set_shard": (C.c_int, [_vp, C.c_int64, C.c_int64]),
    "epi_belief_size": (C.c_int64, [_vp]),
    "epi_belief_upload": (C.c_int, [_vp, _i64p, _i64p, _i64p, _i64p]),
    "epi_belief_download": (C.c_int, [_vp, _i64p, _i64p, _i64p, _i64p]),
    "epi_belief_copy": (C.c_int, [_vp, _vp]),
    "epi_belief_init_normal": (C.c_int, [_vp, C.c_double, C.c_double, C.c_uint64]),
    "epi_belief_init_simplex": (C.c_int, [_vp, C.c_uint64]),
    "epi_step_batch": (C.c_int, [_vp, _vp, _f64p, C.c_int, _f64p, C.c_int, C.c_uint64, C.c_uint32,
                                 _i64p, _f64p, _i64p, _u32p]),
    "epi_step_injected": (C.c_int, [_vp, _vp, _f64p, C.c_int, _i64p, _i64p, _f64p, _i64p, _i64p, _u32p]),
    "epi_step_variates": (C.c_int, [_vp, _vp, C.c_double, C.c_uint64, C.c_uint32, _i64p, _i64p, _f64p, _u32p]),
    "epi_sample_binomial": (C.c_int, [_vp, C.c_int64, _i64p, _f64p, C.c_uint64, C.c_uint32, C.c_uint32, C.c_int, _i64p]),
    "epi_sample_gamma": (C.c_int, [_vp, C.c_int64, _f64p, _f64p, C.c_uint64, C.c_uint32, C.c_uint32, _f64p]),
    "epi_rollout_batch": (C.c_int, [_vp, _vp, C.c_int, _f64p, C.c_int, C.c_int, C.c_double, C.c_uint64, C.c_uint32,
                                    _i64p, _f64p, _u32p]),
    "epi_rollout_score": (C.c_int, [_vp, _vp, C.c_int64, C.c_int, C.c_int, _f64p, C.c_double, C.c_double, C.c_uint64, C.c_uint32,
                                    _f64p, _u32p]),
    "epi_ensemble_simulate": (C.c_int, [_vp, _vp, C.c_int, _f64p, C.c_int, C.c_int, C.c_double, C.c_uint64, C.c_uint32,
                                        C.POINTER(_vp), _u32p]),
    "epi_ensemble_destroy": (C.c_int, [_vp]),
    "epi_ensemble_shape": (C.c_int, [_vp, _i64p, C.POINTER(C.c_int)]),
    "epi_ensemble_download": (C.c_int, [_vp, _i64p]),
    "epi_ensemble_loss": (C.c_int, [_vp, _vp, C.c_int, _f64p, C.c_int, C.c_int, _f64p, _f64p]),
    "epi_pf_update": (C.c_int, [_vp, _vp, C.c_double, C.c_int64, C.c_uint64, C.c_uint32, _f64p, _f64p, _u32p]),
    "epi_pf_update_noalloc": (C.c_int, [_vp, _vp, C.c_double, C.c_int64, C.c_uint64, C.c_uint32, _f64p, _f64p, _u32p]),
    "epi_closed_loop_days": (C.c_int, [_vp, _vp, _vp, C.c_int, _f64p, C.c_double, C.c_uint64, C.c_uint32, C.c_uint64, C.c_uint32,
                                       C.c_int, _f64p, _i64p, _f64p, _i64p, _f64p, _f64p, _u32p, _i64p, _i64p]),
    "epi_mctx_create": (C.c_int, [C.POINTER(C.c_int), C.c_int, C.POINTER(epi_params), C.c_int64, C.c_int, C.POINTER(_vp)]),
    "epi_mctx_destroy": (C.c_int, [_vp]),
    "epi_mctx_shards": (C.c_int, [_vp]),
    "epi_mctx_size": (C.c_int64, [_vp]),
    "epi_mctx_init_normal": (C.c_int, [_vp, C.c_double, C.c_double, C.c_uint64]),
    "epi_mctx_upload": (C.c_int, [_vp, _i64p, _i64p, _i64p, _i64p]),
    "epi_mctx_download": (C.c_int, [_vp, _i64p, _i64p, _i64p, _i64p]),
    "epi_mpf_update": (C.c_int, [_vp, C.c_double, C.c_int64, C.c_uint64, C.c_uint32, _f64p, _f64p, _u32p, _i64p]),
    "epi_mctx_set_overlap": (C.c_int, [_vp, C.c_int]),
    "epi_mctx_overlapped_updates": (C.c_int64, [_vp]),
    "epi_mctx_sums": (C.c_int, [_vp, _i64p]),
    "epi_mctx_mean": (C.c_int, [_vp, _i64p, _i64p, _i64p, _i64p]),
    "epi_pf_frac_bits": (C.c_int, [C.c_int64]),
    "epi_pf_predict": (C.c_int, [_vp, _vp, C.c_double, C.c_int64, C.c_uint64, C.c_uint32, _f64p]),
    "epi_pf_weigh": (C.c_int, [_vp, _vp, C.c_double, _u64p, _f64p, _u32p]),
    "epi_pf_predict_dev": (C.c_int, [_vp, _vp, C.c_double, C.c_int64, C.c_uint64, C.c_uint32, _vp]),
    "epi_pf_weigh_dev": (C.c_int, [_vp, _vp, _vp, _vp]),
    "epi_ctx_set_stream": (C.c_int, [_vp, _vp, C.c_int]),
    "epi_pf_plan": (C.c_int, [C.c_int64, C.c_int, _u64p, C.c_uint64, C.c_uint32, _i64p]),
    "epi_pf_select_local": (C.c_int, [_vp, _vp, C.c_int, C.c_int, _u64p, C.c_uint64, C.c_uint32, C.c_int64, C.c_int64]),
    "epi_pf_export": (C.c_int, [_vp, _vp, C.c_int, C.c_int, _u64p, C.c_uint64, C.c_uint32, C.c_int64, C.c_int64, _vp]),
    "epi_pf_import": (C.c_int, [_vp, _vp, C.c_int64, C.c_int64, _vp]),
    "epi_pf_import_late": (C.c_int, [_vp, _vp, C.c_int64, C.c_int64, _vp]),
    "epi_pf_predict_part_dev": (C.c_int, [_vp, _vp, C.c_double, C.c_int64, C.c_uint64, C.c_uint32, C.c_int64, C.c_int64, _vp]),
    "epi_pf_finish": (C.c_int, [_vp, _vp, _u32p]),
    "epi_belief_sums": (C.c_int, [_vp, _vp, _i64p]),
    "epi_belief_mean": (C.c_int, [_vp, _vp, _i64p, _i64p, _i64p, _i64p]),
    "epi_mean_from_sums": (C.c_int, [C.POINTER(epi_params), C.c_int64, _i64p, _i64p, _i64p, _i64p, _i64p]),
    "epi_belief_suminf_quantiles": (C.c_int, [_vp, _vp, _f64p, C.c_int, _i64p]),
    "epi_belief_sample": (C.c_int, [_vp, _vp, C.c_uint64, C.c_uint64, _i64p, _i64p, _i64p, _i64p]),
    "epi_belief_sample_many": (C.c_int, [_vp, _vp, C.c_uint64, C.c_uint64, C.c_int64, _i64p, _i64p, _i64p, _i64p]),
    "epi_ctx_stream": (_vp, [_vp]),
    "epi_pf_update_async": (C.c_int, [_vp, _vp, C.c_double, C.c_int64, C.c_uint64, C.c_uint32]),
    "epi_pf_collect": (C.c_int, [_vp, _vp, _f64p, _f64p, _u32p]),
    "epi_ctx_timing": (C.c_int, [_vp, C.c_int]),
    "epi_ctx_timing_read": (C.c_int, [_vp, _f64p, _i64p]),
    "epi_last_predict_ms": (C.c_float, [_vp]),
}

# the extra entry points of the testing build (include/epi_testing.h, libepi_gfx950_testing.so)
TESTING_SIGNATURES = {
    "epi_debug_screen": (C.c_int, [C.c_int]),
    "epi_debug_step_impl": (C.c_int, [C.c_int]),
    "epi_debug_node_exchange": (C.c_int, [C.c_int]),
    "epi_debug_f32_forms": (C.c_int, [_vp, C.c_int, C.c_float, C.c_float, C.POINTER(C.c_uint64)]),
    "epi_debug_btrs_quick": (C.c_int, [_vp, C.c_uint64, C.c_int32, C.c_uint32, C.c_double, C.c_int, C.POINTER(C.c_uint64)]),
}
TESTING_LIB_PATH = os.path.join(_HERE, "csrc", "libepi_gfx950_testing.so")

_lib = None
_testing_lib = None


def load():
    """Load the shared library (once) and set the prototypes."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise EpiLibraryError(
            f"{LIB_PATH} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(hipcc --offload-arch=gfx950).  There is no CPU fallback.")
    try:
        lib = C.CDLL(LIB_PATH)
    except OSError as e:
        raise EpiLibraryError(f"cannot load {LIB_PATH}: {e}") from e
    _set_prototypes(lib, SIGNATURES)
    _lib = lib
    return lib


def _set_prototypes(lib, table):
    for name, (res, args) in table.items():
        fn = getattr(lib, name)
        fn.restype = res
        fn.argtypes = args


def load_testing():
    """The build of the library with the switches of include/epi_testing.h (the test suite's own; never the default)."""
    global _testing_lib
    if _testing_lib is None:
        if not os.path.exists(TESTING_LIB_PATH):
            raise EpiLibraryError(f"{TESTING_LIB_PATH} not found: build it with `make -C epidemicmodeling_amd/csrc all`")
        lib = C.CDLL(TESTING_LIB_PATH)
        _set_prototypes(lib, SIGNATURES)
        _set_prototypes(lib, TESTING_SIGNATURES)
        _testing_lib = lib
    return _testing_lib


def check(rc: int):
    if rc != 0:
        raise EpiError(rc, load().epi_last_error().decode("utf-8", "replace"))


def as_i64p(a):
    return a.ctypes.data_as(_i64p) if a is not None else None


def as_f64p(a):
    return a.ctypes.data_as(_f64p) if a is not None else None


def as_u64p(a):
    return a.ctypes.data_as(_u64p) if a is not None else None
