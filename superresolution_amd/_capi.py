"""ctypes binding of libmmc_hip.so (include/mmc_hip.h).

This is the stand-in for the MEX gateway: it hands raw column-major float32
buffers to the C ABI exactly as mex/mmc_hip_mex.cpp does.  There is no CPU
fallback: if the library is missing or fails to load, importing the solver
raises.
"""
import ctypes as C
import os

import numpy as np

from .build import LIB_PATH

MMC_ABI_VERSION = 4
MMC_NUM_SUMS = 4
MMC_NUM_DIAG = 9
DIAG_OFF, DIAG_AUTO, DIAG_ALWAYS = 0, 1, 2
MMC_STATIC_MAX_BLOCKS = 524288

FLOW_FORWARD, FLOW_BACKWARD, FLOW_FORWARD_BACKWARD = 0, 1, 2
SAMPLING_AVERAGE, SAMPLING_STRIDE, SAMPLING_TAPS = 0, 1, 2

STATUS_NAMES = {0: "MMC_OK", -1: "MMC_ERR_INVALID_ARG", -2: "MMC_ERR_STATE", -3: "MMC_ERR_HIP",
                -4: "MMC_ERR_UNSUPPORTED", -5: "MMC_ERR_NO_DEVICE"}


class MMCError(RuntimeError):
    def __init__(self, status, message):
        super().__init__("%s: %s" % (STATUS_NAMES.get(status, status), message))
        self.status = status
        self.message = message


class mmc_params(C.Structure):
    _fields_ = [("size", C.c_uint32), ("version", C.c_uint32),
                ("ny", C.c_int32), ("nx", C.c_int32), ("factor", C.c_int32),
                ("nc_global", C.c_int32), ("frame_begin", C.c_int32), ("frame_end", C.c_int32),
                ("alpha", C.c_float), ("kappa", C.c_float), ("h", C.c_float),
                ("flow_direction", C.c_int32), ("sampling", C.c_int32), ("device", C.c_int32),
                ("stream", C.c_void_p)]


class mmc_sf_params(C.Structure):
    _fields_ = [("size", C.c_uint32), ("version", C.c_uint32),
                ("ny", C.c_int32), ("nx", C.c_int32), ("factor", C.c_int32), ("num_frames", C.c_int32),
                ("alpha", C.c_float), ("huber_eps", C.c_float),
                ("sampling", C.c_int32), ("device", C.c_int32), ("stream", C.c_void_p)]


class mmc_stats(C.Structure):
    _fields_ = [("iterations", C.c_int32), ("converged", C.c_int32),
                ("res_primal", C.c_double), ("res_dual", C.c_double), ("solve_seconds", C.c_double),
                ("primal_energy", C.c_double), ("dual_energy", C.c_double), ("gap", C.c_double),
                ("dual_infeasibility", C.c_double), ("opt_res_primal", C.c_double), ("opt_res_dual", C.c_double),
                ("norm_KTy", C.c_double), ("norm_Kx", C.c_double), ("step_balance", C.c_double)]

    def as_dict(self):
        """Every field; the diagnostics group is NaN when it was not evaluated."""
        d = {name: getattr(self, name) for name, _ in self._fields_}
        d["converged"] = bool(d["converged"])
        d["seconds"] = d.pop("solve_seconds")
        return d


class mmc_halo(C.Structure):
    _fields_ = [("frame_bytes", C.c_uint64),
                ("send_prev", C.c_void_p * 2), ("recv_next", C.c_void_p * 2),
                ("send_next", C.c_void_p * 2), ("recv_prev", C.c_void_p * 2)]


PROGRESS_CB = C.CFUNCTYPE(C.c_int, C.c_int, C.POINTER(mmc_stats), C.c_void_p)

_fp = C.POINTER(C.c_float)
_dp = C.POINTER(C.c_double)
_u8p = C.POINTER(C.c_uint8)
_H = C.c_void_p

# name -> (restype, argtypes); every symbol include/mmc_hip.h declares
SIGNATURES = {
    "mmc_abi_version": (C.c_int, []),
    "mmc_hip_versions": (C.c_int, [C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    "mmc_last_error": (C.c_char_p, [_H]),
    "mmc_create": (C.c_int, [C.POINTER(mmc_params), C.POINTER(_H)]),
    "mmc_destroy": (C.c_int, [_H]),
    "mmc_set_lowres": (C.c_int, [_H, _fp]),
    "mmc_set_blur": (C.c_int, [_H, _fp, C.c_int, C.c_int]),
    "mmc_set_flow": (C.c_int, [_H, _fp, _u8p]),
    "mmc_set_sampling_taps": (C.c_int, [_H, C.POINTER(C.c_int32), _fp, C.c_int, C.POINTER(C.c_int32), _fp, C.c_int]),
    "mmc_num_flow_pairs": (C.c_int, [_H]),
    "mmc_set_state": (C.c_int, [_H, _fp, _fp]),
    "mmc_num_state_frames": (C.c_int, [_H]),
    "mmc_set_fixed_frame": (C.c_int, [_H, _fp, _fp]),
    "mmc_set_flexbox_backward_terms": (C.c_int, [_H, C.c_int]),
    "mmc_init": (C.c_int, [_H]),
    "mmc_h_partials": (C.c_int, [_H, _dp]),
    "mmc_set_h": (C.c_int, [_H, C.c_float]),
    "mmc_get_h": (C.c_int, [_H, _fp]),
    "mmc_h_partials_static": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int,
                                        C.c_void_p, C.c_void_p]),
    "mmc_solve": (C.c_int, [_H, C.c_int, C.c_float, C.c_int, C.POINTER(mmc_stats), PROGRESS_CB, C.c_void_p]),
    "mmc_sweep_primal": (C.c_int, [_H, C.c_int, C.c_int, C.c_int]),
    "mmc_sweep_dual": (C.c_int, [_H, C.c_int, C.c_int, C.c_int]),
    "mmc_sums_reset": (C.c_int, [_H]),
    "mmc_sums_read": (C.c_int, [_H, _dp]),
    "mmc_halo_info": (C.c_int, [_H, C.POINTER(mmc_halo)]),
    "mmc_sync": (C.c_int, [_H]),
    "mmc_set_diagnostics": (C.c_int, [_H, C.c_int]),
    "mmc_diag_partials": (C.c_int, [_H, _dp]),
    "mmc_state_halo_info": (C.c_int, [_H, C.POINTER(mmc_halo)]),
    "mmc_problem_size": (C.c_int, [_H, _dp, _dp]),
    "mmc_stats_from_diag": (C.c_int, [_dp, C.c_double, C.c_double, C.c_float, C.POINTER(mmc_stats)]),
    "mmc_group_set_diagnostics": (C.c_int, [_H, C.c_int]),
    "mmc_set_step_balance": (C.c_int, [_H, C.c_float]),
    "mmc_get_step_balance": (C.c_int, [_H, _fp]),
    "mmc_set_step_adaptation": (C.c_int, [_H, C.c_int]),
    "mmc_step_sums_read": (C.c_int, [_H, _dp]),
    "mmc_step_balance_rule": (C.c_int, [_dp, _dp, _dp]),
    "mmc_group_set_step_balance": (C.c_int, [_H, C.c_float]),
    "mmc_group_set_step_adaptation": (C.c_int, [_H, C.c_int]),
    "mmc_set_profiling": (C.c_int, [_H, C.c_int]),
    "mmc_get_profile": (C.c_int, [_H, _dp, C.POINTER(C.c_int)]),
    "mmc_get_sweep_profile": (C.c_int, [_H, _dp, C.POINTER(C.c_int)]),
    "mmc_get_state": (C.c_int, [_H, _fp, _fp]),
    "mmc_energy": (C.c_int, [_H, _dp]),
    "mmc_group_create": (C.c_int, [C.POINTER(mmc_params), C.c_int, C.POINTER(C.c_int32), C.POINTER(_H)]),
    "mmc_group_destroy": (C.c_int, [_H]),
    "mmc_group_last_error": (C.c_char_p, [_H]),
    "mmc_group_num_blocks": (C.c_int, [_H]),
    "mmc_group_block_range": (C.c_int, [_H, C.c_int, C.POINTER(C.c_int32), C.POINTER(C.c_int32), C.POINTER(C.c_int32)]),
    "mmc_group_block_handle": (_H, [_H, C.c_int]),
    "mmc_group_set_lowres": (C.c_int, [_H, _fp]),
    "mmc_group_set_blur": (C.c_int, [_H, _fp, C.c_int, C.c_int]),
    "mmc_group_set_sampling_taps": (C.c_int, [_H, C.POINTER(C.c_int32), _fp, C.c_int, C.POINTER(C.c_int32), _fp, C.c_int]),
    "mmc_group_set_flow": (C.c_int, [_H, _fp, _u8p]),
    "mmc_group_set_state": (C.c_int, [_H, _fp, _fp]),
    "mmc_group_set_fixed_frame": (C.c_int, [_H, _fp, _fp]),
    "mmc_group_set_flexbox_backward_terms": (C.c_int, [_H, C.c_int]),
    "mmc_group_init": (C.c_int, [_H]),
    "mmc_group_get_h": (C.c_int, [_H, _fp]),
    "mmc_group_set_h": (C.c_int, [_H, C.c_float]),
    "mmc_group_solve": (C.c_int, [_H, C.c_int, C.c_float, C.c_int, C.POINTER(mmc_stats), PROGRESS_CB, C.c_void_p]),
    "mmc_group_get_state": (C.c_int, [_H, _fp, _fp]),
    "mmc_debug_apply_K": (C.c_int, [_H, _fp, _fp, _fp, _fp, _fp]),
    "mmc_debug_apply_KT": (C.c_int, [_H, _fp, _fp, _fp, _fp, _fp, _fp, _fp]),
    "mmc_debug_steps": (C.c_int, [_H, _fp, _fp, _fp, _fp, _fp]),
    "mmc_debug_steps_update": (C.c_int, [_H, _fp, _fp]),
    "mmc_debug_plane_skew": (C.c_size_t, [_H]),
    "mmc_debug_separable": (C.c_int, [_H]),
    "mmc_sf_create": (C.c_int, [C.POINTER(mmc_sf_params), C.POINTER(_H)]),
    "mmc_sf_destroy": (C.c_int, [_H]),
    "mmc_sf_last_error": (C.c_char_p, [_H]),
    "mmc_sf_set_lowres": (C.c_int, [_H, _fp]),
    "mmc_sf_set_blur": (C.c_int, [_H, _fp, C.c_int, C.c_int]),
    "mmc_sf_set_flow": (C.c_int, [_H, _fp, _u8p]),
    "mmc_sf_set_state": (C.c_int, [_H, _fp]),
    "mmc_sf_init": (C.c_int, [_H]),
    "mmc_sf_solve": (C.c_int, [_H, C.c_int, C.c_float, C.c_int, C.POINTER(mmc_stats), PROGRESS_CB, C.c_void_p]),
    "mmc_sf_get_state": (C.c_int, [_H, _fp]),
    "mmc_sf_debug_apply_K": (C.c_int, [_H, _fp, _fp, _fp]),
    "mmc_sf_debug_apply_KT": (C.c_int, [_H, _fp, _fp, _fp]),
    "mmc_sf_debug_steps": (C.c_int, [_H, _fp, _fp, _fp]),
}

_lib = None


def load_library(path=None):
    """dlopen libmmc_hip.so and attach the prototypes.  Raises if it is absent."""
    global _lib
    if _lib is not None and path is None:
        return _lib
    path = path or os.environ.get("MMC_HIP_LIB") or LIB_PATH
    if not os.path.exists(path):
        raise ImportError("libmmc_hip.so not found at %s -- build it with "
                          "`python -m superresolution_amd.build` (there is no CPU fallback)" % path)
    # One process can hold one HIP runtime.  PyTorch-ROCm ships its own copy under the same
    # SONAME; loaded after the system one it finds no GPU.  Streams and device pointers are
    # shared with torch (sharding, bench), so let torch bring in the runtime first.
    try:
        import torch  # noqa: F401
    except ImportError:
        pass
    lib = C.CDLL(path)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)      # AttributeError if a declared symbol is missing
        fn.restype = res
        fn.argtypes = args
    if lib.mmc_abi_version() != MMC_ABI_VERSION:
        raise ImportError("libmmc_hip.so ABI version mismatch")
    built, runtime = hip_versions(lib)
    if runtime and built // 100000 != runtime // 100000:
        import warnings
        warnings.warn("libmmc_hip.so was built against HIP %s and runs on the HIP %s runtime that was loaded first in "
                      "this process (PyTorch-ROCm brings its own copy)" % (_fmt_hip(built), _fmt_hip(runtime)), RuntimeWarning)
    _lib = lib
    return lib


def _fmt_hip(v):
    return "%d.%d.%d" % (v // 10000000, (v // 100000) % 100, v % 100000)


def hip_versions(lib=None):
    """(HIP version the library was compiled against, HIP runtime version of this process; 0 if unknown)."""
    lib = lib or load_library()
    b, r = C.c_int(0), C.c_int(0)
    lib.mmc_hip_versions(C.byref(b), C.byref(r))
    return int(b.value), int(r.value)


def as_f32(x):
    """Column-major float32 copy of x (what the MEX gateway hands over)."""
    return np.asfortranarray(np.asarray(x), dtype=np.float32)


def fptr(a):
    return None if a is None else a.ctypes.data_as(_fp)
