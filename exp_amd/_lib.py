"""ctypes binding of libexp_amd.so (the C ABI declared in include/exp_amd.h).

There is no CPU fallback: if the shared library is missing, or no HIP device is
usable, every entry point raises.  The library must be built in-tree
(``make -j8`` or ``__graft_entry__.build()``)."""
from __future__ import annotations

import ctypes
import os
from ctypes import (POINTER, c_char_p, c_double, c_int, c_int32, c_longlong, c_size_t, c_uint,
                    c_void_p)

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("EXP_AMD_LIB") or os.path.join(_HERE, "libexp_amd.so")   # (override: A/B builds)

c_double_p = POINTER(c_double)


class ExpAmdError(RuntimeError):
    pass


class SphConfig(ctypes.Structure):
    """exp_amd_sph_config (include/exp_amd.h)"""
    _fields_ = [("lmax", c_int), ("nmax", c_int), ("numr", c_int), ("cmap", c_int),
                ("rmap", c_double), ("scale", c_double), ("rmin", c_double), ("rmax", c_double),
                ("xmin", c_double), ("dxi", c_double),
                ("NO_L0", c_int), ("NO_L1", c_int), ("EVEN_L", c_int), ("EVEN_M", c_int),
                ("M0_only", c_int), ("multistep", c_int)]


class CylConfig(ctypes.Structure):
    """exp_amd_cyl_config (include/exp_amd.h)"""
    _fields_ = [("mmax", c_int), ("nmax", c_int), ("numx", c_int), ("numy", c_int),
                ("cmapr", c_int), ("cmapz", c_int), ("ascale", c_double), ("hscale", c_double),
                ("rtable", c_double), ("xmin", c_double), ("dx", c_double), ("ymin", c_double),
                ("dy", c_double), ("rcylmax", c_double), ("EVEN_M", c_int), ("multistep", c_int)]


ALLREDUCE_FN = ctypes.CFUNCTYPE(c_int, c_void_p, c_size_t, c_void_p, c_void_p)

# name -> (restype, argtypes); every symbol the header declares
SIGNATURES = {
    "exp_amd_abi_version": (c_int, []),
    "exp_amd_last_error": (c_char_p, [c_void_p]),
    "exp_amd_last_global_error": (c_char_p, []),
    "exp_amd_ctx_create": (c_int, [c_int, c_void_p, POINTER(c_void_p)]),
    "exp_amd_ctx_destroy": (None, [c_void_p]),
    "exp_amd_ctx_synchronize": (c_int, [c_void_p]),
    "exp_amd_ctx_stream": (c_void_p, [c_void_p]),
    "exp_amd_comm_get_unique_id": (c_int, [c_void_p]),
    "exp_amd_comm_init_rank": (c_int, [c_void_p, c_void_p, c_int, c_int]),
    "exp_amd_comm_set_callback": (c_int, [c_void_p, ALLREDUCE_FN, c_void_p]),
    "exp_amd_comm_set_world": (c_int, [c_void_p, c_int, c_int]),
    "exp_amd_comp_set_rtrunc": (c_int, [c_void_p, c_double, c_void_p]),
    "exp_amd_force_set_mass_scale": (c_int, [c_void_p, c_double]),
    "exp_amd_force_set_self_consistent": (c_int, [c_void_p, c_int]),
    "exp_amd_force_set_initializing": (c_int, [c_void_p, c_int]),
    "exp_amd_force_coefs_frozen": (c_int, [c_void_p]),
    "exp_amd_sph_set_fix_l0": (c_int, [c_void_p, c_int]),
    "exp_amd_sph_set_subset": (c_int, [c_void_p, c_double, c_int]),
    "exp_amd_sph_set_noise": (c_int, [c_void_p, c_void_p, c_void_p, c_double, c_uint]),
    "exp_amd_cyl_set_mlim": (c_int, [c_void_p, c_int]),
    "exp_amd_sim_set_adiabatic": (c_int, [c_void_p, c_int, c_double, c_double, c_double]),
    "exp_amd_sim_set_time": (c_int, [c_void_p, c_double]),
    "exp_amd_comm_info": (c_int, [c_void_p, POINTER(c_int), POINTER(c_int), POINTER(c_int), POINTER(c_longlong)]),
    "exp_amd_comm_streams": (c_int, [c_void_p]),
    "exp_amd_comm_allreduce_max": (c_int, [c_void_p, POINTER(c_double)]),
    "exp_amd_comm_allreduce": (c_int, [c_void_p, c_void_p, c_size_t]),
    "exp_amd_comp_create": (c_int, [c_void_p, c_size_t, POINTER(c_void_p)]),
    "exp_amd_comp_destroy": (None, [c_void_p]),
    "exp_amd_comp_size": (c_size_t, [c_void_p]),
    "exp_amd_comp_upload": (c_int, [c_void_p] + [c_void_p] * 7),
    "exp_amd_comp_upload_frame": (c_int, [c_void_p] + [c_void_p] * 7 + [c_int, c_void_p, c_void_p]),
    "exp_amd_comp_upload_acc": (c_int, [c_void_p] + [c_void_p] * 4),
    "exp_amd_comp_upload_levels": (c_int, [c_void_p, c_void_p]),
    "exp_amd_comp_download": (c_int, [c_void_p] + [c_void_p] * 11),
    "exp_amd_comp_download_levels": (c_int, [c_void_p, c_void_p]),
    "exp_amd_comp_upload_device": (c_int, [c_void_p] + [c_void_p] * 7),
    "exp_amd_comp_set_center": (c_int, [c_void_p, c_double_p]),
    "exp_amd_comp_get_center": (c_int, [c_void_p, c_void_p]),
    "exp_amd_comp_drift": (c_int, [c_void_p, c_double, c_int]),
    "exp_amd_comp_kick": (c_int, [c_void_p, c_double, c_int]),
    "exp_amd_comp_zero_acc": (c_int, [c_void_p, c_int]),
    "exp_amd_sph_create": (c_int, [c_void_p, POINTER(SphConfig), c_void_p, c_void_p, c_void_p,
                                   c_void_p, POINTER(c_void_p)]),
    "exp_amd_comp_fix_positions": (c_int, [c_void_p, c_int, c_void_p]),
    "exp_amd_comp_set_consp": (c_int, [c_void_p, c_int, c_double]),
    "exp_amd_comp_set_level_policy": (c_int, [c_void_p, c_int, c_int, c_int]),
    "exp_amd_comp_get_escaped": (c_int, [c_void_p, c_void_p]),
    "exp_amd_comp_set_escaped": (c_int, [c_void_p, c_void_p]),
    "exp_amd_comp_log_sums": (c_int, [c_void_p, c_void_p]),
    "exp_amd_orient_create": (c_int, [c_void_p, c_int, c_int, c_uint, c_uint, c_double, c_double,
                                      POINTER(c_void_p)]),
    "exp_amd_ctx_set_split_min": (c_int, [c_void_p, c_longlong]),
    "exp_amd_ctx_set_append_min": (c_int, [c_void_p, c_longlong]),
    "exp_amd_ctx_set_append_lean": (c_int, [c_void_p, c_int]),
    "exp_amd_ctx_set_dense_min": (c_int, [c_void_p, c_longlong]),
    "exp_amd_ctx_set_thin_max": (c_int, [c_void_p, c_longlong]),
    "exp_amd_ctx_set_mover_list_min": (c_int, [c_void_p, c_longlong]),
    "exp_amd_ctx_set_deterministic": (c_int, [c_void_p, c_int]),
    "exp_amd_ctx_set_prekick": (c_int, [c_void_p, c_int]),
    "exp_amd_comp_set_orientation": (c_int, [c_void_p, c_void_p]),
    "exp_amd_orient_flags": (c_uint, [c_void_p]),
    "exp_amd_orient_set_naccel": (c_int, [c_void_p, c_int]),
    "exp_amd_orient_accel": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p]),
    "exp_amd_comp_set_pseudo_accel": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p]),
    "exp_amd_orient_destroy": (None, [c_void_p]),
    "exp_amd_orient_open_log": (c_int, [c_void_p, c_char_p, c_uint, c_double, c_double, c_int, c_void_p]),
    "exp_amd_orient_log_entry": (c_int, [c_void_p, c_double, c_void_p, c_void_p]),
    "exp_amd_sim_set_restart": (c_int, [c_void_p, c_int]),
    "exp_amd_sim_set_eqmotion": (c_int, [c_void_p, c_int]),
    "exp_amd_sim_set_center_from": (c_int, [c_void_p, c_int, c_int]),
    "exp_amd_sim_set_orient": (c_int, [c_void_p, c_int, c_void_p, c_int, c_int]),
    "exp_amd_orient_set_center": (c_int, [c_void_p, c_void_p]),
    "exp_amd_orient_set_cenvel": (c_int, [c_void_p, c_void_p]),
    "exp_amd_orient_set_linear": (c_int, [c_void_p]),
    "exp_amd_orient_accumulate": (c_int, [c_void_p, c_double, c_double, c_void_p]),
    "exp_amd_orient_get": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
    "exp_amd_sph_set_exterior": (c_int, [c_void_p, c_int]),
    "exp_amd_sph_set_accumulate_all_m": (c_int, [c_void_p, c_int]),
    "exp_amd_host_binsum_f32": (c_int, [ctypes.c_longlong, c_void_p, c_void_p, c_int, c_void_p]),
    "exp_amd_host_psp_unpack": (c_int, [ctypes.c_longlong, c_void_p, ctypes.c_longlong, c_int, c_int, c_int, c_int,
                                        c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
    "exp_amd_sph_set_dsmall": (c_int, [c_void_p, c_double]),
    "exp_amd_sph_set_density": (c_int, [c_void_p, c_void_p]),
    "exp_amd_cyl_cov_enable": (c_int, [c_void_p, c_int]),
    "exp_amd_cyl_cov_reset": (c_int, [c_void_p]),
    "exp_amd_cyl_cov_accumulate": (c_int, [c_void_p, c_void_p, c_void_p, POINTER(c_longlong)]),
    "exp_amd_cyl_cov_get": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
    "exp_amd_sph_cov_enable": (c_int, [c_void_p, c_int]),
    "exp_amd_sph_cov_reset": (c_int, [c_void_p]),
    "exp_amd_sph_cov_accumulate": (c_int, [c_void_p, c_void_p, c_longlong, POINTER(c_longlong)]),
    "exp_amd_sph_cov_get": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
    "exp_amd_cyl_set_density": (c_int, [c_void_p, c_void_p]),
    "exp_amd_cyl_fields": (c_int, [c_void_p, c_size_t, c_void_p, c_void_p, c_void_p, c_int, c_void_p]),
    "exp_amd_sph_fields": (c_int, [c_void_p, c_size_t, c_void_p, c_void_p, c_void_p, c_int, c_void_p]),
    "exp_amd_sph_basis": (c_int, [c_void_p, c_size_t, c_void_p, c_void_p]),
    "exp_amd_sph_window_mass": (c_int, [c_void_p, c_void_p, POINTER(c_double)]),
    "exp_amd_cyl_basis": (c_int, [c_void_p, c_size_t, c_void_p, c_void_p, c_void_p]),
    "exp_amd_cyl_orthocheck": (c_int, [c_void_p, c_void_p]),
    "exp_amd_force_destroy": (None, [c_void_p]),
    "exp_amd_force_set_level": (c_int, [c_void_p, c_int]),
    "exp_amd_force_determine_coefficients": (c_int, [c_void_p, c_void_p]),
    "exp_amd_force_get_coefs": (c_int, [c_void_p, c_void_p, c_size_t]),
    "exp_amd_force_set_coefs": (c_int, [c_void_p, c_void_p, c_size_t]),
    "exp_amd_force_ncoef": (c_size_t, [c_void_p]),
    "exp_amd_force_get_level_coefs": (c_int, [c_void_p, c_int, c_int, c_void_p, c_size_t]),
    "exp_amd_force_adjust_multistep_level": (c_int, [c_void_p, c_void_p, c_double, c_double_p, c_int,
                                                     c_int, c_int, POINTER(c_longlong)]),
    "exp_amd_cyl_create": (c_int, [c_void_p, POINTER(CylConfig), c_void_p, POINTER(c_void_p)]),
    "exp_amd_cyl_get_cylmass": (c_int, [c_void_p, POINTER(c_double)]),
    "exp_amd_cyl_set_cylmass": (c_int, [c_void_p, c_double]),
    "exp_amd_force_used": (c_int, [c_void_p, POINTER(c_longlong)]),
    "exp_amd_force_get_acceleration": (c_int, [c_void_p, c_void_p, c_int]),
    "exp_amd_force_multistep_reset": (c_int, [c_void_p]),
    "exp_amd_force_compute_multistep_coefficients": (c_int, [c_void_p, c_int]),
    "exp_amd_step_kdk": (c_int, [c_void_p, c_void_p, c_double]),
    "exp_amd_step_kdk_n": (c_int, [c_void_p, c_void_p, c_double, c_int]),
    "exp_amd_sim_create": (c_int, [c_void_p, c_int, c_double, c_double_p, c_int, POINTER(c_void_p)]),
    "exp_amd_sim_destroy": (None, [c_void_p]),
    "exp_amd_sim_add_component": (c_int, [c_void_p, c_void_p, c_void_p, POINTER(c_int)]),
    "exp_amd_sim_add_interaction": (c_int, [c_void_p, c_int, c_int]),
    "exp_amd_sim_init": (c_int, [c_void_p]),
    "exp_amd_sim_step": (c_int, [c_void_p, c_int]),
    "exp_amd_sim_time": (c_double, [c_void_p]),
    "exp_amd_sim_last_switches": (c_longlong, [c_void_p]),
    "exp_amd_sim_step_switches": (c_longlong, [c_void_p]),
    "exp_amd_profile_enable": (c_int, [c_void_p, c_int]),
    "exp_amd_profile_get": (c_int, [c_void_p, c_int, POINTER(c_char_p), POINTER(c_double),
                                    POINTER(c_longlong)]),
    "exp_amd_profile_reset": (c_int, [c_void_p]),
}

_lib = None


def load() -> ctypes.CDLL:
    """dlopen libexp_amd.so and type every entry point.  Raises if it is not built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ExpAmdError(
            f"{LIB_PATH} is missing: build the HIP extension first (`make -j8` at the repo "
            "root or `python -c 'import __graft_entry__ as g; g.build()'`). "
            "exp_amd has no CPU fallback.")
    lib = ctypes.CDLL(LIB_PATH, mode=ctypes.RTLD_GLOBAL)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)       # AttributeError if the symbol is not exported
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def check(rc: int, ctx=None) -> None:
    if rc != 0:
        lib = load()
        msg = lib.exp_amd_last_error(ctx) if ctx else lib.exp_amd_last_global_error()
        raise ExpAmdError(f"exp_amd error {rc}: {msg.decode() if msg else '?'}")


def as_f64(a, n=None):
    """Contiguous float64 view/copy + its pointer (None stays None)."""
    if a is None:
        return None, None
    arr = np.ascontiguousarray(a, dtype=np.float64)
    if n is not None and arr.size != n:
        raise ValueError(f"array of length {arr.size}, expected {n}")
    return arr, arr.ctypes.data_as(c_void_p)
