"""ctypes binding of include/fluca_hip.h (the C-ABI).  Loads fluca_amd/lib/libflucahip.so; raises if it is absent.

torch is imported first on purpose: libflucahip.so needs libamdhip64.so.7, and inside a torch process the copy torch
already loaded must be the one that resolves (one HIP runtime per process).
"""
import ctypes as C
import os

import torch  # noqa: F401  (loads the process-wide HIP runtime first)

_HERE = os.path.dirname(os.path.abspath(__file__))
# FLUCA_LIB_DIR: another build of the library than the product's (lib_kbench/, fluca_amd/build.py) -- measurement tools only
LIB_PATH = os.path.join(os.environ.get("FLUCA_LIB_DIR") or os.path.join(_HERE, "lib"), "libflucahip.so")

if not os.path.exists(LIB_PATH):
    raise ImportError(
        f"{LIB_PATH} is missing: build it with `python -m fluca_amd.build` (hipcc --offload-arch=gfx950). "
        "There is no CPU fallback for the product path.")

lib = C.CDLL(LIB_PATH)

BC_NONE, BC_VELOCITY, BC_PRESSURE_OUTLET, BC_PERIODIC, BC_SYMMETRY = range(5)
KSP_CG, KSP_BCGS, KSP_CHEBYSHEV, KSP_GMRES = range(4)
PC_NONE, PC_JACOBI = range(2)
NORM_PRECONDITIONED, NORM_UNPRECONDITIONED, NORM_NATURAL, NORM_NONE = range(4)
DELTA_PESKIN4, DELTA_ROMA3 = range(2)
UNIQUE_ID_BYTES = 128

ERRORS = {0: "FL_SUCCESS", -55: "FL_ERR_MEM", -56: "FL_ERR_SUP", -60: "FL_ERR_ARG_SIZ", -62: "FL_ERR_ARG_WRONG",
          -63: "FL_ERR_ARG_OUTOFRANGE", -73: "FL_ERR_ARG_WRONGSTATE", -76: "FL_ERR_LIB", -85: "FL_ERR_ARG_NULL",
          -91: "FL_ERR_NOT_CONVERGED", -97: "FL_ERR_GPU"}


class FlucaError(RuntimeError):
    def __init__(self, rc, what):
        super().__init__(f"{what} -> {ERRORS.get(rc, rc)} ({rc})")
        self.rc = rc


def check(rc, what="libflucahip"):
    if rc != 0:
        raise FlucaError(rc, what)


class fl_comm_info(C.Structure):
    _fields_ = [("transport", C.c_int), ("rank", C.c_int), ("nranks", C.c_int), ("loopback", C.c_int), ("neighbours", C.c_int), ("messages", C.c_int), ("halo_bytes", C.c_int64)]


class fl_grid(C.Structure):
    _fields_ = [("n", C.c_int64 * 3), ("xf", C.c_void_p * 3), ("xc", C.c_void_p * 3)]


class fl_decomp(C.Structure):
    _fields_ = [("ranks", C.c_int * 3), ("coord", C.c_int * 3), ("lo", C.c_int64 * 3), ("len", C.c_int64 * 3)]


class fl_ksp_opts(C.Structure):
    _fields_ = [("type", C.c_int), ("pc", C.c_int), ("norm_type", C.c_int), ("remove_nullspace", C.c_int),
                ("maxit", C.c_int), ("rtol", C.c_double), ("atol", C.c_double), ("dtol", C.c_double),
                ("emin", C.c_double), ("emax", C.c_double), ("variant", C.c_int), ("check_every", C.c_int),
                ("profile", C.c_int), ("history", C.POINTER(C.c_double)), ("nhistory", C.c_int),
                ("mg_levels", C.c_int), ("mg_smooth_its", C.c_int), ("gmres_restart", C.c_int), ("cg_single_reduction", C.c_int),
                ("initial_guess_nonzero", C.c_int)]


class fl_ksp_stats(C.Structure):
    _fields_ = [("iters", C.c_int), ("reason", C.c_int), ("rnorm0", C.c_double), ("rnorm", C.c_double),
                ("seconds", C.c_double), ("kernel_ms", C.c_double), ("kernel_launches", C.c_int),
                ("kernel2_ms", C.c_double), ("kernel2_launches", C.c_int)]


class fl_dmstag_local(C.Structure):
    _fields_ = [("gstart", C.c_int64 * 3), ("gsize", C.c_int64 * 3), ("start", C.c_int64 * 3), ("entries", C.c_int)]


class fl_halo_msg(C.Structure):
    _fields_ = [("peer", C.c_int), ("send_boundary", C.c_int), ("recv_boundary", C.c_int), ("sendtag", C.c_int), ("recvtag", C.c_int)]


EXCHANGE_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_int),
                          C.POINTER(C.c_void_p), C.POINTER(C.c_void_p), C.POINTER(C.c_int64))
ALLREDUCE_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.POINTER(C.c_double), C.c_int)

# every symbol include/fluca_hip.h declares (tests/test_capi_symbols.py checks the header against this table)
_P = C.c_void_p
PROTOTYPES = {
    "fl_poisson_create": (C.c_int, [C.POINTER(fl_grid), C.POINTER(C.c_int), C.c_double, C.POINTER(fl_decomp), C.c_int, C.POINTER(_P)]),
    "fl_poisson_destroy": (C.c_int, [_P]),
    "fl_poisson_set_stream": (C.c_int, [_P, _P]),
    "fl_poisson_synchronize": (C.c_int, [_P]),
    "fl_poisson_barrier": (C.c_int, [_P]),
    "fl_poisson_sizes": (C.c_int, [_P, C.POINTER(C.c_int64)]),
    "fl_poisson_allreduce_sum": (C.c_int, [_P, C.POINTER(C.c_double), C.c_int]),
    "fl_ksp_opts_default": (None, [C.POINTER(fl_ksp_opts)]),
    "fl_version": (C.c_char_p, []),
    "fl_abi_version": (C.c_int, []),
    "fl_poisson_comm_info": (C.c_int, [_P, C.POINTER(fl_comm_info)]),
    "fl_current_device": (C.c_int, [C.POINTER(C.c_int)]),
    "fl_malloc": (C.c_int, [C.c_int, C.c_size_t, C.POINTER(_P)]),
    "fl_free": (C.c_int, [C.c_int, _P]),
    "fl_memcpy_h2d": (C.c_int, [C.c_int, _P, _P, C.c_size_t]),
    "fl_memcpy_d2h": (C.c_int, [C.c_int, _P, _P, C.c_size_t]),
    "fl_malloc_host": (C.c_int, [C.c_size_t, C.POINTER(C.c_void_p)]),
    "fl_free_host": (C.c_int, [_P]),
    "fl_poisson_upload": (C.c_int, [_P, _P, _P, C.c_size_t]),
    "fl_poisson_upload_fence": (C.c_int, [_P]),
    "fl_tuning_set": (C.c_int, [C.c_char_p, C.c_int]),
    "fl_tuning_get": (C.c_int, [C.c_char_p, C.POINTER(C.c_int)]),
    "fl_poisson_tune_placement": (C.c_int, [_P, C.c_int, C.POINTER(C.c_double)]),
    "fl_poisson_vector_bytes": (C.c_int, [_P, C.POINTER(C.c_int64)]),
    "fl_poisson_apply": (C.c_int, [_P, _P, _P]),
    "fl_poisson_diagonal": (C.c_int, [_P, _P]),
    "fl_poisson_solve": (C.c_int, [_P, _P, _P, C.POINTER(fl_ksp_opts), C.POINTER(fl_ksp_stats)]),
    "fl_poisson_rhs": (C.c_int, [_P, _P, _P, _P, _P, _P]),
    "fl_poisson_project": (C.c_int, [_P, _P, _P, _P, _P, _P, _P, _P]),
    "fl_poisson_gst_bc": (C.c_int, [_P, C.c_int, _P, _P]),
    "fl_pressure_update": (C.c_int, [_P, C.c_int, _P, _P, _P, _P]),
    "fl_layout_from_dmstag_local": (C.c_int, [_P, C.POINTER(fl_dmstag_local), C.c_int, C.c_int, _P, _P]),
    "fl_layout_to_dmstag_local": (C.c_int, [_P, C.POINTER(fl_dmstag_local), C.c_int, C.c_int, _P, _P]),
    "fl_layout_from_dmstag_global": (C.c_int, [_P, C.POINTER(C.c_int), C.c_int, C.c_int, _P, _P]),
    "fl_layout_to_dmstag_global": (C.c_int, [_P, C.POINTER(C.c_int), C.c_int, C.c_int, _P, _P]),
    "fl_dmstag_global_entries": (C.c_int, [_P, C.POINTER(C.c_int), C.POINTER(C.c_int64)]),
    "fl_comm_unique_id": (C.c_int, [_P]),
    "fl_poisson_comm_init_rccl": (C.c_int, [_P, _P, C.c_int, C.c_int]),
    "fl_poisson_comm_init_host": (C.c_int, [_P, EXCHANGE_FN, ALLREDUCE_FN, _P, C.c_int, C.c_int]),
    "fl_poisson_comm_oneshot_handle": (C.c_int, [_P, _P, C.POINTER(_P)]),
    "fl_poisson_comm_oneshot_attach": (C.c_int, [_P, _P, C.POINTER(_P)]),
    "fl_poisson_comm_oneshot_error": (C.c_int, [_P, C.POINTER(C.c_int)]),
    "fl_halo_plan": (C.c_int, [C.POINTER(fl_decomp), C.POINTER(C.c_int), C.POINTER(fl_halo_msg)]),
    "fl_decomp_default": (C.c_int, [C.POINTER(C.c_int64), C.POINTER(C.c_int), C.c_int, C.POINTER(fl_decomp)]),
    "fl_decomp_neighbor": (C.c_int, [C.POINTER(fl_decomp), C.POINTER(C.c_int), C.c_int]),
    "fl_momentum_create": (C.c_int, [_P, C.POINTER(_P)]),
    "fl_momentum_destroy": (C.c_int, [_P]),
    "fl_momentum_set_state": (C.c_int, [_P, C.c_double, C.c_double, C.c_double, _P, _P]),
    "fl_momentum_set_state_v0": (C.c_int, [_P, C.c_double, C.c_double, C.c_double, _P, _P, _P]),
    "fl_momentum_set_coefficients": (C.c_int, [_P, C.c_double, C.c_double, C.c_double]),
    "fl_momentum_apply": (C.c_int, [_P, _P, _P]),
    "fl_momentum_diagonal": (C.c_int, [_P, _P]),
    "fl_momentum_rowsum": (C.c_int, [_P, _P]),
    "fl_momentum_gershgorin": (C.c_int, [_P, C.POINTER(C.c_double)]),
    "fl_momentum_chebyshev_interval": (C.c_int, [_P, C.POINTER(C.c_double), C.POINTER(C.c_double)]),
    "fl_abf_set_ainv_types": (C.c_int, [_P, C.c_int, C.c_int]),
    "fl_abf_schur_apply": (C.c_int, [_P, _P, _P]),
    "fl_momentum_solve": (C.c_int, [_P, _P, _P, C.POINTER(fl_ksp_opts), C.POINTER(fl_ksp_stats)]),
    "fl_momentum_face_interp": (C.c_int, [_P, _P, _P, _P]),
    "fl_poisson_gershgorin": (C.c_int, [_P, C.c_int, C.POINTER(C.c_double)]),
    "fl_vec_lincomb": (C.c_int, [_P, C.c_int64, C.c_double, _P, C.c_double, _P, _P]),
    "fl_vec_dot": (C.c_int, [_P, C.c_int64, _P, _P, C.POINTER(C.c_double)]),
    "fl_vec_mdot": (C.c_int, [_P, C.c_int64, _P, C.POINTER(C.c_void_p), C.c_int, C.POINTER(C.c_double)]),
    "fl_vec_maxpy": (C.c_int, [_P, C.c_int64, _P, C.POINTER(C.c_double), C.POINTER(C.c_void_p), C.c_int]),
    "fl_boundary_set_faces": (C.c_int, [_P, C.c_int, C.c_double, _P, _P]),
    "fl_boundary_add_faces": (C.c_int, [_P, C.c_int, C.c_double, _P, _P]),
    "fl_momentum_face_interp_scaled": (C.c_int, [_P, C.c_double, _P, _P, _P]),
    "fl_boundary_add_cells": (C.c_int, [_P, C.c_int, C.c_double, _P, _P]),
    "fl_momentum_rhs": (C.c_int, [_P, C.c_double, C.c_double, C.c_double, _P, _P, _P, _P]),
    "fl_momentum_interp_faces": (C.c_int, [_P, _P, _P, _P]),
    "fl_momentum_interp_faces_ends": (C.c_int, [_P, _P, _P, _P]),
    "fl_abf_jacobian_mult": (C.c_int, [_P, _P, _P, _P, _P, _P, _P]),
    "fl_abf_apply": (C.c_int, [_P, C.POINTER(fl_ksp_opts), C.POINTER(fl_ksp_opts), _P, _P, _P, _P, _P, _P, C.POINTER(fl_ksp_stats)]),
    "fl_ibm_create": (C.c_int, [_P, C.c_int, C.c_int64, _P, _P, _P, C.POINTER(_P)]),
    "fl_ibm_update": (C.c_int, [_P, _P, _P, _P]),
    "fl_ibm_interp": (C.c_int, [_P, C.c_int, _P, _P]),
    "fl_ibm_spread": (C.c_int, [_P, C.c_int, _P, _P, _P]),
    "fl_ibm_destroy": (C.c_int, [_P]),
}
for _name, (_res, _args) in PROTOTYPES.items():
    _f = getattr(lib, _name)  # AttributeError if the library lacks a declared symbol
    _f.restype = _res
    _f.argtypes = _args
