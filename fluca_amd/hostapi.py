"""ctypes binding of include/fluca_host.h (libfluca_host.so, the C host mirror of Fluca's Mesh / NS surface)."""
import ctypes as C
import os

from . import capi  # loads libflucahip.so first (libfluca_host.so links against it)

LIB_PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), "lib", "libfluca_host.so")
if not os.path.exists(LIB_PATH):
    raise ImportError(f"{LIB_PATH} is missing: run `python -m fluca_amd.build`")
lib = C.CDLL(LIB_PATH)

MESHCART_BOUNDARY_NONE, MESHCART_BOUNDARY_PERIODIC = 0, 1
MESHCART_LEFT, MESHCART_RIGHT, MESHCART_DOWN, MESHCART_UP, MESHCART_BACK, MESHCART_FRONT = range(6)
NS_BC_NONE, NS_BC_VELOCITY, NS_BC_PRESSURE_OUTLET, NS_BC_PERIODIC, NS_BC_SYMMETRY = range(5)
FL_DECIDE = -1
# positive PETSC_ERR_* values
ERR_SUP, ERR_ARG_WRONG, ERR_ARG_OUTOFRANGE, ERR_ARG_WRONGSTATE, ERR_ARG_NULL, ERR_ARG_UNKNOWN_TYPE = 56, 62, 63, 73, 85, 86

BCFunc = C.CFUNCTYPE(C.c_int, C.c_int, C.c_double, C.POINTER(C.c_double), C.POINTER(C.c_double), C.c_void_p)


class NSBoundaryCondition(C.Structure):
    _fields_ = [("type", C.c_int), ("velocity", BCFunc), ("ctx_velocity", C.c_void_p), ("pressure", BCFunc), ("ctx_pressure", C.c_void_p)]


_P = C.c_void_p
_i64p = C.POINTER(C.c_int64)
_ip = C.POINTER(C.c_int)
_argv = C.POINTER(C.c_char_p)
_dp3 = C.POINTER(C.c_void_p)
PROTOTYPES = {
    "MeshCreate": [C.POINTER(_P)], "MeshSetType": [_P, C.c_char_p], "MeshSetRank": [_P, C.c_int, C.c_int],
    "MeshCartCreate3d": [C.c_int, C.c_int, C.c_int, C.c_int64, C.c_int64, C.c_int64, C.c_int, C.c_int, C.c_int, _i64p, _i64p, _i64p, C.POINTER(_P)],
    "MeshSetFromOptions": [_P, C.c_int, _argv], "MeshSetUp": [_P],
    "MeshCartSetUniformCoordinates": [_P] + [C.c_double] * 6, "MeshCartSetCoordinates": [_P, _P, _P, _P],
    "MeshCartGetGlobalSizes": [_P, _i64p, _i64p, _i64p], "MeshCartGetNumRanks": [_P, _ip, _ip, _ip],
    "MeshCartSetRefinementFactor": [_P, C.c_int64, C.c_int64, C.c_int64], "MeshCartGetRefinementFactor": [_P, _i64p, _i64p, _i64p],
    "MeshCartGetCorners": [_P] + [_i64p] * 6, "MeshCartGetIsFirstRank": [_P, _ip, _ip, _ip], "MeshCartGetIsLastRank": [_P, _ip, _ip, _ip],
    "MeshCartGetBoundaryIndex": [_P, C.c_int, _ip], "MeshGetNumberBoundaries": [_P, _ip], "MeshDestroy": [C.POINTER(_P)],
    "NSCreate": [C.POINTER(_P)], "NSSetType": [_P, C.c_char_p], "NSGetType": [_P, C.POINTER(C.c_char_p)], "NSSetMesh": [_P, _P],
    "NSSetDevice": [_P, C.c_int], "NSSetDensity": [_P, C.c_double], "NSSetViscosity": [_P, C.c_double], "NSSetTimeStepSize": [_P, C.c_double],
    "NSSetMaxSteps": [_P, C.c_int64], "NSSetBoundaryCondition": [_P, C.c_int, NSBoundaryCondition],
    "NSGetBoundaryCondition": [_P, C.c_int, C.POINTER(NSBoundaryCondition)], "NSSetFromOptions": [_P, C.c_int, _argv], "NSSetUp": [_P],
    "NSStep": [_P], "NSGetTimeStep": [_P, _i64p], "NSGetTime": [_P, C.POINTER(C.c_double)], "NSDestroy": [C.POINTER(_P)],
    "NSGetPoisson": [_P, C.POINTER(_P)], "NSGetSchurKSPOptions": [_P, C.POINTER(C.POINTER(capi.fl_ksp_opts))], "NSGetNeedsNullSpace": [_P, _ip],
    "NSGetLocalSizes": [_P, _i64p], "NSPressureCorrection": [_P, _dp3, _dp3, _P, _P, C.POINTER(capi.fl_ksp_stats)],
    "NSSolve": [_P], "NSSetImmersedBoundary": [_P, C.c_int, C.c_int64, _P, _P, _P, _P, _P], "NSGetSolutionArrays": [_P, C.POINTER(_P), _dp3, C.POINTER(_P)],
    "NSGetLinearSolveInfo": [_P, _ip, C.POINTER(C.c_double), _ip], "NSGetLinearSolveResidualNorms": [_P, C.POINTER(C.c_double), C.POINTER(C.c_double)], "NSGetInnerIterations": [_P, _ip, _ip], "NSGetImmersedBoundary": [_P, C.POINTER(_P)],
    "NSSetPreviousState": [_P, _dp3, _dp3], "NSGetMomentum": [_P, C.POINTER(_P)],
    "NSGetMomentumKSPOptions": [_P, C.POINTER(C.POINTER(capi.fl_ksp_opts))],
    "NSApplyPreconditioner": [_P, _P, _dp3, _P, _P, _dp3, _P, C.POINTER(capi.fl_ksp_stats)],
    "NSUpdatePressure": [_P, _P, _P, _P, _P], "NSComputeStaggeredPressureGradientBC": [_P, C.c_double, _dp3],
    "MeshCartGetCoordinateArraysRead": [_P, C.POINTER(_P), C.POINTER(_P), C.POINTER(_P)], "MeshGetRank": [_P, _ip, _ip],
    "NSGetPressureHalfStep": [_P, C.POINTER(_P)], "NSGetMesh": [_P, C.POINTER(_P)], "NSGetDevice": [_P, _ip],
    "NSSetTimeStepAndTime": [_P, C.c_int64, C.c_double], "NSBarrier": [_P],
    "NSSetMaxTime": [_P, C.c_double], "NSGetMaxTime": [_P, C.POINTER(C.c_double)], "NSGetMaxSteps": [_P, _i64p],
    "NSGetDensity": [_P, C.POINTER(C.c_double)], "NSGetViscosity": [_P, C.POINTER(C.c_double)], "NSGetTimeStepSize": [_P, C.POINTER(C.c_double)],
    "NSSetTime": [_P, C.c_double], "NSSetTimeStep": [_P, C.c_int64], "NSSetErrorIfStepFailed": [_P, C.c_int], "NSGetErrorIfStepFailed": [_P, _ip],
    "NSGetConvergedReason": [_P, _ip],
    "NSMonitorSet": [_P, _P, _P, _P], "NSMonitorCancel": [_P], "NSMonitor": [_P],
    "FlucaTraceEnabled": [],
    # the ops-table entry points (nsimpl.h:21-31, meshimpl.h:16-25) and the viewer they take
    "NSRegister": [C.c_char_p, _P], "MeshRegister": [C.c_char_p, _P],
    "NSFormJacobian": [_P, _P, _P, C.c_int], "NSFormFunction": [_P, _P, _P], "NSGetJacobian": [_P, C.POINTER(_P)], "NSGetSolverVectors": [_P, _P, _P],
    "NSView": [_P, _P], "NSViewSolution": [_P, _P], "NSLoadSolution": [_P, _P],
    "MeshView": [_P, _P], "MeshLoad": [_P, _P], "MeshCreateGlobalVector": [_P, C.c_int, C.c_int, C.POINTER(_P), _i64p], "MeshCreateMatrix": [_P, C.c_int, C.c_int, C.POINTER(_P)],
    "FlucaViewerASCIIOpen": [C.c_char_p, C.POINTER(_P)], "FlucaViewerGetType": [_P, C.POINTER(C.c_char_p)], "FlucaViewerDestroy": [C.POINTER(_P)],
}
MESH_DM_SCALAR, MESH_DM_VECTOR, MESH_DM_STAG_SCALAR, MESH_DM_STAG_VECTOR = range(4)
NS_INIT_JACOBIAN, NS_UPDATE_JACOBIAN = 0, 1


class NSVec(C.Structure):
    """The composite vector (v, V[3], p) of device arrays: include/fluca_host.h."""
    _fields_ = [("v", C.c_void_p), ("V", C.c_void_p * 3), ("p", C.c_void_p)]


# struct _NSOps / struct _MeshOps: the slot names in the reference's order (nsimpl.h:21-31, meshimpl.h:16-25); the table is the first
# member of the object, so a handle can be read as an array of that many function pointers
NS_OPS = ("setfromoptions", "setup", "step", "formjacobian", "formfunction", "destroy", "view", "viewsolution", "loadsolution")
MESH_OPS = ("setfromoptions", "setup", "destroy", "view", "load", "createglobalvector", "creatematrix", "getnumberboundaries")


def ops_table(handle, names):
    """{slot name: function address or None} of a Mesh / NS handle."""
    tab = C.cast(handle, C.POINTER(C.c_void_p * len(names))).contents
    return {n: tab[i] for i, n in enumerate(names)}
MonitorFunc = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p)
for _n, _a in PROTOTYPES.items():
    _f = getattr(lib, _n)
    _f.restype = C.c_int
    _f.argtypes = _a


class FlucaCGNSLayout(C.Structure):
    _fields_ = [("N", C.c_int64 * 3), ("periodic", C.c_int * 3), ("rank", C.c_int), ("size", C.c_int), ("first", C.c_int * 3), ("last", C.c_int * 3),
                ("lo", C.c_int64 * 3), ("len", C.c_int64 * 3)]


class FlucaCGNSMonitor(C.Structure):
    _fields_ = [("viewer", C.c_void_p), ("view_interval", C.c_int)]


CGNS_LIB_PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), "lib", "libfluca_cgns.so")
_Lp = C.POINTER(FlucaCGNSLayout)
_names = C.POINTER(C.c_char_p)
CGNS_PROTOTYPES = {
    "FlucaViewerCGNSOpen": [C.c_char_p, C.c_char, C.POINTER(_P)], "FlucaViewerCGNSSetBatchSize": [_P, C.c_int], "FlucaViewerCGNSGetBatchSize": [_P, _ip],
    "FlucaViewerCGNSGetFileName": [_P, C.POINTER(C.c_char_p)], "FlucaViewerCGNSDestroy": [C.POINTER(_P)],
    "NSMonitorSolutionCGNS": [_P, _P],
    "FlucaCGNSCreateFile": [C.c_char_p, _Lp, _P, _P, _P], "FlucaCGNSWriteCellInfo": [C.c_char_p, _Lp],
    "FlucaCGNSCreateSolution": [C.c_char_p, _Lp, C.c_int64, C.c_int, _names, C.c_int, _names],
    "FlucaCGNSWriteCellField": [C.c_char_p, _Lp, C.c_int64, C.c_char_p, _P], "FlucaCGNSWriteFaceField": [C.c_char_p, _Lp, C.c_int64, C.c_char_p, _dp3],
    "FlucaCGNSWriteIterativeData": [C.c_char_p, C.c_int, _i64p, C.POINTER(C.c_double)],
    "FlucaCGNSReadInfo": [C.c_char_p, _i64p, _i64p, C.POINTER(C.c_double), _ip], "FlucaCGNSReadCoordinates": [C.c_char_p, _P, _P, _P],
    "FlucaCGNSReadCellField": [C.c_char_p, _Lp, C.c_int64, C.c_char_p, _P], "FlucaCGNSReadFaceField": [C.c_char_p, _Lp, C.c_int64, C.c_char_p, _dp3],
}
_cgns = None


def load_cgns():
    """libfluca_cgns.so (include/fluca_cgns.h): built only where an HDF5 C library exists (fluca_amd.build.build_cgns)."""
    global _cgns
    if _cgns is None:
        if not os.path.exists(CGNS_LIB_PATH):
            raise ImportError(f"{CGNS_LIB_PATH} is missing: it needs an HDF5 C library at build time (HDF5_ROOT, default /opt/conda); run `python -m fluca_amd.build`")
        _cgns = C.CDLL(CGNS_LIB_PATH)
        for n, a in list(CGNS_PROTOTYPES.items()) + [("NSViewSolution", [_P, _P]), ("NSLoadSolution", [_P, _P])]:   # the last two live in libfluca_host.so
            f = getattr(_cgns, n)
            f.restype = C.c_int
            f.argtypes = a
    return _cgns


def argv(*opts):
    a = [b"prog"] + [str(o).encode() for o in opts]
    return len(a), (C.c_char_p * len(a))(*a)
