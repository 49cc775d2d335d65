"""CPU-only: the C-ABI library loads, exports every symbol include/fluca_hip.h declares, and its host-only helpers work.
No compute call is made (there is no GPU here)."""
import ctypes as C
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_functions():
    src = open(os.path.join(ROOT, "include", "fluca_hip.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    names = re.findall(r"^\s*(?:const\s+char\s*\*\s*|int\s+|void\s+)(fl_\w+)\s*\(", src, flags=re.M)
    assert len(names) >= 20
    return sorted(set(names))


@pytest.fixture(scope="module")
def capi():
    from fluca_amd import build
    build.build()
    from fluca_amd import capi
    return capi


def test_every_declared_symbol_is_exported_and_bound(capi):
    for name in declared_functions():
        assert hasattr(capi.lib, name), f"{name} declared in fluca_hip.h but not exported by libflucahip.so"
        assert name in capi.PROTOTYPES, f"{name} has no ctypes prototype"
    assert set(capi.PROTOTYPES) == set(declared_functions())


def test_version_and_defaults(capi):
    assert b"gfx950" in capi.lib.fl_version()
    o = capi.fl_ksp_opts()
    capi.lib.fl_ksp_opts_default(C.byref(o))
    # PETSc KSP defaults + what the reference's docs recommend for kspS (-ns_abf_schur_ksp_type cg -pc_type jacobi)
    assert (o.type, o.pc, o.norm_type, o.remove_nullspace) == (capi.KSP_CG, capi.PC_JACOBI, capi.NORM_PRECONDITIONED, 1)
    assert (o.maxit, o.rtol, o.atol, o.dtol) == (10000, 1e-5, 1e-50, 1e5)


def test_default_decomposition_matches_dmstag_rule(capi):
    # DMStag: N/m cells per rank, the first N%m ranks get one more; ranks numbered x-fastest
    n = (C.c_int64 * 3)(10, 7, 5)
    ranks = (C.c_int * 3)(3, 2, 1)
    seen = set()
    for rank in range(6):
        d = capi.fl_decomp()
        assert capi.lib.fl_decomp_default(n, ranks, rank, C.byref(d)) == 0
        assert tuple(d.coord) == (rank % 3, rank // 3, 0)
        assert d.len[0] == (4, 3, 3)[d.coord[0]] and d.lo[0] == (0, 4, 7)[d.coord[0]]
        assert d.len[1] == (4, 3)[d.coord[1]] and d.lo[1] == (0, 4)[d.coord[1]]
        assert (d.lo[2], d.len[2]) == (0, 5)
        seen.add(tuple(d.lo) + tuple(d.len))
    assert len(seen) == 6
    d = capi.fl_decomp()
    assert capi.lib.fl_decomp_default(n, ranks, 6, C.byref(d)) == -63       # PETSC_ERR_ARG_OUTOFRANGE
    assert capi.lib.fl_decomp_default((C.c_int64 * 3)(2, 7, 5), ranks, 0, C.byref(d)) == -63


def test_neighbors(capi):
    n = (C.c_int64 * 3)(8, 8, 8)
    ranks = (C.c_int * 3)(2, 2, 2)
    d = capi.fl_decomp()
    capi.lib.fl_decomp_default(n, ranks, 5, C.byref(d))      # coord (1,0,1)
    nonper = (C.c_int * 3)(0, 0, 0)
    per = (C.c_int * 3)(1, 1, 1)
    nb = [capi.lib.fl_decomp_neighbor(C.byref(d), nonper, b) for b in range(6)]
    assert nb == [4, -1, -1, 7, 1, -1]
    nb = [capi.lib.fl_decomp_neighbor(C.byref(d), per, b) for b in range(6)]
    assert nb == [4, 4, 7, 7, 1, 1]


def test_null_arguments_are_rejected_not_dereferenced(capi):
    assert capi.lib.fl_poisson_create(None, None, 1.0, None, 0, None) == -85   # PETSC_ERR_ARG_NULL
    assert capi.lib.fl_poisson_apply(None, None, None) == -85
    assert capi.lib.fl_poisson_destroy(None) == 0
