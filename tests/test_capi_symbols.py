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


def host_declared(header):
    src = open(os.path.join(ROOT, "include", header)).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    src = re.sub(r"struct\s+\w+\s*\{.*?\};", "", src, flags=re.S)          # the ops tables hold function POINTERS, not exports
    return sorted(set(re.findall(r"^\s*FlErrorCode\s+(\w+)\s*\(", src, flags=re.M)))


def test_host_mirror_and_cgns_headers_are_exported_and_bound(capi):
    from fluca_amd import build, hostapi
    names = host_declared("fluca_host.h")
    assert len(names) >= 50
    for name in names:
        assert hasattr(hostapi.lib, name), f"{name} declared in fluca_host.h but not exported by libfluca_host.so"
    assert set(hostapi.PROTOTYPES) <= set(names)
    if not build.have_hdf5():
        pytest.skip("no HDF5 C library in this image: libfluca_cgns.so is not built")
    G = hostapi.load_cgns()
    cg = host_declared("fluca_cgns.h")
    assert len(cg) >= 15
    for name in cg:
        assert hasattr(G, name), f"{name} declared in fluca_cgns.h but not exported by libfluca_cgns.so"
    assert set(hostapi.CGNS_PROTOTYPES) == set(cg)


def test_version_and_defaults(capi):
    assert b"gfx950" in capi.lib.fl_version()
    # the ABI number of the header the library was built from = the one in the tree's header, and it is what fl_version prints
    import re
    hdr = open(os.path.join(ROOT, "include", "fluca_hip.h")).read()
    want = int(re.search(r"#define FL_ABI_VERSION (\d+)", hdr).group(1))
    assert capi.lib.fl_abi_version() == want and f"abi {want}".encode() in capi.lib.fl_version()
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


def _grid(capi, n, xf=None):
    import numpy as np
    g = capi.fl_grid()
    keep = []
    for d in range(3):
        a = np.linspace(0.0, 1.0, n[d] + 1) if xf is None else np.asarray(xf[d], dtype=np.float64)
        keep.append(a)
        g.n[d] = n[d]
        g.xf[d] = a.ctypes.data
        g.xc[d] = None
    return g, keep


def test_create_validates_before_touching_the_gpu(capi):
    """argument errors carry the PETSc error class the reference would raise; nothing is leaked or dereferenced"""
    import numpy as np
    h = C.c_void_p()
    bc = lambda *v: (C.c_int * 6)(*v)
    g, keep = _grid(capi, (4, 4, 4))
    V, O, PER, SYM = 1, 2, 3, 4
    # periodic on one side only (the mesh would be inconsistent)
    assert capi.lib.fl_poisson_create(C.byref(g), bc(PER, V, V, V, V, V), 1.0, None, 0, C.byref(h)) == -62
    # NS_BC_NONE is "Unsupported boundary condition type" (PETSC_ERR_SUP, cnlinearcart3d.c:2461)
    assert capi.lib.fl_poisson_create(C.byref(g), bc(0, V, V, V, V, V), 1.0, None, 0, C.byref(h)) == -56
    # kappa = dt/rho must be positive and finite
    assert capi.lib.fl_poisson_create(C.byref(g), bc(V, V, V, V, V, V), 0.0, None, 0, C.byref(h)) == -63
    assert capi.lib.fl_poisson_create(C.byref(g), bc(V, V, V, V, V, V), float("nan"), None, 0, C.byref(h)) == -63
    # non-monotone coordinates
    g2, keep2 = _grid(capi, (4, 4, 4), [np.array([0, .2, .1, .5, 1.]), np.linspace(0, 1, 5), np.linspace(0, 1, 5)])
    assert capi.lib.fl_poisson_create(C.byref(g2), bc(V, V, V, V, V, V), 1.0, None, 0, C.byref(h)) == -62
    # a decomposition that does not tile the grid
    d = capi.fl_decomp()
    for a in range(3):
        d.ranks[a], d.coord[a], d.lo[a], d.len[a] = 1, 0, 0, 4
    d.ranks[0], d.len[0] = 2, 3          # first of two ranks but [0,3) + nothing = not the whole axis when coord==last? coord 0 of 2: fine; lo+len != n only allowed if not last
    d.coord[0] = 1                        # last rank must end at n and not start at 0
    assert capi.lib.fl_poisson_create(C.byref(g), bc(V, V, V, V, V, V), 1.0, C.byref(d), 0, C.byref(h)) in (-62, -63)
    assert not h.value


def test_ibm_and_solver_entry_points_reject_null_handles(capi):
    st = capi.fl_ksp_stats()
    o = capi.fl_ksp_opts()
    capi.lib.fl_ksp_opts_default(C.byref(o))
    assert capi.lib.fl_poisson_solve(None, None, None, C.byref(o), C.byref(st)) == -85
    assert capi.lib.fl_poisson_rhs(None, None, None, None, None, None) == -85
    assert capi.lib.fl_poisson_project(None, None, None, None, None, None, None, None) == -85
    assert capi.lib.fl_poisson_tune_placement(None, 3, None) == -85
    assert capi.lib.fl_ibm_create(None, 0, 1, None, None, None, None) == -85
    assert capi.lib.fl_ibm_destroy(None) == 0
    assert capi.lib.fl_halo_plan(None, None, None) == -85
