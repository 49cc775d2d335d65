"""The plugin boundary of the C host mirror: struct _NSOps and struct _MeshOps carry the reference's
slots in the reference's order (fluca/include/fluca/private/nsimpl.h:21-31, meshimpl.h:16-25), NSRegister accepts a type written
against that table (tests/plugins/ns_probe.c: derives from NSCNLINEAR, replaces formfunction / formjacobian), and NSStep, NSSetUp,
NSView, NSViewSolution, NSLoadSolution, MeshView, MeshLoad go through the table."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

P = C.c_void_p
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def H():
    from fluca_amd import build
    build.build()
    from fluca_amd import hostapi
    return hostapi


@pytest.fixture(scope="module")
def probe(H):
    """tests/plugins/ns_probe.c built with gcc against include/fluca_host_impl.h, as a type implementation of the reference is built
    against nsimpl.h; registered once per process."""
    out = os.path.join(ROOT, "tests", "plugins", "libns_probe.so")
    src = os.path.join(ROOT, "tests", "plugins", "ns_probe.c")
    libdir = os.path.join(ROOT, "fluca_amd", "lib")
    if not os.path.exists(out) or os.path.getmtime(out) < max(os.path.getmtime(src), os.path.getmtime(os.path.join(libdir, "libfluca_host.so"))):
        subprocess.check_call(["gcc", "-std=gnu99", "-O1", "-Wall", "-Werror", "-fPIC", "-shared", "-o", out, src, "-L" + libdir, "-lfluca_host", "-lflucahip",
                               "-Wl,-rpath," + libdir])
    lib = C.CDLL(out)
    lib.ProbeSetScale.argtypes = [C.c_double]
    assert lib.ProbeRegister() == 0
    return lib


def _mesh(H, n=(12, 10, 8), periodic=(0, 0, 0)):
    mesh = P()
    assert H.lib.MeshCartCreate3d(periodic[0], periodic[1], periodic[2], n[0], n[1], n[2], -1, -1, -1, None, None, None, C.byref(mesh)) == 0
    assert H.lib.MeshSetUp(mesh) == 0
    assert H.lib.MeshCartSetUniformCoordinates(mesh, 0., 1., 0., 1., 0., 0.5) == 0
    return mesh


def test_ops_tables_have_the_reference_slots(H, probe):
    assert probe.ProbeSlotCount() == len(H.NS_OPS) == 9            # nsimpl.h:21-31
    assert probe.ProbeMeshSlotCount() == len(H.MESH_OPS) == 8      # meshimpl.h:16-25
    ns, ns2 = P(), P()
    assert H.lib.NSCreate(C.byref(ns)) == 0 and H.lib.NSSetType(ns, b"cnlinear") == 0
    base = H.ops_table(ns, H.NS_OPS)
    # cnlinear.c:177-185 fills every slot (setfromoptions reads no option there and stays empty here)
    assert all(base[k] for k in H.NS_OPS if k != "setfromoptions"), base
    assert H.lib.NSCreate(C.byref(ns2)) == 0 and H.lib.NSSetType(ns2, b"cnprobe") == 0
    t = C.c_char_p()
    assert H.lib.NSGetType(ns2, C.byref(t)) == 0 and t.value == b"cnprobe"
    derived = H.ops_table(ns2, H.NS_OPS)
    changed = {k for k in H.NS_OPS if derived[k] != base[k]}
    assert changed == {"formfunction", "formjacobian"}, changed
    assert H.lib.NSSetType(ns2, b"nonsense") == 86                 # PETSC_ERR_ARG_UNKNOWN_TYPE
    mesh = _mesh(H)
    mops = H.ops_table(mesh, H.MESH_OPS)
    assert all(mops[k] for k in H.MESH_OPS if k != "creatematrix"), mops
    m = P()
    assert H.lib.MeshCreateMatrix(mesh, H.MESH_DM_VECTOR, H.MESH_DM_VECTOR, C.byref(m)) == 56     # matrix-free: PETSC_ERR_SUP
    for h in (ns, ns2):
        assert H.lib.NSDestroy(C.byref(h)) == 0
    assert H.lib.MeshDestroy(C.byref(mesh)) == 0


def test_ascii_views(H, tmp_path):
    """NSView / MeshView (nsbasic.c:353-374, meshbasic.c:93-104, cart.c:182-201) through the view slots."""
    mesh = _mesh(H)
    ns = P()
    assert H.lib.NSCreate(C.byref(ns)) == 0 and H.lib.NSSetType(ns, b"cnlinear") == 0 and H.lib.NSSetMesh(ns, mesh) == 0
    assert H.lib.NSSetDensity(ns, 2.0) == 0 and H.lib.NSSetViscosity(ns, 0.25) == 0 and H.lib.NSSetTimeStepSize(ns, 1e-3) == 0
    path = tmp_path / "view.txt"
    v = P()
    assert H.lib.FlucaViewerASCIIOpen(str(path).encode(), C.byref(v)) == 0
    t = C.c_char_p()
    assert H.lib.FlucaViewerGetType(v, C.byref(t)) == 0 and t.value == b"ascii"
    assert H.lib.MeshView(mesh, v) == 0 and H.lib.NSView(ns, v) == 0
    assert H.lib.NSViewSolution(ns, v) == 73          # before NSSetUp
    assert H.lib.MeshLoad(mesh, v) == 73              # a write viewer is not readable
    assert H.lib.FlucaViewerDestroy(C.byref(v)) == 0 and not v.value
    text = path.read_text().splitlines()
    assert text[0].startswith("Mesh Object:") and text[1].strip() == "type: cart"
    assert text[2] == "Processor [0] M 12 N 10 P 8 m 1 n 1 p 1"
    assert text[3] == "X range of indices: 0 12, Y range of indices: 0 10, Z range of indices: 0 8"
    assert text[4].startswith("NS Object:") and text[5].strip() == "type: cnlinear"
    assert text[6] == "Density: 2, Viscosity: 0.25, Time step size: 0.001"
    assert text[7] == "Current time step: 0, Current time: 0"
    assert H.lib.NSDestroy(C.byref(ns)) == 0 and H.lib.MeshDestroy(C.byref(mesh)) == 0


def test_meshload_takes_sizes_and_coordinates_from_a_cgns_file(H, tmp_path):
    """MeshLoad -> MeshLoad_Cart -> MeshLoad_Cart_CGNS (meshbasic.c:114-127, cart.c:208-216, cartcgns.c:120-158): sizes and face
    coordinates come from the file, boundary types become NONE, MeshSetUp installs the coordinates (cart.c:131-140)."""
    if not os.path.exists(H.CGNS_LIB_PATH):
        pytest.skip("libfluca_cgns.so not built (no HDF5)")
    G = H.load_cgns()
    n = (6, 5, 4)
    xf = [np.cumsum(np.r_[0.0, np.random.default_rng(3 + d).uniform(0.5, 1.5, n[d])]) for d in range(3)]
    lay = H.FlucaCGNSLayout()
    lay.N[:], lay.lo[:], lay.len[:] = n, (0, 0, 0), n
    lay.periodic[:], lay.first[:], lay.last[:] = (0, 0, 0), (1, 1, 1), (1, 1, 1)
    lay.rank, lay.size = 0, 1
    f = str(tmp_path / "mesh.cgns").encode()
    assert G.FlucaCGNSCreateFile(f, C.byref(lay), *(a.ctypes.data_as(P) for a in xf)) == 0
    mesh, asc, v = P(), P(), P()
    assert H.lib.MeshCreate(C.byref(mesh)) == 0
    assert H.lib.FlucaViewerASCIIOpen(None, C.byref(asc)) == 0
    assert G.FlucaViewerCGNSOpen(f, b"r", C.byref(v)) == 0
    t = C.c_char_p()
    assert H.lib.FlucaViewerGetType(v, C.byref(t)) == 0 and t.value == b"flucacgns"
    assert H.lib.MeshLoad(mesh, v) == 0               # sets the type to cart when none is set
    assert H.lib.MeshSetUp(mesh) == 0
    M, N, Q = C.c_int64(), C.c_int64(), C.c_int64()
    assert H.lib.MeshCartGetGlobalSizes(mesh, C.byref(M), C.byref(N), C.byref(Q)) == 0 and (M.value, N.value, Q.value) == n
    ptr = [P(), P(), P()]
    assert H.lib.MeshCartGetCoordinateArraysRead(mesh, *(C.byref(q) for q in ptr)) == 0
    for d in range(3):
        got = np.ctypeslib.as_array(C.cast(ptr[d], C.POINTER(C.c_double)), shape=(n[d] + 1,))
        assert np.array_equal(got, xf[d])
    assert H.lib.MeshLoad(mesh, v) == 73              # after MeshSetUp the sizes are fixed
    mesh2 = P()
    assert H.lib.MeshCreate(C.byref(mesh2)) == 0
    H.lib.FlucaViewerDestroy(C.byref(asc))
    assert H.lib.FlucaViewerASCIIOpen(str(tmp_path / "x.txt").encode(), C.byref(asc)) == 0
    assert H.lib.MeshLoad(mesh2, asc) == 73           # PetscViewerCheckReadable comes first: an ASCII viewer is a write viewer
    for h in (mesh, mesh2):
        assert H.lib.MeshDestroy(C.byref(h)) == 0
    assert H.lib.FlucaViewerDestroy(C.byref(v)) == 0 and H.lib.FlucaViewerDestroy(C.byref(asc)) == 0


def _lid_ns(H, mesh, nstype, opts=()):
    ns = P()
    assert H.lib.NSCreate(C.byref(ns)) == 0 and H.lib.NSSetType(ns, nstype) == 0 and H.lib.NSSetMesh(ns, mesh) == 0

    @H.BCFunc
    def lid(dim, t, x, val, ctx):
        val[0], val[1], val[2] = 1.0 + 0.5 * np.sin(3.0 * x[0]) * np.cos(2.0 * t), 0.0, 0.0
        return 0

    @H.BCFunc
    def wall(dim, t, x, val, ctx):
        val[0] = val[1] = val[2] = 0.0
        return 0

    for b in range(6):
        assert H.lib.NSSetBoundaryCondition(ns, b, H.NSBoundaryCondition(type=H.NS_BC_VELOCITY, velocity=lid if b == 3 else wall)) == 0
    argc, av = H.argv("-ns_time_step_size", 2e-3, "-ns_viscosity", 0.05, "-ns_ksp_type", "preonly", "-ns_abf_schur_ksp_rtol", 1e-10, "-ns_abf_momentum_ksp_rtol", 1e-10, *opts)
    assert H.lib.NSSetFromOptions(ns, argc, av) == 0 and H.lib.NSSetUp(ns) == 0
    return ns, (lid, wall)


def _velocity(H, ns):
    sz = (C.c_int64 * 4)()
    assert H.lib.NSGetLocalSizes(ns, sz) == 0
    v = P()
    assert H.lib.NSGetSolutionArrays(ns, C.byref(v), None, None) == 0
    out = np.empty(3 * sz[0])
    H.capi.check(H.capi.lib.fl_memcpy_d2h(0, out.ctypes.data_as(P), v, out.size * 8))
    return out


@pytest.mark.gpu
def test_registered_type_overriding_formfunction_is_what_nsstep_calls(H, probe):
    """A type registered through NSRegister that replaces formfunction sees it called once per step by NSStep (SNESPicard's b,
    nsbasic.c:115-131, 248), formjacobian once at NSSetUp with NS_INIT_JACOBIAN (nsbasic.c:207) and once per step with
    NS_UPDATE_JACOBIAN; with the override doing nothing the step is bit for bit NSCNLINEAR's, with the override scaling
    momrhs it is not."""
    mesh = _mesh(H)
    ref, keep0 = _lid_ns(H, mesh, b"cnlinear")
    for _ in range(2):
        assert H.lib.NSStep(ref) == 0
    vref = _velocity(H, ref)
    assert np.abs(vref).max() > 1e-4
    f0, i0, u0 = probe.ProbeFunctionCalls(), probe.ProbeJacobianCalls(0), probe.ProbeJacobianCalls(1)
    probe.ProbeSetScale(1.0)
    ns, keep1 = _lid_ns(H, mesh, b"cnprobe")
    assert probe.ProbeJacobianCalls(0) == i0 + 1 and probe.ProbeJacobianCalls(1) == u0        # NSSetUp: NS_INIT_JACOBIAN once
    x, r = H.NSVec(), H.NSVec()
    assert H.lib.NSGetSolverVectors(ns, C.byref(x), C.byref(r)) == 0 and r.v and r.p and all(r.V)
    assert H.lib.NSFormFunction(ns, C.byref(x), C.byref(r)) == 73                           # no sol0 before the first step
    f0 = probe.ProbeFunctionCalls()
    for _ in range(2):
        assert H.lib.NSStep(ns) == 0
    assert probe.ProbeFunctionCalls() == f0 + 2 and probe.ProbeJacobianCalls(1) == u0 + 2
    assert np.array_equal(_velocity(H, ns), vref)
    # the public entry points reach the same slots
    J = P()
    assert H.lib.NSGetJacobian(ns, C.byref(J)) == 0 and J.value
    assert H.lib.NSFormFunction(ns, C.byref(x), C.byref(r)) == 0 and probe.ProbeFunctionCalls() == f0 + 3
    assert H.lib.NSFormJacobian(ns, C.byref(x), J, H.NS_UPDATE_JACOBIAN) == 0 and probe.ProbeJacobianCalls(1) == u0 + 3
    assert H.lib.NSDestroy(C.byref(ns)) == 0
    probe.ProbeSetScale(2.0)
    ns, keep2 = _lid_ns(H, mesh, b"cnprobe")
    for _ in range(2):
        assert H.lib.NSStep(ns) == 0
    v2 = _velocity(H, ns)
    probe.ProbeSetScale(1.0)
    assert np.abs(v2 - vref).max() > 0.1 * np.abs(vref).max()                                 # the override's right-hand side was used
    assert H.lib.NSDestroy(C.byref(ns)) == 0 and H.lib.NSDestroy(C.byref(ref)) == 0 and H.lib.MeshDestroy(C.byref(mesh)) == 0


@pytest.mark.gpu
def test_mesh_create_global_vector_sizes(H):
    """MeshCreateGlobalVector through the createglobalvector slot (cart.c:218-229): the block of each of the four DMs."""
    mesh = _mesh(H, n=(6, 5, 4), periodic=(0, 1, 0))
    cells, faces = 6 * 5 * 4, 7 * 5 * 4 + 6 * 5 * 4 + 6 * 5 * 5        # the periodic axis owns no extra face
    for dm, want in ((H.MESH_DM_SCALAR, cells), (H.MESH_DM_VECTOR, 3 * cells), (H.MESH_DM_STAG_SCALAR, faces), (H.MESH_DM_STAG_VECTOR, 3 * faces)):
        v, n = P(), C.c_int64()
        assert H.lib.MeshCreateGlobalVector(mesh, dm, 0, C.byref(v), C.byref(n)) == 0 and n.value == want
        out = np.ones(want)
        H.capi.check(H.capi.lib.fl_memcpy_d2h(0, out.ctypes.data_as(P), v, out.size * 8))
        assert not out.any()
        H.capi.check(H.capi.lib.fl_free(0, v))
    v = P()
    assert H.lib.MeshCreateGlobalVector(mesh, 7, 0, C.byref(v), None) == 63
    assert H.lib.MeshDestroy(C.byref(mesh)) == 0
