"""-m gpu: NSViewSolution / NSLoadSolution / NSMonitorSolutionCGNS of the C host mirror (include/fluca_cgns.h; reference:
nssol.c:130-203, nsmon.c:91-100, cartcgns.c, flucacgns.c).  A run that is dumped, destroyed, loaded into a new NS and
continued must arrive where the uninterrupted run arrives."""
import ctypes as C

import numpy as np
import pytest

from fluca_amd import build as flbuild

pytestmark = [pytest.mark.gpu, pytest.mark.skipif(not flbuild.have_hdf5(), reason="no HDF5 C library in this image")]
P = C.c_void_p
N = (16, 12, 8)


def make_ns(H, max_steps, keep):
    """lid-driven cavity like cavity_flow_3d.c on a small stretched mesh"""
    mesh = P()
    assert H.lib.MeshCartCreate3d(0, 0, 0, N[0], N[1], N[2], 1, 1, 1, None, None, None, C.byref(mesh)) == 0
    assert H.lib.MeshSetUp(mesh) == 0
    xf = [np.linspace(0., 1., N[0] + 1) ** 1.2, np.linspace(0., 1., N[1] + 1), 0.5 * np.linspace(0., 1., N[2] + 1) ** 0.9]
    assert H.lib.MeshCartSetCoordinates(mesh, *[a.ctypes.data_as(C.c_void_p) for a in xf]) == 0
    ns = P()
    assert H.lib.NSCreate(C.byref(ns)) == 0 and H.lib.NSSetType(ns, b"cnlinear") == 0 and H.lib.NSSetMesh(ns, mesh) == 0
    assert H.lib.NSSetDensity(ns, 1.0) == 0 and H.lib.NSSetViscosity(ns, 0.01) == 0

    @H.BCFunc
    def lid(dim, t, x, val, ctx):
        val[0], val[1], val[2] = 1.0, 0.0, 0.0
        return 0

    @H.BCFunc
    def wall(dim, t, x, val, ctx):
        val[0] = val[1] = val[2] = 0.0
        return 0

    keep += [lid, wall, xf]
    for b in range(6):
        bc = H.NSBoundaryCondition(type=H.NS_BC_SYMMETRY) if b == 4 else H.NSBoundaryCondition(type=H.NS_BC_VELOCITY, velocity=lid if b == 3 else wall)
        assert H.lib.NSSetBoundaryCondition(ns, b, bc) == 0
    argc, av = H.argv("-ns_time_step_size", 0.01, "-ns_max_steps", max_steps, "-ns_ksp_rtol", 1e-10, "-ns_abf_schur_ksp_rtol", 1e-12,
                      "-ns_abf_momentum_ksp_rtol", 1e-12)
    assert H.lib.NSSetFromOptions(ns, argc, av) == 0 and H.lib.NSSetUp(ns) == 0
    return mesh, ns, xf


def fetch(H, ns):
    sz = (C.c_int64 * 4)()
    assert H.lib.NSGetLocalSizes(ns, sz) == 0
    v, p, ph, V = P(), P(), P(), (C.c_void_p * 3)()
    assert H.lib.NSGetSolutionArrays(ns, C.byref(v), V, C.byref(p)) == 0 and H.lib.NSGetPressureHalfStep(ns, C.byref(ph)) == 0
    out = {}
    for name, ptr, n in [("v", v, 3 * sz[0]), ("p", p, sz[0]), ("phalf", ph, sz[0])] + [(f"V{d}", C.c_void_p(V[d]), sz[1 + d]) for d in range(3)]:
        a = np.empty(int(n))
        H.capi.check(H.capi.lib.fl_memcpy_d2h(0, a.ctypes.data_as(C.c_void_p), ptr, a.size * 8))
        out[name] = a
    return out


def full_layout(H):
    lay = H.FlucaCGNSLayout()
    for d in range(3):
        lay.N[d], lay.len[d], lay.lo[d], lay.first[d], lay.last[d], lay.periodic[d] = N[d], N[d], 0, 1, 1, 0
    lay.rank, lay.size = 0, 1
    return lay


@pytest.fixture(scope="module")
def H():
    flbuild.build()
    from fluca_amd import hostapi
    hostapi.capi = __import__("fluca_amd.capi", fromlist=["capi"])
    return hostapi


def test_dump_load_continue_equals_the_uninterrupted_run(H, tmp_path):
    G = H.load_cgns()
    keep = []
    mesh, ns, xf = make_ns(H, 5, keep)
    assert H.lib.NSSolve(ns) == 0
    straight = fetch(H, ns)
    H.lib.NSDestroy(C.byref(ns)), H.lib.MeshDestroy(C.byref(mesh))

    mesh, ns, _ = make_ns(H, 3, keep)
    assert H.lib.NSSolve(ns) == 0
    at3 = fetch(H, ns)
    ck = str(tmp_path / "ckpt.cgns").encode()
    viewer = P()
    assert G.FlucaViewerCGNSOpen(ck, b"w", C.byref(viewer)) == 0
    assert G.NSLoadSolution(ns, viewer) == 73                     # a write viewer is not readable
    assert G.NSViewSolution(ns, viewer) == 0
    assert G.NSViewSolution(ns, viewer) == 0                      # same step again: nothing new is written
    assert G.FlucaViewerCGNSDestroy(C.byref(viewer)) == 0
    H.lib.NSDestroy(C.byref(ns)), H.lib.MeshDestroy(C.byref(mesh))

    # the file holds the numbers that were on the device, in the reference's arrays
    lay = full_layout(H)
    Nr, step, t, nst = (C.c_int64 * 3)(), C.c_int64(), C.c_double(), C.c_int()
    assert G.FlucaCGNSReadInfo(ck, Nr, C.byref(step), C.byref(t), C.byref(nst)) == 0
    assert tuple(Nr) == N and step.value == 3 and nst.value == 1 and abs(t.value - 0.03) < 1e-15
    ncell = N[0] * N[1] * N[2]
    for c, name in enumerate(("VelocityX", "VelocityY", "VelocityZ")):
        a = np.empty(ncell)
        assert G.FlucaCGNSReadCellField(ck, C.byref(lay), 3, name.encode(), a.ctypes.data) == 0
        assert np.array_equal(a, at3["v"][c * ncell:(c + 1) * ncell])
    a = np.empty(ncell)
    assert G.FlucaCGNSReadCellField(ck, C.byref(lay), 3, b"PressureHalfStep", a.ctypes.data) == 0 and np.array_equal(a, at3["phalf"])
    got = [np.empty(N[d] + 1) for d in range(3)]
    assert G.FlucaCGNSReadCoordinates(ck, *[g.ctypes.data for g in got]) == 0
    assert all(np.array_equal(got[d], xf[d]) for d in range(3))

    mesh, ns, _ = make_ns(H, 5, keep)
    viewer = P()
    assert G.FlucaViewerCGNSOpen(ck, b"r", C.byref(viewer)) == 0
    assert G.NSViewSolution(ns, viewer) == 73                     # and a read viewer is not writable
    assert G.NSLoadSolution(ns, viewer) == 0
    assert G.FlucaViewerCGNSDestroy(C.byref(viewer)) == 0
    step, t = C.c_int64(), C.c_double()
    H.lib.NSGetTimeStep(ns, C.byref(step)), H.lib.NSGetTime(ns, C.byref(t))
    assert step.value == 3 and abs(t.value - 0.03) < 1e-15        # nssol.c:199-201
    loaded = fetch(H, ns)
    for k in at3:
        assert np.array_equal(loaded[k], at3[k]), k
    assert H.lib.NSSolve(ns) == 0                                 # steps 3 and 4
    resumed = fetch(H, ns)
    H.lib.NSDestroy(C.byref(ns)), H.lib.MeshDestroy(C.byref(mesh))
    for k in straight:
        scale = max(np.abs(straight[k]).max(), 1e-30)
        assert np.abs(resumed[k] - straight[k]).max() <= 1e-9 * scale, (k, np.abs(resumed[k] - straight[k]).max() / scale)

    # a mesh of another size refuses the file ("Mesh size does not match CGNS zone size", cartcgns.c:697)
    mesh2 = P()
    assert H.lib.MeshCartCreate3d(0, 0, 0, 8, 8, 8, 1, 1, 1, None, None, None, C.byref(mesh2)) == 0 and H.lib.MeshSetUp(mesh2) == 0
    assert H.lib.MeshCartSetUniformCoordinates(mesh2, 0., 1., 0., 1., 0., 1.) == 0
    ns2 = P()
    assert H.lib.NSCreate(C.byref(ns2)) == 0 and H.lib.NSSetType(ns2, b"cnlinear") == 0 and H.lib.NSSetMesh(ns2, mesh2) == 0
    for b in range(6):
        assert H.lib.NSSetBoundaryCondition(ns2, b, H.NSBoundaryCondition(type=H.NS_BC_VELOCITY, velocity=keep[1])) == 0
    assert H.lib.NSSetDensity(ns2, 1.0) == 0 and H.lib.NSSetViscosity(ns2, 0.01) == 0 and H.lib.NSSetTimeStepSize(ns2, 0.01) == 0
    assert H.lib.NSSetUp(ns2) == 0
    viewer = P()
    assert G.FlucaViewerCGNSOpen(ck, b"r", C.byref(viewer)) == 0
    assert G.NSLoadSolution(ns2, viewer) == 76
    G.FlucaViewerCGNSDestroy(C.byref(viewer)), H.lib.NSDestroy(C.byref(ns2)), H.lib.MeshDestroy(C.byref(mesh2))


def test_monitor_writes_batches_of_steps(H, tmp_path):
    """-ns_monitor_solution cgns:out_%d.cgns with -viewer_cgns_batch_size 2: NSSolve monitors before every step and after
    the last (nsbasic.c:337-345), so 3 steps give the solutions 0..3 in the files 0 and 2."""
    G = H.load_cgns()
    keep = []
    mesh, ns, _ = make_ns(H, 3, keep)
    viewer = P()
    assert G.FlucaViewerCGNSOpen(str(tmp_path / "out_%d.cgns").encode(), b"w", C.byref(viewer)) == 0
    assert G.FlucaViewerCGNSSetBatchSize(viewer, 2) == 0
    mon = H.FlucaCGNSMonitor(viewer=viewer.value, view_interval=1)
    fn = C.cast(G.NSMonitorSolutionCGNS, C.c_void_p)
    assert H.lib.NSMonitorSet(ns, fn, C.cast(C.pointer(mon), C.c_void_p), None) == 0
    assert H.lib.NSSolve(ns) == 0
    final = fetch(H, ns)
    assert G.FlucaViewerCGNSDestroy(C.byref(viewer)) == 0
    H.lib.NSDestroy(C.byref(ns)), H.lib.MeshDestroy(C.byref(mesh))
    lay = full_layout(H)
    for first, last in ((0, 1), (2, 3)):
        f = str(tmp_path / f"out_{first}.cgns").encode()
        step, t, nst = C.c_int64(), C.c_double(), C.c_int()
        assert G.FlucaCGNSReadInfo(f, None, C.byref(step), C.byref(t), C.byref(nst)) == 0
        assert step.value == last and nst.value == 2 and abs(t.value - 0.01 * last) < 1e-15
        a = np.empty(N[0] * N[1] * N[2])
        assert G.FlucaCGNSReadCellField(f, C.byref(lay), first, b"Pressure", a.ctypes.data) == 0
    assert G.FlucaCGNSReadCellField(f, C.byref(lay), 3, b"Pressure", a.ctypes.data) == 0 and np.array_equal(a, final["p"])
    assert not (tmp_path / "out_1.cgns").exists() and not (tmp_path / "out_3.cgns").exists()


def test_nssolve_stops_at_max_time(H):
    """-ns_max_time / NSSetMaxTime: NSSolve ends at whichever of max_steps and max_time comes first (nsbasic.c:333-343)."""
    keep = []
    mesh, ns, _ = make_ns(H, 100, keep)
    mt = C.c_double()
    assert H.lib.NSGetMaxTime(ns, C.byref(mt)) == 0 and mt.value > 1e300          # PETSC_MAX_REAL: not set
    assert H.lib.NSSetMaxTime(ns, 0.025) == 0
    assert H.lib.NSSolve(ns) == 0
    step, t = C.c_int64(), C.c_double()
    H.lib.NSGetTimeStep(ns, C.byref(step)), H.lib.NSGetTime(ns, C.byref(t))
    assert step.value == 3 and abs(t.value - 0.03) < 1e-15                         # dt = 0.01: t = 0.03 is the first time >= 0.025
    H.lib.NSDestroy(C.byref(ns)), H.lib.MeshDestroy(C.byref(mesh))
