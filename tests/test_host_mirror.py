"""The C host mirror (include/fluca_host.h) used the way the reference's drivers use Mesh / NS
(fluca/tests/cavity_flow/cavity_flow_3d.c).  CPU tests cover the host logic; the -m gpu test runs PCApply_ABF's
pressure half against the oracle."""
import ctypes as C

import numpy as np
import pytest

P = C.c_void_p


@pytest.fixture(scope="module")
def H():
    from fluca_amd import build
    build.build()
    from fluca_amd import hostapi
    return hostapi


def cavity_mesh(H, opts=(), rank=0, size=1):
    mesh = P()
    # MeshCartCreate3d(PETSC_COMM_WORLD, NONE, NONE, NONE, 64, 64, 32, PETSC_DECIDE x3, NULL x3, &mesh)  cavity_flow_3d.c:39
    assert H.lib.MeshCartCreate3d(0, 0, 0, 64, 64, 32, H.FL_DECIDE, H.FL_DECIDE, H.FL_DECIDE, None, None, None, C.byref(mesh)) == 0
    assert H.lib.MeshSetRank(mesh, rank, size) == 0
    argc, av = H.argv(*opts)
    assert H.lib.MeshSetFromOptions(mesh, argc, av) == 0
    assert H.lib.MeshSetUp(mesh) == 0
    assert H.lib.MeshCartSetUniformCoordinates(mesh, 0., 1., 0., 1., 0., 0.5) == 0           # cavity_flow_3d.c:42
    return mesh


def set_cavity_bcs(H, ns, mesh):
    idx = {}
    for loc in range(6):
        i = C.c_int()
        assert H.lib.MeshCartGetBoundaryIndex(mesh, loc, C.byref(i)) == 0
        idx[loc] = i.value
    wall = H.NSBoundaryCondition(type=H.NS_BC_VELOCITY)
    sym = H.NSBoundaryCondition(type=H.NS_BC_SYMMETRY)
    for loc in (H.MESHCART_LEFT, H.MESHCART_RIGHT, H.MESHCART_DOWN, H.MESHCART_UP, H.MESHCART_FRONT):
        assert H.lib.NSSetBoundaryCondition(ns, idx[loc], wall) == 0
    assert H.lib.NSSetBoundaryCondition(ns, idx[H.MESHCART_BACK], sym) == 0               # cavity_flow_3d.c:72-77


def test_mesh_options_and_decomposition(H):
    mesh = cavity_mesh(H, ("-cart_grid_x", 10, "-cart_grid_y", 7, "-cart_grid_z", 5, "-cart_ranks_x", 3, "-cart_ranks_y", 2, "-cart_ranks_z", 1), rank=4, size=6)
    M, N, Pz = C.c_int64(), C.c_int64(), C.c_int64()
    assert H.lib.MeshCartGetGlobalSizes(mesh, C.byref(M), C.byref(N), C.byref(Pz)) == 0
    assert (M.value, N.value, Pz.value) == (10, 7, 5)
    c = [C.c_int64() for _ in range(6)]
    assert H.lib.MeshCartGetCorners(mesh, *[C.byref(v) for v in c]) == 0
    assert [v.value for v in c] == [4, 4, 0, 3, 3, 5]            # rank 4 = coord (1,1,0): x [4,7), y [4,7)
    f = [C.c_int() for _ in range(3)]
    l = [C.c_int() for _ in range(3)]
    H.lib.MeshCartGetIsFirstRank(mesh, *[C.byref(v) for v in f])
    H.lib.MeshCartGetIsLastRank(mesh, *[C.byref(v) for v in l])
    assert [v.value for v in f] == [0, 0, 1] and [v.value for v in l] == [0, 1, 1]
    nb = C.c_int()
    assert H.lib.MeshGetNumberBoundaries(mesh, C.byref(nb)) == 0 and nb.value == 6
    assert H.lib.MeshDestroy(C.byref(mesh)) == 0 and not mesh.value


def test_cart_refine_options(H):
    """-cart_refine n / -cart_refine_{x,y,z} f (fluca/src/mesh/impl/cart/cart.c:37-52): global sizes (and ownership ranges) times f^n, default factor 2"""
    mesh = cavity_mesh(H, ("-cart_grid_x", 10, "-cart_grid_y", 7, "-cart_grid_z", 5, "-cart_refine", 2, "-cart_refine_y", 3))
    M, N, Pz = C.c_int64(), C.c_int64(), C.c_int64()
    assert H.lib.MeshCartGetGlobalSizes(mesh, C.byref(M), C.byref(N), C.byref(Pz)) == 0
    assert (M.value, N.value, Pz.value) == (40, 63, 20)
    r = [C.c_int64() for _ in range(3)]
    assert H.lib.MeshCartGetRefinementFactor(mesh, *[C.byref(v) for v in r]) == 0 and [v.value for v in r] == [2, 3, 2]
    assert H.lib.MeshDestroy(C.byref(mesh)) == 0
    # ownership ranges grow with the sizes: 2 ranks in x owning 6 + 4 cells -> 12 + 8 after one refinement
    mesh = P()
    lx = (C.c_int64 * 2)(6, 4)
    assert H.lib.MeshCartCreate3d(0, 0, 0, 10, 8, 8, 2, 1, 1, lx, None, None, C.byref(mesh)) == 0
    assert H.lib.MeshSetRank(mesh, 1, 2) == 0
    argc, av = H.argv("-cart_refine", 1)
    assert H.lib.MeshSetFromOptions(mesh, argc, av) == 0 and H.lib.MeshSetUp(mesh) == 0
    c = [C.c_int64() for _ in range(6)]
    assert H.lib.MeshCartGetCorners(mesh, *[C.byref(v) for v in c]) == 0
    assert [v.value for v in c] == [12, 0, 0, 8, 16, 16]
    assert H.lib.MeshDestroy(C.byref(mesh)) == 0
    # the setter is refused once the mesh is set up, a factor < 1 in the options is out of range
    mesh = cavity_mesh(H)
    assert H.lib.MeshCartSetRefinementFactor(mesh, 2, 2, 2) == H.ERR_ARG_WRONGSTATE
    H.lib.MeshDestroy(C.byref(mesh))
    mesh = P()
    H.lib.MeshCartCreate3d(0, 0, 0, 8, 8, 8, H.FL_DECIDE, H.FL_DECIDE, H.FL_DECIDE, None, None, None, C.byref(mesh))
    argc, av = H.argv("-cart_refine_x", 0)
    assert H.lib.MeshSetFromOptions(mesh, argc, av) == H.ERR_ARG_OUTOFRANGE
    H.lib.MeshDestroy(C.byref(mesh))


def test_mesh_errors_match_reference_behaviour(H):
    mesh = P()
    H.lib.MeshCartCreate3d(0, 0, 0, 8, 8, 8, -1, -1, -1, None, None, None, C.byref(mesh))
    # "This function must be called after MeshSetUp()"  (cart.c:462)
    assert H.lib.MeshCartSetUniformCoordinates(mesh, 0., 1., 0., 1., 0., 1.) == H.ERR_ARG_WRONGSTATE
    argc, av = H.argv("-cart_boundary_type_x", "bogus")
    assert H.lib.MeshSetFromOptions(mesh, argc, av) == H.ERR_ARG_WRONG
    assert H.lib.MeshSetType(mesh, b"tetra") == H.ERR_ARG_UNKNOWN_TYPE
    i = C.c_int()
    assert H.lib.MeshCartGetBoundaryIndex(mesh, 9, C.byref(i)) == H.ERR_ARG_WRONG      # "Invalid boundary location"
    H.lib.MeshSetRank(mesh, 0, 3)
    argc, av = H.argv("-cart_ranks_x", 2, "-cart_ranks_y", 2, "-cart_ranks_z", 1)     # 4 ranks requested, job has 3
    H.lib.MeshSetFromOptions(mesh, argc, av)
    assert H.lib.MeshSetUp(mesh) == H.ERR_ARG_WRONG
    H.lib.MeshDestroy(C.byref(mesh))


def test_ns_registry_options_and_state_errors(H):
    ns = P()
    assert H.lib.NSCreate(C.byref(ns)) == 0
    assert H.lib.NSSetType(ns, b"fsm") == H.ERR_ARG_UNKNOWN_TYPE      # only "cnlinear" is registered (nsreg.c:17)
    assert H.lib.NSSetType(ns, b"cnlinear") == 0
    t = C.c_char_p()
    H.lib.NSGetType(ns, C.byref(t))
    assert t.value == b"cnlinear"
    assert H.lib.NSSetUp(ns) == H.ERR_ARG_WRONGSTATE                   # "Mesh not set"
    bc = H.NSBoundaryCondition(type=H.NS_BC_VELOCITY)
    assert H.lib.NSSetBoundaryCondition(ns, 0, bc) == H.ERR_ARG_WRONGSTATE
    mesh = cavity_mesh(H)
    assert H.lib.NSSetMesh(ns, mesh) == 0
    assert H.lib.NSSetBoundaryCondition(ns, 6, bc) == H.ERR_ARG_OUTOFRANGE
    argc, av = H.argv("-ns_density", 2.0, "-ns_viscosity", 0.01, "-ns_time_step_size", 1e-3, "-ns_max_steps", 5,
                      "-ns_abf_schur_ksp_type", "bcgs", "-ns_abf_schur_pc_type", "none", "-ns_abf_schur_ksp_rtol", 1e-8,
                      "-ns_abf_schur_ksp_max_it", 77, "-ns_abf_schur_ksp_norm_type", "preconditioned")
    assert H.lib.NSSetFromOptions(ns, argc, av) == 0
    o = C.POINTER(H.capi.fl_ksp_opts)()
    assert H.lib.NSGetSchurKSPOptions(ns, C.byref(o)) == 0
    assert (o.contents.type, o.contents.pc, o.contents.rtol, o.contents.maxit) == (1, 0, 1e-8, 77)
    argc, av = H.argv("-ns_abf_schur_ksp_type", "gmres")
    assert H.lib.NSSetFromOptions(ns, argc, av) == H.ERR_ARG_UNKNOWN_TYPE
    argc, av = H.argv("-ns_pc_abf_schur_ainv_type", "DIAG", "-ns_pc_abf_upper_ainv_type", "rowsum")     # PCABFAinvType: ID, DIAG, ROWSUM
    assert H.lib.NSSetFromOptions(ns, argc, av) == 0
    argc, av = H.argv("-ns_pc_abf_schur_ainv_type", "lumped")
    assert H.lib.NSSetFromOptions(ns, argc, av) == H.ERR_ARG_UNKNOWN_TYPE
    rho, mu, dt, ms, flg, why = C.c_double(), C.c_double(), C.c_double(), C.c_int64(), C.c_int(), C.c_int()
    assert H.lib.NSGetDensity(ns, C.byref(rho)) == 0 and H.lib.NSGetViscosity(ns, C.byref(mu)) == 0 and H.lib.NSGetTimeStepSize(ns, C.byref(dt)) == 0
    assert (rho.value, mu.value, dt.value) == (2.0, 0.01, 1e-3)
    assert H.lib.NSGetMaxSteps(ns, C.byref(ms)) == 0 and ms.value == 5
    assert H.lib.NSGetErrorIfStepFailed(ns, C.byref(flg)) == 0 and flg.value == 1          # nsbasic.c:46
    argc, av = H.argv("-ns_error_if_step_failed", "0", "-ns_max_time", 0.5)
    assert H.lib.NSSetFromOptions(ns, argc, av) == 0
    assert H.lib.NSGetErrorIfStepFailed(ns, C.byref(flg)) == 0 and flg.value == 0
    assert H.lib.NSGetConvergedReason(ns, C.byref(why)) == 0 and why.value == 0             # NS_CONVERGED_ITERATING
    assert H.lib.NSSetTimeStep(ns, 5) == 0 and H.lib.NSGetConvergedReason(ns, C.byref(why)) == 0 and why.value == 2   # NS_CONVERGED_ITS
    assert H.lib.NSSetTimeStep(ns, 0) == 0 and H.lib.NSSetTime(ns, 0.75) == 0
    assert H.lib.NSGetConvergedReason(ns, C.byref(why)) == 0 and why.value == 1             # NS_CONVERGED_TIME
    assert H.lib.NSSetTime(ns, 0.0) == 0 and H.lib.NSSetTimeStep(ns, -1) == H.ERR_ARG_OUTOFRANGE
    # PetscOptionsBool: a bare flag means true -- as the last argument too, and in front of another option -- and a value must be a logical one
    for opts, want in ((("-ns_abf_schur_ksp_cg_single_reduction",), 1), (("-ns_abf_schur_ksp_cg_single_reduction", "false"), 0),
                       (("-ns_abf_schur_ksp_cg_single_reduction", "-ns_max_steps", 5), 1), (("-ns_abf_schur_ksp_cg_single_reduction", "0"), 0),
                       (("-ns_max_steps", 5, "-ns_abf_schur_ksp_cg_single_reduction", "yes"), 1)):
        argc, av = H.argv(*opts)
        assert H.lib.NSSetFromOptions(ns, argc, av) == 0 and o.contents.cg_single_reduction == want, opts
    argc, av = H.argv("-ns_abf_schur_ksp_cg_single_reduction", "maybe")
    assert H.lib.NSSetFromOptions(ns, argc, av) == H.ERR_ARG_WRONG
    argc, av = H.argv("-ns_error_if_step_failed")
    assert H.lib.NSSetFromOptions(ns, argc, av) == 0 and H.lib.NSGetErrorIfStepFailed(ns, C.byref(flg)) == 0 and flg.value == 1
    assert H.lib.NSStep(ns) == H.ERR_ARG_WRONGSTATE                    # before NSSetUp
    argc, av = H.argv("-ns_ksp_type", "fgmres")                        # outer KSP: gmres (default), richardson, preonly
    assert H.lib.NSSetFromOptions(ns, argc, av) == H.ERR_SUP
    argc, av = H.argv("-ns_ksp_type", "gmres", "-ns_ksp_gmres_restart", 12)
    assert H.lib.NSSetFromOptions(ns, argc, av) == 0
    argc, av = H.argv("-ns_ksp_type", "preonly", "-ns_ksp_rtol", 1e-7)
    assert H.lib.NSSetFromOptions(ns, argc, av) == 0
    assert H.lib.NSDestroy(C.byref(ns)) == 0
    H.lib.MeshDestroy(C.byref(mesh))


@pytest.mark.gpu
def test_pressure_half_of_pcapply_abf_matches_oracle(H):
    """cavity_flow_3d-style set-up at 32x24x16, then PCApply_ABF stage 1 (without kspA) + stage 2 + pressure update."""
    import torch
    from oracle import fluca_oracle as fo
    n = (32, 24, 16)
    mesh = cavity_mesh(H, ("-cart_grid_x", n[0], "-cart_grid_y", n[1], "-cart_grid_z", n[2]))
    ns = P()
    assert H.lib.NSCreate(C.byref(ns)) == 0
    assert H.lib.NSSetType(ns, b"cnlinear") == 0
    assert H.lib.NSSetMesh(ns, mesh) == 0
    assert H.lib.NSSetDensity(ns, 1.0) == 0 and H.lib.NSSetViscosity(ns, 0.01) == 0
    set_cavity_bcs(H, ns, mesh)
    argc, av = H.argv("-ns_time_step_size", 1e-3, "-ns_abf_schur_ksp_type", "cg", "-ns_abf_schur_pc_type", "jacobi", "-ns_abf_schur_ksp_rtol", 1e-8)
    assert H.lib.NSSetFromOptions(ns, argc, av) == 0
    assert H.lib.NSSetUp(ns) == 0
    needs = C.c_int()
    assert H.lib.NSGetNeedsNullSpace(ns, C.byref(needs)) == 0 and needs.value == 1
    sz = (C.c_int64 * 4)()
    assert H.lib.NSGetLocalSizes(ns, sz) == 0
    bc = [fo.BC_VELOCITY] * 4 + [fo.BC_SYMMETRY, fo.BC_VELOCITY]
    g = fo.Grid.uniform(n, [(0, 1), (0, 1), (0, 0.5)], bc, 1e-3)
    assert tuple(sz) == (g.ncell,) + tuple(g.nface)
    rng = np.random.default_rng(8)
    # a discretely divergence-compatible V*: V* = kappa Gst q  ->  Srhs = -D V* = S q
    q = rng.standard_normal(g.ncell)
    q -= q.mean()
    Vs = g.apply_gst(q)
    vs = [rng.standard_normal(g.ncell) for _ in range(3)]
    dev = lambda a: torch.as_tensor(np.ascontiguousarray(a), device="cuda")
    Vd, vd = [dev(a) for a in Vs], [dev(a) for a in vs]
    dp = torch.zeros(g.ncell, dtype=torch.float64, device="cuda")
    torch.cuda.synchronize()
    st = H.capi.fl_ksp_stats()
    arr = lambda ts: (C.c_void_p * 3)(*[t.data_ptr() for t in ts])
    assert H.lib.NSPressureCorrection(ns, arr(vd), arr(Vd), None, C.c_void_p(dp.data_ptr()), C.byref(st)) == 0
    assert st.reason == 2 and st.iters > 3
    S = g.assemble_S()
    b = g.rhs(*Vs)
    xo, io = S.solve(b, rtol=1e-8)
    assert abs(st.iters - io["iters"]) <= 2
    dph = dp.cpu().numpy()
    assert np.linalg.norm((dph - dph.mean()) - (xo - xo.mean())) <= 1e-5 * np.linalg.norm(xo)
    # stage 2 against the oracle operators applied to the GPU's own dp
    Gst, Gc = g.apply_gst(dph), g.apply_G(dph)
    for d in range(3):
        assert abs(Vd[d].cpu().numpy() - (Vs[d] - Gst[d])).max() <= 1e-11 * max(1.0, abs(Vs[d]).max())
        assert abs(vd[d].cpu().numpy() - (vs[d] - Gc[d])).max() <= 1e-11 * max(1.0, abs(vs[d]).max())
    # the projected face velocity is discretely divergence-free to the solver tolerance: || D V || <= 10 rtol ||b||
    div = g.rhs(*[t.cpu().numpy() for t in Vd])
    assert np.linalg.norm(div) <= 10 * 1e-8 * np.linalg.norm(b) + 1e-12
    # pressure update, first step then second (cnlinearcart3d.c:2846-2854)
    p0 = dev(rng.standard_normal(g.ncell))
    phalf, p = torch.zeros_like(p0), torch.zeros_like(p0)
    torch.cuda.synchronize()
    ptr = lambda t: C.c_void_p(t.data_ptr())
    assert H.lib.NSUpdatePressure(ns, ptr(dp), ptr(p0), ptr(phalf), ptr(p)) == 0
    torch.cuda.synchronize()
    assert np.allclose(p.cpu().numpy(), p0.cpu().numpy() + 2 * dph, rtol=1e-15, atol=1e-15)
    step, t = C.c_int64(), C.c_double()
    H.lib.NSGetTimeStep(ns, C.byref(step))
    H.lib.NSGetTime(ns, C.byref(t))
    assert step.value == 1 and t.value == 1e-3
    ph1 = phalf.cpu().numpy().copy()
    assert H.lib.NSUpdatePressure(ns, ptr(dp), None, ptr(phalf), ptr(p)) == 0
    torch.cuda.synchronize()
    assert np.allclose(p.cpu().numpy(), ph1 + 1.5 * dph, rtol=1e-15, atol=1e-15)
    assert H.lib.NSDestroy(C.byref(ns)) == 0
    H.lib.MeshDestroy(C.byref(mesh))


@pytest.mark.gpu
def test_outlet_bc_vector_from_host_callbacks(H):
    import torch
    from oracle import fluca_oracle as fo
    n = (9, 12, 7)
    mesh = P()
    assert H.lib.MeshCartCreate3d(0, 0, 1, n[0], n[1], n[2], -1, -1, -1, None, None, None, C.byref(mesh)) == 0
    assert H.lib.MeshSetUp(mesh) == 0
    assert H.lib.MeshCartSetUniformCoordinates(mesh, 0., 2., 0., 1., 0., 1.) == 0
    ns = P()
    H.lib.NSCreate(C.byref(ns))
    H.lib.NSSetMesh(ns, mesh)
    H.lib.NSSetTimeStepSize(ns, 0.25)

    @H.BCFunc
    def outlet_pressure(dim, t, x, val, ctx):
        val[0] = 3.0 + x[1] - 2.0 * x[2] + t
        return 0

    H.lib.NSSetBoundaryCondition(ns, 0, H.NSBoundaryCondition(type=H.NS_BC_VELOCITY))
    H.lib.NSSetBoundaryCondition(ns, 1, H.NSBoundaryCondition(type=H.NS_BC_PRESSURE_OUTLET, pressure=outlet_pressure))
    for b in (2, 3):
        H.lib.NSSetBoundaryCondition(ns, b, H.NSBoundaryCondition(type=H.NS_BC_VELOCITY))
    for b in (4, 5):
        H.lib.NSSetBoundaryCondition(ns, b, H.NSBoundaryCondition(type=H.NS_BC_PERIODIC))
    assert H.lib.NSSetUp(ns) == 0
    needs = C.c_int()
    H.lib.NSGetNeedsNullSpace(ns, C.byref(needs))
    assert needs.value == 0                                               # an outlet pins the pressure level (nsbasic.c:226-229)
    g = fo.Grid.uniform(n, [(0, 2), (0, 1), (0, 1)], [1, 2, 1, 1, 3, 3], 0.25)
    V = [torch.zeros(nf, dtype=torch.float64, device="cuda") for nf in g.nface]
    torch.cuda.synchronize()
    assert H.lib.NSComputeStaggeredPressureGradientBC(ns, 0.5, (C.c_void_p * 3)(*[t.data_ptr() for t in V])) == 0
    Vx = V[0].cpu().numpy().reshape(n[2], n[1], n[0] + 1)
    yc = (np.arange(n[1]) + 0.5) / n[1]
    zc = (np.arange(n[2]) + 0.5) / n[2]
    pb = 3.0 + yc[None, :] - 2.0 * zc[:, None] + 0.5
    assert np.allclose(Vx[:, :, -1], g.gst_bc_coeff(0, 1) * pb, rtol=1e-14)
    assert abs(Vx[:, :, :-1]).max() == 0 and float(V[1].abs().max()) == 0 and float(V[2].abs().max()) == 0
    H.lib.NSDestroy(C.byref(ns))
    H.lib.MeshDestroy(C.byref(mesh))


@pytest.mark.gpu
def test_c_driver_runs_without_python(H):
    """examples/cavity_pressure_step.c = the reference's cavity driver on the C host mirror, as its own process."""
    import os
    import subprocess
    from fluca_amd import build
    exe = build.build_example()
    out = subprocess.run([exe, "-cart_grid_x", "48", "-cart_grid_y", "40", "-cart_grid_z", "24", "-ns_abf_schur_ksp_rtol", "1e-9"],
                         capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "1 NS dt 0.001 time 0.001" in out.stdout and "reason 2" in out.stdout


def test_tracing_hooks_find_roctx(H):
    """NSSetUp / NSStep / NSFormFunction / NSFormJacobian emit roctx ranges named like the reference's PetscLogEvents
    (nspkg.c:21-24) when libroctx64 can be loaded -- it ships with ROCm, so it must be found in this image."""
    assert H.lib.FlucaTraceEnabled() == 1


def test_momentum_options_and_state_checks(H):
    """-ns_abf_momentum_* (abfpc.c:205): what is built is accepted, PETSc's own defaults are refused as unsupported."""
    mesh = cavity_mesh(H, ("-cart_grid_x", 8, "-cart_grid_y", 8, "-cart_grid_z", 8))
    ns = P()
    assert H.lib.NSCreate(C.byref(ns)) == 0 and H.lib.NSSetMesh(ns, mesh) == 0
    argc, av = H.argv("-ns_time_step_size", 1e-3, "-ns_abf_momentum_ksp_type", "bcgs", "-ns_abf_momentum_pc_type", "none", "-ns_abf_momentum_ksp_rtol", 1e-9, "-ns_abf_momentum_ksp_max_it", 77)
    assert H.lib.NSSetFromOptions(ns, argc, av) == 0
    o = C.POINTER(H.capi.fl_ksp_opts)()
    assert H.lib.NSGetMomentumKSPOptions(ns, C.byref(o)) == 0
    assert (o.contents.type, o.contents.pc, o.contents.rtol, o.contents.maxit) == (H.capi.KSP_BCGS, 0, 1e-9, 77)
    for bad, rc in (("fgmres", 56), ("nonsense", 86)):         # PETSC_ERR_SUP, PETSC_ERR_ARG_UNKNOWN_TYPE
        argc, av = H.argv("-ns_abf_momentum_ksp_type", bad)
        assert H.lib.NSSetFromOptions(ns, argc, av) == rc
    argc, av = H.argv("-ns_abf_momentum_ksp_type", "gmres", "-ns_abf_momentum_ksp_gmres_restart", 12)   # the reference's default type of kspA
    assert H.lib.NSSetFromOptions(ns, argc, av) == 0
    assert (o.contents.type, o.contents.gmres_restart) == (H.capi.KSP_GMRES, 12)
    argc, av = H.argv("-ns_abf_momentum_ksp_type", "bcgs")
    assert H.lib.NSSetFromOptions(ns, argc, av) == 0
    argc, av = H.argv("-ns_abf_momentum_pc_type", "ilu")
    assert H.lib.NSSetFromOptions(ns, argc, av) == 56
    # before NSSetUp / NSSetPreviousState there is no A
    three = (C.c_void_p * 3)()
    assert H.lib.NSApplyPreconditioner(ns, C.c_void_p(8), three, None, C.c_void_p(8), three, C.c_void_p(8), None) == 73   # PETSC_ERR_ARG_WRONGSTATE
    assert H.lib.NSDestroy(C.byref(ns)) == 0 and H.lib.MeshDestroy(C.byref(mesh)) == 0


@pytest.mark.gpu
def test_full_pcapply_abf_through_the_mirror(H):
    """NSSetPreviousState + NSApplyPreconditioner == PCApply_ABF (abfpc.c:48-111) composed from the oracle."""
    import torch
    from oracle import fluca_oracle as fo
    n = (24, 16, 12)
    mesh = cavity_mesh(H, ("-cart_grid_x", n[0], "-cart_grid_y", n[1], "-cart_grid_z", n[2]))
    ns = P()
    assert H.lib.NSCreate(C.byref(ns)) == 0 and H.lib.NSSetType(ns, b"cnlinear") == 0 and H.lib.NSSetMesh(ns, mesh) == 0
    rho, mu, dt = 1.3, 0.02, 2e-3
    assert H.lib.NSSetDensity(ns, rho) == 0 and H.lib.NSSetViscosity(ns, mu) == 0
    set_cavity_bcs(H, ns, mesh)
    argc, av = H.argv("-ns_time_step_size", dt, "-ns_abf_schur_ksp_rtol", 1e-10, "-ns_abf_momentum_ksp_rtol", 1e-11)
    assert H.lib.NSSetFromOptions(ns, argc, av) == 0 and H.lib.NSSetUp(ns) == 0
    bc = [fo.BC_VELOCITY] * 4 + [fo.BC_SYMMETRY, fo.BC_VELOCITY]
    g = fo.Grid.uniform(n, [(0, 1), (0, 1), (0, 0.5)], bc, dt / rho)
    rng = np.random.default_rng(12)
    V0 = [rng.standard_normal(g.nface[d]) for d in range(3)]
    W = [rng.standard_normal(g.nface[d]) for c in range(3) for d in range(3)]
    momrhs = rng.standard_normal(3 * g.ncell)
    dev = lambda a: torch.as_tensor(np.ascontiguousarray(a), device="cuda")
    arr = lambda ts: (C.c_void_p * len(ts))(*[t.data_ptr() for t in ts])
    V0d, Wd, md = [dev(a) for a in V0], [dev(a) for a in W], dev(momrhs)
    v = torch.zeros(3 * g.ncell, dtype=torch.float64, device="cuda")
    p = torch.zeros(g.ncell, dtype=torch.float64, device="cuda")
    Vd = [torch.zeros(g.nface[d], dtype=torch.float64, device="cuda") for d in range(3)]
    torch.cuda.synchronize()
    assert H.lib.NSSetPreviousState(ns, arr(V0d), arr(Wd)) == 0
    st = (H.capi.fl_ksp_stats * 2)()
    assert H.lib.NSApplyPreconditioner(ns, C.c_void_p(md.data_ptr()), None, None, C.c_void_p(v.data_ptr()), arr(Vd), C.c_void_p(p.data_ptr()), st) == 0
    assert st[0].reason == 2 and st[1].reason == 2
    A = g.assemble_momentum(1.0, dt, -0.5 * mu * dt / rho, V0, W)
    vs, i0 = A.solve(momrhs, ksp=fo.KSP_BCGS, pc=fo.PC_JACOBI, nullspace=False, rtol=1e-11, maxit=10000)
    Vs = g.apply_T(vs)
    S = g.assemble_S()
    po, i1 = S.solve(g.rhs(*Vs), rtol=1e-10)
    assert abs(st[0].iters - i0["iters"]) <= 2 and abs(st[1].iters - i1["iters"]) <= max(3, i1["iters"] // 20)
    pg = p.cpu().numpy()
    assert np.linalg.norm((pg - pg.mean()) - (po - po.mean())) <= 1e-6 * np.linalg.norm(po - po.mean())
    v_ref = vs - np.concatenate(g.apply_G(po))
    assert np.linalg.norm(v.cpu().numpy() - v_ref) <= 1e-7 * np.linalg.norm(v_ref)
    Gst = g.apply_gst(po)
    for d in range(3):
        assert np.linalg.norm(Vd[d].cpu().numpy() - (Vs[d] - Gst[d])) <= 1e-7 * np.linalg.norm(Vs[d])
    assert H.lib.NSDestroy(C.byref(ns)) == 0 and H.lib.MeshDestroy(C.byref(mesh)) == 0


def _tgv_mirror(H, n, nsteps, walls, t_final=0.4, rho=1.0, mu=0.1, ksp="richardson", extra=(), info=None):
    """The reference's own check (fluca/tests/taylor_green_vortex/taylor_green_vortex.c) in 3-D: Taylor-Green vortex in
    x-y, periodic in z; walls=True puts time-dependent VELOCITY conditions from the exact solution on the four side walls
    (the boundary-condition vectors of L, C, B and T), walls=False is the fully periodic box."""
    import torch
    L = 2 * np.pi
    nu = mu / rho
    bt = 0 if walls else 1                      # MESHCART_BOUNDARY_NONE / PERIODIC
    mesh = P()
    assert H.lib.MeshCartCreate3d(bt, bt, 1, n, n, 4, -1, -1, -1, None, None, None, C.byref(mesh)) == 0
    assert H.lib.MeshSetUp(mesh) == 0
    assert H.lib.MeshCartSetUniformCoordinates(mesh, 0., L, 0., L, 0., L * 4 / n) == 0
    ns = P()
    assert H.lib.NSCreate(C.byref(ns)) == 0 and H.lib.NSSetType(ns, b"cnlinear") == 0 and H.lib.NSSetMesh(ns, mesh) == 0
    assert H.lib.NSSetDensity(ns, rho) == 0 and H.lib.NSSetViscosity(ns, mu) == 0

    @H.BCFunc
    def velocity(dim, t, x, val, ctx):
        d = np.exp(-2.0 * nu * t)
        val[0] = np.sin(x[0]) * np.cos(x[1]) * d
        val[1] = -np.cos(x[0]) * np.sin(x[1]) * d
        val[2] = 0.0
        return 0

    for b in range(4):
        bc = H.NSBoundaryCondition(type=H.NS_BC_VELOCITY, velocity=velocity) if walls else H.NSBoundaryCondition(type=H.NS_BC_PERIODIC)
        assert H.lib.NSSetBoundaryCondition(ns, b, bc) == 0
    for b in (4, 5):
        assert H.lib.NSSetBoundaryCondition(ns, b, H.NSBoundaryCondition(type=H.NS_BC_PERIODIC)) == 0
    dt = t_final / nsteps
    argc, av = H.argv("-ns_time_step_size", dt, "-ns_max_steps", nsteps, "-ns_ksp_type", ksp, "-ns_ksp_rtol", 1e-8,
                      "-ns_abf_schur_ksp_rtol", 1e-10, "-ns_abf_momentum_ksp_rtol", 1e-10, *extra)
    assert H.lib.NSSetFromOptions(ns, argc, av) == 0 and H.lib.NSSetUp(ns) == 0
    v, p, V = P(), P(), (C.c_void_p * 3)()
    assert H.lib.NSGetSolutionArrays(ns, C.byref(v), V, C.byref(p)) == 0
    h = L / n
    xc, xf = (np.arange(n) + 0.5) * h, np.arange(n + 1) * h
    Z = np.ones((4, 1, 1))
    ex = lambda xs, ys, t: (Z * (np.sin(xs)[None, None, :] * np.cos(ys)[None, :, None]) * np.exp(-2 * nu * t),
                            Z * (-np.cos(xs)[None, None, :] * np.sin(ys)[None, :, None]) * np.exp(-2 * nu * t))
    u0, w0 = ex(xc, xc, 0.0)
    nfx = n + 1 if walls else n
    put = lambda ptr, a: H.capi.check(H.capi.lib.fl_memcpy_h2d(0, ptr, np.ascontiguousarray(a, dtype=np.float64).ctypes.data_as(C.c_void_p), a.size * 8))
    put(v, np.stack([u0, w0, np.zeros_like(u0)]))
    put(C.c_void_p(V[0]), ex(xf[:nfx], xc, 0.0)[0])
    put(C.c_void_p(V[1]), ex(xc, xf[:nfx], 0.0)[1])
    X, Y = np.meshgrid(xc, xc, indexing="xy")
    put(p, Z * (rho / 4 * (np.cos(2 * X) + np.cos(2 * Y)))[None, :, :])
    assert H.lib.NSSolve(ns) == 0
    its, rn, reason = C.c_int(), C.c_double(), C.c_int()
    H.lib.NSGetLinearSolveInfo(ns, C.byref(its), C.byref(rn), C.byref(reason))
    step, t = C.c_int64(), C.c_double()
    H.lib.NSGetTimeStep(ns, C.byref(step))
    H.lib.NSGetTime(ns, C.byref(t))
    assert step.value == nsteps and abs(t.value - t_final) < 1e-12 and reason.value >= 0
    out = np.empty(3 * 4 * n * n)
    H.capi.check(H.capi.lib.fl_memcpy_d2h(0, out.ctypes.data_as(C.c_void_p), v, out.size * 8))
    vh = out.reshape(3, 4, n, n)
    ue, we = ex(xc, xc, t_final)
    err = np.sqrt(((vh[0] - ue) ** 2 + (vh[1] - we) ** 2).mean())
    wmax = np.abs(vh[2]).max()
    if info is not None:                        # the last step's inner iteration counts and the velocity itself
        ma, ms = C.c_int(), C.c_int()
        assert H.lib.NSGetInnerIterations(ns, C.byref(ma), C.byref(ms)) == 0
        info.update(kspA_its=ma.value, kspS_its=ms.value, v=vh.copy())
    H.lib.NSDestroy(C.byref(ns))
    H.lib.MeshDestroy(C.byref(mesh))
    return err, wmax, its.value


@pytest.mark.gpu
@pytest.mark.parametrize("walls", [False, True])
def test_nssolve_taylor_green_second_order(H, walls):
    e1, w1, _ = _tgv_mirror(H, 16, 4, walls)
    e2, w2, _ = _tgv_mirror(H, 32, 8, walls)
    assert w1 < 1e-10 and w2 < 1e-10
    assert e1 < 0.05 and e2 < e1 / 3.0, (e1, e2)


@pytest.mark.gpu
def test_nssolve_preonly_is_the_fractional_step_method(H):
    """-ns_ksp_type preonly applies PCApply_ABF once per step: still second-order accurate, slightly larger error."""
    e_r, _, its_r = _tgv_mirror(H, 32, 8, True, ksp="richardson")
    e_p, _, its_p = _tgv_mirror(H, 32, 8, True, ksp="preonly")
    assert its_p == 1 and its_r > 1
    assert e_p < 3 * e_r + 1e-3


@pytest.mark.gpu
@pytest.mark.parametrize("kspA", ["bcgs", "chebyshev"])
def test_fractional_step_from_a_velocity_guess_is_the_same_step_in_fewer_iterations(H, kspA):
    """-ns_abf_momentum_guess_previous / -ns_abf_momentum_guess_extrapolate (mirror only): kspA of the fractional step starts from v^n or from
    2 v^n - v^(n-1) and stops at the same test (|| M r || <= rtol || M momrhs ||): the same velocity to the solver tolerance, in fewer iterations,
    fewest with the extrapolated guess (the decaying vortex is smooth in time)."""
    runs = {}
    for name, extra in (("zero", ()), ("previous", ("-ns_abf_momentum_guess_previous",)), ("extrapolated", ("-ns_abf_momentum_guess_extrapolate",))):
        runs[name] = {}
        e, _, its = _tgv_mirror(H, 32, 8, True, ksp="preonly", extra=("-ns_abf_momentum_ksp_type", kspA) + extra, info=runs[name])
        runs[name]["err"] = e
        assert its == 1
    ref = runs["zero"]["v"]
    for name in ("previous", "extrapolated"):
        assert np.abs(runs[name]["v"] - ref).max() <= 1e-8 * np.abs(ref).max(), name
        assert abs(runs[name]["err"] - runs["zero"]["err"]) < 1e-8
    assert runs["extrapolated"]["kspA_its"] < runs["previous"]["kspA_its"] < runs["zero"]["kspA_its"], {k: r["kspA_its"] for k, r in runs.items()}


@pytest.mark.gpu
def test_nssolve_gmres_is_the_default_and_beats_richardson(H):
    """The reference's outer solver (nssol.c:21-29: GMRES, rtol, unpreconditioned norm => right preconditioning with PC_ABF):
    same answer as the Richardson iteration, in fewer applications of the preconditioner."""
    e_g, _, its_g = _tgv_mirror(H, 32, 8, True, ksp="gmres")
    e_r, _, its_r = _tgv_mirror(H, 32, 8, True, ksp="richardson")
    assert abs(e_g - e_r) < 1e-6
    assert 1 < its_g < its_r


@pytest.mark.gpu
def test_c_cavity_driver_full_time_steps(H):
    """examples/cavity_flow_3d.c: the reference's lid-driven cavity (cavity_flow_3d.c) stepping on the GPU from C."""
    import re
    import subprocess
    from fluca_amd import build
    exe = build.build_example(name="cavity_flow_3d")
    out = subprocess.run([exe, "-cart_grid_x", "32", "-cart_grid_y", "32", "-cart_grid_z", "16", "-ns_time_step_size", "1e-2", "-ns_max_steps", "12",
                          "-ns_ksp_rtol", "1e-6"], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stdout + out.stderr
    lines = [l for l in out.stdout.splitlines() if " NS time " in l]
    assert len(lines) == 12 and lines[-1].startswith("12 NS time 0.12")
    # the same run with the multigrid-preconditioned pressure solve gives the same flow
    out2 = subprocess.run([exe, "-cart_grid_x", "32", "-cart_grid_y", "32", "-cart_grid_z", "16", "-ns_time_step_size", "1e-2", "-ns_max_steps", "12",
                           "-ns_ksp_rtol", "1e-6", "-ns_abf_schur_pc_type", "mg"], capture_output=True, text=True, timeout=300)
    assert out2.returncode == 0, out2.stdout + out2.stderr
    ke2 = [float(re.search(r"kinetic energy (\S+)", l).group(1)) for l in out2.stdout.splitlines() if " NS time " in l]
    ke = [float(re.search(r"kinetic energy (\S+)", l).group(1)) for l in lines]
    assert all(b > a > 0 for a, b in zip(ke, ke[1:]))                 # the lid keeps feeding the flow at early times
    assert np.allclose(ke, ke2, rtol=1e-4)
    prof = [float(x) for x in out.stdout.splitlines()[-1].split(":")[1].split()]
    assert prof[-1] > 0.2 and min(prof) < 0.0                          # dragged along under the lid, return flow below
    assert all(abs(x) < 1.0 + 1e-9 for x in prof)


@pytest.mark.gpu
def test_c_cavity_driver_writes_cgns_like_the_reference_options(H, tmp_path):
    """-ns_monitor_solution cgns:<template> -viewer_cgns_batch_size 2 -ns_view_solution cgns:<file> on the C driver."""
    import subprocess
    from fluca_amd import build
    if not build.have_hdf5():
        pytest.skip("no HDF5 C library in this image")
    exe = build.build_example(name="cavity_flow_3d")
    out = subprocess.run([exe, "-cart_grid_x", "16", "-cart_grid_y", "16", "-cart_grid_z", "8", "-ns_time_step_size", "1e-2", "-ns_max_steps", "4",
                          "-ns_monitor_solution", f"cgns:{tmp_path}/mon_%d.cgns", "-ns_monitor_solution_interval", "2", "-viewer_cgns_batch_size", "2",
                          "-ns_view_solution", f"cgns:{tmp_path}/end.cgns"], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stdout + out.stderr
    G = H.load_cgns()
    # monitored steps 0, 2, 4 in batches of two: files 0 (steps 0, 2) and 4 (step 4)
    for name, last, count in (("mon_0.cgns", 2, 2), ("mon_4.cgns", 4, 1), ("end.cgns", 4, 1)):
        N, step, t, n = (C.c_int64 * 3)(), C.c_int64(), C.c_double(), C.c_int()
        assert G.FlucaCGNSReadInfo(str(tmp_path / name).encode(), N, C.byref(step), C.byref(t), C.byref(n)) == 0, name
        assert tuple(N) == (16, 16, 8) and step.value == last and n.value == count and abs(t.value - 0.01 * last) < 1e-15
    assert sorted(p.name for p in tmp_path.iterdir()) == ["end.cgns", "mon_0.cgns", "mon_4.cgns"]


@pytest.mark.gpu
@pytest.mark.parametrize("ainv", [(), ("-ns_pc_abf_schur_ainv_type", "DIAG", "-ns_pc_abf_upper_ainv_type", "DIAG"),
                                  ("-ns_pc_abf_schur_ainv_type", "rowsum"),
                                  ("-ns_abf_momentum_ksp_type", "gmres"),          # kspA as the reference runs it (abfpc.c:72), Jacobi for ILU
                                  ("-ns_abf_momentum_ksp_type", "gmres", "-ns_abf_momentum_ksp_gmres_restart", 4),
                                  ("-ns_keep_boundary_values", "false"),           # the callbacks swept again in every step
                                  # the first PCApply_ABF of a step starts kspA from the previous velocity (round 5): same converged step
                                  ("-ns_ksp_type", "richardson", "-ns_abf_momentum_guess_previous"),
                                  ("-ns_ksp_type", "richardson", "-ns_abf_momentum_guess_previous", "true", "-ns_abf_momentum_ksp_type", "chebyshev"),
                                  ("-ns_ksp_type", "richardson", "-ns_abf_momentum_guess_extrapolate")])
def test_nsstep_matches_the_oracle_step(H, ainv):
    """Velocity, face velocity and pressure after two lid-driven-cavity steps: the C mirror on the GPU vs the CPU oracle's
    composition of the same reference formulas (StepOracle), including the wall terms of L, C, B and T."""
    from oracle import fluca_oracle as fo
    n = (16, 12, 10)
    rho, mu, dt = 1.0, 0.05, 5e-3
    mesh = cavity_mesh(H, ("-cart_grid_x", n[0], "-cart_grid_y", n[1], "-cart_grid_z", n[2]))
    ns = P()
    assert H.lib.NSCreate(C.byref(ns)) == 0 and H.lib.NSSetType(ns, b"cnlinear") == 0 and H.lib.NSSetMesh(ns, mesh) == 0
    assert H.lib.NSSetDensity(ns, rho) == 0 and H.lib.NSSetViscosity(ns, mu) == 0

    def lid(t, x):
        return np.array([1.0 + 0.5 * t + 0.1 * x[0], 0.0, 0.2 * x[2]])       # a lid that is neither uniform nor steady

    @H.BCFunc
    def wall(dim, t, x, val, ctx):
        val[0] = val[1] = val[2] = 0.0
        return 0

    calls = []

    @H.BCFunc
    def moving(dim, t, x, val, ctx):
        calls.append(t)
        val[0], val[1], val[2] = lid(t, x)
        return 0

    idx = {}
    for loc in range(6):
        i = C.c_int()
        assert H.lib.MeshCartGetBoundaryIndex(mesh, loc, C.byref(i)) == 0
        idx[loc] = i.value
    for loc in (H.MESHCART_LEFT, H.MESHCART_RIGHT, H.MESHCART_DOWN, H.MESHCART_FRONT):
        assert H.lib.NSSetBoundaryCondition(ns, idx[loc], H.NSBoundaryCondition(type=H.NS_BC_VELOCITY, velocity=wall)) == 0
    assert H.lib.NSSetBoundaryCondition(ns, idx[H.MESHCART_UP], H.NSBoundaryCondition(type=H.NS_BC_VELOCITY, velocity=moving)) == 0
    assert H.lib.NSSetBoundaryCondition(ns, idx[H.MESHCART_BACK], H.NSBoundaryCondition(type=H.NS_BC_SYMMETRY)) == 0
    argc, av = H.argv("-ns_time_step_size", dt, "-ns_max_steps", 2, "-ns_ksp_rtol", 1e-9, "-ns_abf_schur_ksp_rtol", 1e-10,
                      "-ns_abf_momentum_ksp_rtol", 1e-10, "-ns_abf_schur_ksp_max_it", 20000, *ainv)
    # the Ainv types and the Krylov type of kspA only change the preconditioner: the converged step is the same
    assert H.lib.NSSetFromOptions(ns, argc, av) == 0 and H.lib.NSSetUp(ns) == 0
    assert H.lib.NSSolve(ns) == 0
    r0, r1 = C.c_double(), C.c_double()
    assert H.lib.NSGetLinearSolveResidualNorms(ns, C.byref(r0), C.byref(r1)) == 0
    assert r0.value > 0 and r1.value <= 1e-9 * r0.value                  # the normalised residual is what -ns_ksp_rtol bounds
    # sweeps of the lid's callback over its 16 x 10 faces in two steps: the values at a step's t + dt are the next step's values at t, and the several
    # boundary-condition vectors of a step share one sweep per time -- 0, dt, 2 dt; with -ns_keep_boundary_values false dt is swept once more
    sweeps = len(calls) / (n[0] * n[2])
    assert sweeps == (4 if "-ns_keep_boundary_values" in ainv else 3), sweeps
    assert sorted(set(np.round(np.array(calls) / dt).astype(int))) == [0, 1, 2]
    v, p, Vp = P(), P(), (C.c_void_p * 3)()
    assert H.lib.NSGetSolutionArrays(ns, C.byref(v), Vp, C.byref(p)) == 0
    bc = [fo.BC_VELOCITY] * 4 + [fo.BC_SYMMETRY, fo.BC_VELOCITY]
    g = fo.Grid.uniform(n, [(0, 1), (0, 1), (0, 0.5)], bc, dt / rho)

    def get(ptr, m):
        out = np.empty(m)
        H.capi.check(H.capi.lib.fl_memcpy_d2h(0, out.ctypes.data_as(C.c_void_p), ptr, m * 8))
        return out

    vg, pg = get(v, 3 * g.ncell), get(p, g.ncell)
    Vg = [get(C.c_void_p(Vp[d]), g.nface[d]) for d in range(3)]

    def velocity(b, t, X):
        if b == 3:
            return np.stack([lid(t, x) for x in X], axis=1)
        return np.zeros((3, len(X)))

    so = fo.StepOracle(g, dt, rho, mu, velocity, krylov_rtol=1e-10, outer_rtol=1e-9)
    vo, Vo, po = np.zeros(3 * g.ncell), [np.zeros(nf) for nf in g.nface], np.zeros(g.ncell)
    for _ in range(2):
        vo, Vo, po, info = so.step_once(vo, Vo, po)
    assert np.abs(vo).max() > 0.05
    assert np.linalg.norm(vg - vo) <= 1e-6 * np.linalg.norm(vo)
    for d in range(3):
        assert np.linalg.norm(Vg[d] - Vo[d]) <= 1e-6 * max(np.linalg.norm(Vo[d]), 1e-12)
    assert np.linalg.norm((pg - pg.mean()) - (po - po.mean())) <= 1e-5 * np.linalg.norm(po - po.mean())
    H.lib.NSDestroy(C.byref(ns))
    H.lib.MeshDestroy(C.byref(mesh))


def _channel(H, n, dt, nsteps, rho, mu, pout, opts=()):
    """Channel: parabolic VELOCITY inlet at x = 0, PRESSURE_OUTLET at x = Lx, no-slip walls in y, periodic z."""
    Lx, Ly = 2.0, 1.0
    mesh = P()
    assert H.lib.MeshCartCreate3d(0, 0, 1, n[0], n[1], n[2], -1, -1, -1, None, None, None, C.byref(mesh)) == 0
    assert H.lib.MeshSetUp(mesh) == 0
    assert H.lib.MeshCartSetUniformCoordinates(mesh, 0., Lx, 0., Ly, 0., Ly * n[2] / n[1]) == 0
    ns = P()
    assert H.lib.NSCreate(C.byref(ns)) == 0 and H.lib.NSSetType(ns, b"cnlinear") == 0 and H.lib.NSSetMesh(ns, mesh) == 0
    assert H.lib.NSSetDensity(ns, rho) == 0 and H.lib.NSSetViscosity(ns, mu) == 0

    @H.BCFunc
    def inlet(dim, t, x, val, ctx):
        val[0], val[1], val[2] = 4.0 * x[1] * (Ly - x[1]) / Ly ** 2, 0.0, 0.0
        return 0

    @H.BCFunc
    def wall(dim, t, x, val, ctx):
        val[0] = val[1] = val[2] = 0.0
        return 0

    @H.BCFunc
    def outlet(dim, t, x, val, ctx):
        val[0] = pout(t, x)
        return 0

    keep = (inlet, wall, outlet)
    bcs = [H.NSBoundaryCondition(type=H.NS_BC_VELOCITY, velocity=inlet), H.NSBoundaryCondition(type=H.NS_BC_PRESSURE_OUTLET, pressure=outlet),
           H.NSBoundaryCondition(type=H.NS_BC_VELOCITY, velocity=wall), H.NSBoundaryCondition(type=H.NS_BC_VELOCITY, velocity=wall),
           H.NSBoundaryCondition(type=H.NS_BC_PERIODIC), H.NSBoundaryCondition(type=H.NS_BC_PERIODIC)]
    for b in range(6):
        assert H.lib.NSSetBoundaryCondition(ns, b, bcs[b]) == 0
    argc, av = H.argv("-ns_time_step_size", dt, "-ns_max_steps", nsteps, "-ns_abf_schur_ksp_type", "bcgs", *opts)
    assert H.lib.NSSetFromOptions(ns, argc, av) == 0 and H.lib.NSSetUp(ns) == 0
    return mesh, ns, keep, (Lx, Ly)


def _fetch(H, ns, g):
    v, p, Vp = P(), P(), (C.c_void_p * 3)()
    assert H.lib.NSGetSolutionArrays(ns, C.byref(v), Vp, C.byref(p)) == 0

    def get(ptr, m):
        out = np.empty(m)
        H.capi.check(H.capi.lib.fl_memcpy_d2h(0, out.ctypes.data_as(C.c_void_p), ptr, m * 8))
        return out

    return get(v, 3 * g.ncell), [get(C.c_void_p(Vp[d]), g.nface[d]) for d in range(3)], get(p, g.ncell)


@pytest.mark.gpu
def test_nsstep_with_an_unsteady_outlet_matches_the_oracle_step(H):
    """PRESSURE_OUTLET: the G boundary vector in momrhs and the Rhie-Chow boundary terms of interprhs (cnlinearcart3d.c
    :2976-2984, :3013-3044), exercised with an outlet pressure that varies in space and time."""
    from oracle import fluca_oracle as fo
    n, rho, mu, dt = (12, 10, 4), 1.0, 0.05, 5e-3
    pout = lambda t, x: 0.3 * np.sin(3.0 * t) + 0.1 * x[1]
    mesh, ns, keep, (Lx, Ly) = _channel(H, n, dt, 2, rho, mu, pout,
                                        ("-ns_ksp_rtol", 1e-9, "-ns_abf_schur_ksp_rtol", 1e-11, "-ns_abf_momentum_ksp_rtol", 1e-11))
    assert H.lib.NSSolve(ns) == 0
    bc = [fo.BC_VELOCITY, fo.BC_PRESSURE_OUTLET, fo.BC_VELOCITY, fo.BC_VELOCITY, fo.BC_PERIODIC, fo.BC_PERIODIC]
    g = fo.Grid.uniform(n, [(0, Lx), (0, Ly), (0, Ly * n[2] / n[1])], bc, dt / rho)
    vg, Vg, pg = _fetch(H, ns, g)

    def velocity(b, t, X):
        if b == 0:
            return np.stack([4.0 * X[:, 1] * (Ly - X[:, 1]) / Ly ** 2, np.zeros(len(X)), np.zeros(len(X))])
        return np.zeros((3, len(X)))

    pressure = lambda b, t, X: np.array([pout(t, x) for x in X])
    so = fo.StepOracle(g, dt, rho, mu, velocity, krylov_rtol=1e-11, outer_rtol=1e-9, pressure=pressure)
    so.S_ksp = fo.KSP_BCGS
    vo, Vo, po = np.zeros(3 * g.ncell), [np.zeros(nf) for nf in g.nface], np.zeros(g.ncell)
    for _ in range(2):
        vo, Vo, po, info = so.step_once(vo, Vo, po)
    assert np.linalg.norm(vg - vo) <= 1e-6 * np.linalg.norm(vo)
    for d in range(2):
        assert np.linalg.norm(Vg[d] - Vo[d]) <= 1e-6 * np.linalg.norm(Vo[d])
    assert np.linalg.norm(pg - po) <= 1e-5 * np.linalg.norm(po)
    H.lib.NSDestroy(C.byref(ns))
    H.lib.MeshDestroy(C.byref(mesh))


@pytest.mark.gpu
def test_channel_flow_reaches_poiseuille(H):
    """Inlet / outlet channel (BASELINE config 3's geometry): started from rest, the flow settles on the parabolic profile
    with a linear pressure drop -- which the second-order scheme reproduces exactly."""
    from oracle import fluca_oracle as fo
    n, rho, mu, dt = (16, 12, 4), 1.0, 1.0, 0.05
    mesh, ns, keep, (Lx, Ly) = _channel(H, n, dt, 40, rho, mu, lambda t, x: 0.0, ("-ns_ksp_rtol", 1e-9, "-ns_abf_schur_ksp_rtol", 1e-10,
                                                                                  "-ns_abf_momentum_ksp_rtol", 1e-10))
    assert H.lib.NSSolve(ns) == 0
    g = fo.Grid.uniform(n, [(0, Lx), (0, Ly), (0, Ly * n[2] / n[1])], [1, 2, 1, 1, 3, 3], dt / rho)
    vg, Vg, pg = _fetch(H, ns, g)
    u = vg[:g.ncell].reshape(n[2], n[1], n[0])
    yc = (np.arange(n[1]) + 0.5) * Ly / n[1]
    xc = (np.arange(n[0]) + 0.5) * Lx / n[0]
    want = 4.0 * yc * (Ly - yc)
    assert np.abs(u - want[None, :, None]).max() < 2e-3
    assert np.abs(vg[g.ncell:]).max() < 2e-3
    # dp/dx = -mu u'' = -8 mu / Ly^2, p = 0 at the outlet
    pc = pg.reshape(n[2], n[1], n[0])
    assert np.abs(pc - (8.0 * mu / Ly ** 2 * (Lx - xc))[None, None, :]).max() < 0.05 * 8.0 * mu * Lx
    H.lib.NSDestroy(C.byref(ns))
    H.lib.MeshDestroy(C.byref(mesh))


@pytest.mark.gpu
def test_flow_config_driver_with_immersed_sphere(H):
    """examples/flow_configs.c -config sphere: the channel with the direct-forcing IBM active every step."""
    import re
    import subprocess
    from fluca_amd import build
    exe = build.build_example(name="flow_configs")
    run = lambda cfg: subprocess.run([exe, "-config", cfg, "-n", "64", "-ns_max_steps", "4", "-ns_ksp_type", "preonly", "-ns_abf_schur_pc_type", "mg"],
                                     capture_output=True, text=True, timeout=300)
    out = run("sphere")
    assert out.returncode == 0, out.stdout + out.stderr
    assert "markers 12868" in out.stdout and out.stdout.count("step ") == 4
    speed = float(re.search(r"rms fluid speed at the markers (\S+)", out.stdout).group(1))
    ke_s = float(re.search(r"mean kinetic energy (\S+)", out.stdout).group(1))
    ke_c = float(re.search(r"mean kinetic energy (\S+)", run("channel").stdout).group(1))
    assert 0.0 < speed < 0.67          # below the mean speed of the parabolic inflow: the forcing holds the fluid back
    assert ke_s != ke_c                # and the flow differs from the empty channel
    # BASELINE config 5's body: a cylinder of diameter 64 h along the periodic span, rings of markers one h apart
    cyl = subprocess.run([exe, "-config", "cylinder", "-n", "128", "-ns_max_steps", "2", "-ns_ksp_type", "preonly", "-ns_abf_schur_pc_type", "mg"],
                         capture_output=True, text=True, timeout=300)
    assert cyl.returncode == 0, cyl.stdout + cyl.stderr
    assert f"markers {201 * 128}" in cyl.stdout and cyl.stdout.count("step ") == 2
    assert 0.0 < float(re.search(r"rms fluid speed at the markers (\S+)", cyl.stdout).group(1)) < 0.67
