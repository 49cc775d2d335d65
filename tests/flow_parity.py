"""Velocity-field parity of whole time steps (BASELINE north_star: "residual AND velocity field"): the channel with an immersed sphere --
config 4's set-up: parabolic VELOCITY inlet, PRESSURE_OUTLET p = 0, no-slip walls in y, periodic span, a sphere of markers held at rest by
direct forcing -- stepped by the C host mirror on the GPU (NSStep_CNLinear: cnlinearcart3d.c:2807-2863 restated in fluca_amd/host/fluca_host.c)
and by the CPU oracle's composition of the same reference formulas (oracle.fluca_oracle.StepOracle), then compared field by field.

Shared by tests/test_gpu_flow_parity.py and bench.py's configs.flow_step.parity (the checker beside the measurement, never inside it)."""
import ctypes as C
import time

import numpy as np


def sphere_markers(n, diameter_cells):
    """Fibonacci lattice on a sphere at the centre of the unit cube, one marker per h^2 of surface, marker volume h^3 (examples/flow_configs.c)"""
    h = 1.0 / n
    R = 0.5 * diameter_cells * h
    L = int(round(4 * np.pi * R * R / (h * h)))
    i = np.arange(L) + 0.5
    z = 1.0 - 2.0 * i / L
    r = np.sqrt(1.0 - z * z)
    th = np.pi * (3.0 - np.sqrt(5.0)) * np.arange(L)
    return [0.5 + R * r * np.cos(th), 0.5 + R * r * np.sin(th), 0.5 + R * z], np.full(L, h ** 3)


def channel_sphere(n=128, nsteps=2, Re=100.0, diameter_cells=None, rtol_outer=1e-9, rtol_inner=1e-11, oracle_inner_rtol=1e-4, extra_opts=()):
    """-> dict of relative differences GPU vs oracle after `nsteps` steps on an n^3 grid, plus the continuity check of SURVEY 8(d)."""
    from fluca_amd import capi, hostapi as H
    from oracle import fluca_oracle as fo
    P = C.c_void_p
    rho, mu, dt = 1.0, 1.0 / Re, 0.5 / n
    D = diameter_cells if diameter_cells is not None else n // 8
    X, dV = sphere_markers(n, D)
    L = X[0].size
    t_start = time.perf_counter()

    # ---- GPU: the C host mirror -----------------------------------------------------------------------------------------------------
    mesh = P()
    assert H.lib.MeshCartCreate3d(0, 0, 1, n, n, n, -1, -1, -1, None, None, None, C.byref(mesh)) == 0
    assert H.lib.MeshSetUp(mesh) == 0
    assert H.lib.MeshCartSetUniformCoordinates(mesh, 0., 1., 0., 1., 0., 1.) == 0
    ns = P()
    assert H.lib.NSCreate(C.byref(ns)) == 0 and H.lib.NSSetType(ns, b"cnlinear") == 0 and H.lib.NSSetMesh(ns, mesh) == 0
    assert H.lib.NSSetDensity(ns, rho) == 0 and H.lib.NSSetViscosity(ns, mu) == 0

    @H.BCFunc
    def inlet(dim, t, x, val, ctx):
        val[0], val[1], val[2] = 4.0 * x[1] * (1.0 - x[1]), 0.0, 0.0
        return 0

    @H.BCFunc
    def wall(dim, t, x, val, ctx):
        val[0] = val[1] = val[2] = 0.0
        return 0

    @H.BCFunc
    def outlet(dim, t, x, val, ctx):
        val[0] = 0.0
        return 0

    bcs = [H.NSBoundaryCondition(type=H.NS_BC_VELOCITY, velocity=inlet), H.NSBoundaryCondition(type=H.NS_BC_PRESSURE_OUTLET, pressure=outlet),
           H.NSBoundaryCondition(type=H.NS_BC_VELOCITY, velocity=wall), H.NSBoundaryCondition(type=H.NS_BC_VELOCITY, velocity=wall),
           H.NSBoundaryCondition(type=H.NS_BC_PERIODIC), H.NSBoundaryCondition(type=H.NS_BC_PERIODIC)]
    for b in range(6):
        assert H.lib.NSSetBoundaryCondition(ns, b, bcs[b]) == 0
    argc, av = H.argv("-ns_time_step_size", dt, "-ns_max_steps", nsteps, "-ns_ksp_rtol", rtol_outer, "-ns_abf_schur_ksp_rtol", rtol_inner,
                      "-ns_abf_momentum_ksp_rtol", rtol_inner, "-ns_abf_schur_ksp_max_it", 50000, *extra_opts)
    assert H.lib.NSSetFromOptions(ns, argc, av) == 0 and H.lib.NSSetUp(ns) == 0
    keep = []
    for a in X + [dV]:
        dptr = P()
        capi.check(capi.lib.fl_malloc(0, a.size * 8, C.byref(dptr)))
        capi.check(capi.lib.fl_memcpy_h2d(0, dptr, np.ascontiguousarray(a).ctypes.data_as(C.c_void_p), a.size * 8))
        keep.append(dptr)
    assert H.lib.NSSetImmersedBoundary(ns, 0, L, keep[0], keep[1], keep[2], keep[3], None) == 0
    t0 = time.perf_counter()
    assert H.lib.NSSolve(ns) == 0
    t_gpu = time.perf_counter() - t0
    bc = [fo.BC_VELOCITY, fo.BC_PRESSURE_OUTLET, fo.BC_VELOCITY, fo.BC_VELOCITY, fo.BC_PERIODIC, fo.BC_PERIODIC]
    g = fo.Grid.uniform((n, n, n), [(0, 1), (0, 1), (0, 1)], bc, dt / rho)
    v, p, Vp = P(), P(), (C.c_void_p * 3)()
    assert H.lib.NSGetSolutionArrays(ns, C.byref(v), Vp, C.byref(p)) == 0

    def get(ptr, m):
        out = np.empty(m)
        capi.check(capi.lib.fl_memcpy_d2h(0, out.ctypes.data_as(C.c_void_p), ptr, m * 8))
        return out

    vg, pg = get(v, 3 * g.ncell), get(p, g.ncell)
    Vg = [get(C.c_void_p(Vp[d]), g.nface[d]) for d in range(3)]
    its = (C.c_int(), C.c_int(), C.c_double(), C.c_int())
    assert H.lib.NSGetLinearSolveInfo(ns, C.byref(its[0]), C.byref(its[2]), C.byref(its[3])) == 0
    H.lib.NSDestroy(C.byref(ns))
    H.lib.MeshDestroy(C.byref(mesh))
    for dptr in keep:
        capi.lib.fl_free(0, dptr)

    # ---- CPU: the oracle's step -----------------------------------------------------------------------------------------------------
    def velocity(b, t, Xf):
        if b == 0:
            return np.stack([4.0 * Xf[:, 1] * (1.0 - Xf[:, 1]), np.zeros(len(Xf)), np.zeros(len(Xf))])
        return np.zeros((3, len(Xf)))

    t0 = time.perf_counter()
    # the oracle's PCApply_ABF is the PRECONDITIONER of a Richardson iteration on the block system: what the converged step is depends on
    # outer_rtol alone, so its inner solves run to oracle_inner_rtol (same answer to 2e-12 as with 1e-11 at 64^3, in 60 % of the time)
    so = fo.StepOracle(g, dt, rho, mu, velocity, krylov_rtol=oracle_inner_rtol, outer_rtol=rtol_outer, pressure=lambda b, t, Xf: np.zeros(len(Xf)),
                       ibm=dict(kind=0, X=X, dV=dV))
    vo, Vo, po = np.zeros(3 * g.ncell), [np.zeros(nf) for nf in g.nface], np.zeros(g.ncell)
    outer = []
    for _ in range(nsteps):
        vo, Vo, po, info = so.step_once(vo, Vo, po)
        outer.append(info["outer_its"])
    t_cpu = time.perf_counter() - t0
    rel = lambda a, b: float(np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-300))
    # what the forcing does to the flow: the same steps without the body, on the oracle (a parity of two zero fields would prove nothing)
    U_at_markers = g.ibm_interp(0, X, vg.reshape(3, -1))
    div = np.abs(g.rhs(*Vg)).max()      # || D V ||_inf of the GPU's face velocity, D applied by the oracle
    return {"cells_per_axis": n, "steps": nsteps, "markers": int(L), "sphere_diameter_cells": D, "dt": dt, "Re": Re,
            "rtol_outer": rtol_outer, "rtol_inner_gpu": rtol_inner, "rtol_inner_oracle": oracle_inner_rtol,
            "rel_l2_diff_v": rel(vg, vo), "rel_l2_diff_V": [rel(Vg[d], Vo[d]) for d in range(3)], "rel_l2_diff_p": rel(pg, po),
            "rel_max_diff_v": float(np.abs(vg - vo).max() / np.abs(vo).max()),
            "max_abs_v": float(np.abs(vo).max()), "rms_speed_at_markers_over_inflow_peak": float(np.sqrt((U_at_markers ** 2).sum(axis=0).mean())),
            "div_inf": float(div), "b_inf_of_the_last_step": so.b_inf, "div_bound_10_rtol_b_inf": 10.0 * rtol_outer * so.b_inf,
            "outer_its_gpu_last_step": its[0].value, "outer_its_oracle": outer,
            "seconds_gpu": t_gpu, "seconds_oracle": t_cpu, "seconds_total": time.perf_counter() - t_start,
            "oracle": "StepOracle (oracle/fluca_oracle.py): assembled A per step + KSPBCGS, assembled S + KSPCG, Richardson on the block system with PCApply_ABF, IBM direct forcing"}
