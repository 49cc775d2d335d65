"""-m gpu: the matrix-free momentum block (fl_momentum_*) through the C-ABI against the oracle's assembled CSR
(literal restatement of cnlinearcart3d.c:425-632, 873-1294, 2930-2941)."""
import numpy as np
import pytest
import torch

from oracle import fluca_oracle as fo
from tests.gpu_common import CAVITY, CAVITY_BOX, O, PER, SYM, V, dev, host, stretched

pytestmark = pytest.mark.gpu

CASES = [
    ((17, 9, 11), CAVITY, False),
    ((12, 10, 9), [PER] * 6, False),
    ((9, 12, 7), [V, O, V, V, PER, PER], True),
    ((11, 7, 13), [O, V, SYM, V, V, O], True),
    ((8, 6, 5), [SYM, SYM, O, O, SYM, SYM], True),
    ((130, 37, 20), CAVITY, True),          # more than one tile in x, ragged in x and y
    ((70, 5, 3), [PER, PER, V, V, O, SYM], False),
]


def _pair(n, bc, nonuni):
    from fluca_amd.poisson import Momentum, Poisson
    box = CAVITY_BOX
    if nonuni:
        xf = [stretched(n[d], box[d][0], box[d][1], 1.1 + 0.2 * d) for d in range(3)]
        P, g = Poisson(n, xf, bc, 1e-3), fo.Grid(n, xf, bc, 1e-3)
    else:
        P, g = Poisson.uniform(n, box, bc, 1e-3), fo.Grid.uniform(n, box, bc, 1e-3)
    return P, Momentum(P), g


def _fields(g, seed=3):
    rng = np.random.default_rng(seed)
    V0 = [rng.standard_normal(g.nface[d]) for d in range(3)]
    W = [rng.standard_normal(g.nface[d]) for c in range(3) for d in range(3)]
    return V0, W


def _close(got, want, tol=2e-13):
    scale = np.abs(want).max()
    assert np.abs(got - want).max() <= tol * scale, (np.abs(got - want).max(), scale)


@pytest.mark.parametrize("n,bc,nonuni", CASES)
def test_laplacian_convection_and_A_match_assembled_rows(n, bc, nonuni):
    P, M, g = _pair(n, bc, nonuni)
    V0, W = _fields(g)
    dt, rho, mu = 0.013, 1.7, 0.031
    v = np.random.default_rng(11).standard_normal(3 * g.ncell)
    vd = dev(v)
    M.set_state(dt, rho, mu, [dev(a) for a in V0], [dev(a) for a in W])
    A = g.assemble_momentum(1.0, dt, -0.5 * mu * dt / rho, V0, W)
    _close(host(M.apply(vd)), A.mult(v))
    _close(host(M.diagonal()), A.diag())
    # the two operators on their own
    M.set_coefficients(0.0, 0.0, 1.0)
    _close(host(M.apply(vd)), g.assemble_momentum(0.0, 0.0, 1.0).mult(v))
    M.set_coefficients(0.0, 1.0, 0.0)
    Cm = g.assemble_momentum(0.0, 1.0, 0.0, V0, W)
    _close(host(M.apply(vd)), Cm.mult(v))
    _close(host(M.diagonal()), Cm.diag())
    M.close()
    P.close()


def test_state_can_be_replaced():
    """NSFormJacobian re-forms A every step from the new sol0 (cnlinearcart3d.c:2930-2941)."""
    P, M, g = _pair((20, 9, 6), CAVITY, True)
    v = np.random.default_rng(1).standard_normal(3 * g.ncell)
    for seed in (1, 2):
        V0, W = _fields(g, seed)
        M.set_state(0.01, 1.0, 0.01, [dev(a) for a in V0], [dev(a) for a in W])
        _close(host(M.apply(dev(v))), g.assemble_momentum(1.0, 0.01, -0.5 * 0.01 * 0.01, V0, W).mult(v))
    M.close()
    P.close()


@pytest.mark.parametrize("n,bc,nonuni", [CASES[0], CASES[2], CASES[3], CASES[5]])
@pytest.mark.parametrize("pc", [fo.PC_JACOBI, fo.PC_NONE])
def test_momentum_bcgs_matches_oracle(n, bc, nonuni, pc):
    P, M, g = _pair(n, bc, nonuni)
    V0, W = _fields(g)
    # CFL ~ 1 convection and a stiff viscous part so that the solve takes a handful of iterations
    hmin = min(np.diff(g.xf[d]).min() for d in range(3))
    dt, rho, mu = 0.5 * hmin, 1.0, 0.5 * hmin
    M.set_state(dt, rho, mu, [dev(a) for a in V0], [dev(a) for a in W])
    A = g.assemble_momentum(1.0, dt, -0.5 * mu * dt / rho, V0, W)
    b = np.random.default_rng(7).standard_normal(3 * g.ncell)
    rtol = 1e-8
    xo, io = A.solve(b, ksp=fo.KSP_BCGS, pc=pc, nullspace=False, rtol=rtol, maxit=500)
    xg, ig = M.solve(dev(b), history=True, pc=pc, rtol=rtol, maxit=500)
    assert io["reason"] > 0 and ig["reason"] == io["reason"]
    assert io["iters"] >= 3
    m = min(len(ig["history"]), len(io["history"]))
    assert np.allclose(ig["history"][:3], io["history"][:3], rtol=1e-9)
    assert np.allclose(ig["history"][:min(m, 8)], io["history"][:min(m, 8)], rtol=1e-5)
    assert abs(ig["iters"] - io["iters"]) <= max(2, io["iters"] // 6)
    xg = host(xg)
    assert np.linalg.norm(b - A.mult(xg)) <= 50 * rtol * np.linalg.norm(b)
    assert np.linalg.norm(xg - xo) <= 1e-5 * np.linalg.norm(xo)
    M.close()
    P.close()


def test_argument_checks():
    from fluca_amd import capi
    from fluca_amd.poisson import Momentum, Poisson
    P = Poisson.uniform((8, 8, 1), CAVITY_BOX, [V] * 6, 1e-3)
    with pytest.raises(RuntimeError):
        Momentum(P)                              # one cell along z: the wall rows of the reference do not exist
    P.close()
    P = Poisson.uniform((8, 8, 8), CAVITY_BOX, [V] * 6, 1e-3)
    M = Momentum(P)
    b = P.empty(3 * P.ncell).zero_()
    with pytest.raises(RuntimeError):
        M.solve(b, type=capi.KSP_CG)             # A is not symmetric
    x, info = M.solve(b)                          # before set_state: A = I
    assert info["reason"] == 3 and info["iters"] == 0       # zero rhs: CONVERGED_ATOL at iteration 0 (KSPConvergedDefault)
    # ... and the answer is the zero guess even if an earlier solve left something in the work vector (no update kernel ran to write it)
    rhs = torch.ones_like(b)
    y, _ = M.solve(rhs)
    assert float((y - rhs).abs().max()) == 0.0    # A = I
    x, info = M.solve(b)
    assert info["iters"] == 0 and float(x.abs().max()) == 0.0
    # the entry points of the state with v0 refuse missing arrays
    import ctypes as C
    three, nine = (C.c_void_p * 3)(), (C.c_void_p * 9)()
    assert capi.lib.fl_momentum_set_state_v0(M.h, 0.1, 1.0, 0.1, three, nine, None) == -85  # FL_ERR_ARG_NULL
    assert capi.lib.fl_momentum_interp_faces_ends(M.h, None, None, nine) == -85  # FL_ERR_ARG_NULL
    M.close()
    P.close()


@pytest.mark.parametrize("n,bc,nonuni", [CASES[0], CASES[1], CASES[2], CASES[3], CASES[5]])
def test_face_normal_interpolation_matches_oracle(n, bc, nonuni):
    P, M, g = _pair(n, bc, nonuni)
    rng = np.random.default_rng(21)
    v = rng.standard_normal(3 * g.ncell)
    rhs = [rng.standard_normal(g.nface[d]) for d in range(3)]
    want = g.apply_T(v, rhs)
    got = M.face_interp(dev(v), [dev(r) for r in rhs])
    for d in range(3):
        _close(host(got[d]), want[d])
    want0 = g.apply_T(v)
    got0 = M.face_interp(dev(v))
    for d in range(3):
        assert np.abs(host(got0[d]) - want0[d]).max() <= 2e-13 * np.abs(v).max()
    M.close()
    P.close()


@pytest.mark.parametrize("n,bc,nonuni", [((17, 9, 11), CAVITY, False), ((12, 10, 9), [PER] * 6, False), ((20, 12, 9), [V, O, V, V, PER, PER], False)])
def test_pcapply_abf_matches_composed_oracle(n, bc, nonuni):
    """fl_abf_apply == PCApply_ABF (abfpc.c:48-111) composed from the oracle's operators and Krylov restatements."""
    P, M, g = _pair(n, bc, nonuni)
    rng = np.random.default_rng(31)
    V0, W = _fields(g)
    hmin = min(np.diff(g.xf[d]).min() for d in range(3))
    dt, rho, mu = 0.5 * hmin, 1.0, 0.5 * hmin
    # kappa of the pair is 1e-3: use a consistent dt/rho for the momentum block
    kap = g.kappa
    dt = kap * rho
    M.set_state(dt, rho, mu, [dev(a) for a in V0], [dev(a) for a in W])
    A = g.assemble_momentum(1.0, dt, -0.5 * mu * dt / rho, V0, W)
    S = g.assemble_S()
    momrhs = rng.standard_normal(3 * g.ncell)
    interprhs = [1e-2 * rng.standard_normal(g.nface[d]) for d in range(3)]
    nullspace = O not in bc
    if nullspace:
        # no flux through the walls: a singular S needs a right-hand side without a constant component (the time stepper
        # provides that through the boundary-condition vectors)
        fshape = [(n[2], n[1], g.nf[0]), (n[2], g.nf[1], n[0]), (g.nf[2], n[1], n[0])]
        for d in range(3):
            if not g.periodic[d]:
                a = interprhs[d].reshape(fshape[d])
                sl = [slice(None)] * 3
                sl[2 - d] = [0, -1]
                a[tuple(sl)] = 0.0
    # oracle composition
    vs, i0 = A.solve(momrhs, ksp=fo.KSP_BCGS, pc=fo.PC_JACOBI, nullspace=False, rtol=1e-10, maxit=500)
    Vs = g.apply_T(vs, interprhs)
    srhs = g.rhs(*Vs)
    po, i1 = S.solve(srhs, ksp=fo.KSP_CG, pc=fo.PC_JACOBI, nullspace=nullspace, rtol=1e-10, maxit=5000)
    Gp = g.apply_G(po)
    Gst = g.apply_gst(po)
    v_ref = vs - np.concatenate(Gp)
    V_ref = [Vs[d] - Gst[d] for d in range(3)]
    from fluca_amd.poisson import KspOptions
    from fluca_amd import capi
    v, Vf, p, info = M.abf_apply(dev(momrhs), [dev(r) for r in interprhs], None,
                                 momentum=KspOptions(type=capi.KSP_BCGS, rtol=1e-10, maxit=500),
                                 schur=KspOptions(rtol=1e-10, maxit=5000, remove_nullspace=int(nullspace)))
    assert info[0]["reason"] == i0["reason"] and info[1]["reason"] == i1["reason"]
    assert abs(info[0]["iters"] - i0["iters"]) <= 2
    assert abs(info[1]["iters"] - i1["iters"]) <= max(3, i1["iters"] // 20)
    pg = host(p)
    if nullspace:
        pg, po = pg - pg.mean(), po - po.mean()
    assert np.linalg.norm(pg - po) <= 1e-6 * np.linalg.norm(po)
    assert np.linalg.norm(host(v) - v_ref) <= 1e-7 * np.linalg.norm(v_ref)
    for d in range(3):
        assert np.linalg.norm(host(Vf[d]) - V_ref[d]) <= 1e-7 * max(np.linalg.norm(V_ref[d]), 1e-30)
    # the projected face velocity is discretely divergence-free up to the Schur tolerance: D V = contrhs (= 0)
    div = g.rhs(*[host(Vf[d]) for d in range(3)])
    assert np.linalg.norm(div) <= 1e-7 * np.linalg.norm(srhs)
    M.close()
    P.close()


def _state(M, g, mu=0.05):
    """a momentum state with a noticeably non-unit diagonal: dt of the pair's kappa, V0 / v0interp scaled up"""
    V0, W = _fields(g)
    V0, W = [8.0 * a for a in V0], [8.0 * a for a in W]
    rho = 1.0
    dt = g.kappa * rho
    M.set_state(dt, rho, mu, [dev(a) for a in V0], [dev(a) for a in W])
    return g.assemble_momentum(1.0, dt, -0.5 * mu * dt / rho, V0, W)


@pytest.mark.parametrize("n,bc,nonuni", [CASES[0], CASES[1], CASES[2], CASES[3], CASES[5], CASES[6], ((66, 4, 2), [PER] * 6, False)])   # ([1], [6], the last: the seam two cells deep on short axes)
@pytest.mark.parametrize("kind", [fo.AINV_DIAG, fo.AINV_ROWSUM])
def test_schur_complement_of_the_diag_and_rowsum_types(n, bc, nonuni, kind):
    """fl_abf_schur_apply == S = D ((-T) a^-1 kappa G - (-R)) of PCSetUp_ABF (abfpc.c:151-171), a = diag(A) or A 1."""
    P, M, g = _pair(n, bc, nonuni)
    A = _state(M, g)
    _close(host(M.rowsum()), A.mult(np.ones(3 * g.ncell)))                       # MatGetRowSum
    ainv = fo.abf_ainv(A, kind)
    assert np.abs(1.0 / ainv - 1.0).max() > 1e-2                                 # the variant is not the ID operator in disguise
    p = np.random.default_rng(17).standard_normal(g.ncell)
    M.set_ainv_types(schur=kind)
    want = fo.abf_schur_apply(g, ainv, p)
    # the oracle forms -T(a^-1 kGp) + T(kGp) term by term like MatMatMult + MatAXPY; the device scales by (a^-1 - 1) once: the two
    # differ by the cancellation in the former, relative to the size of the cancelling terms (larger than the result on stretched grids)
    got = host(M.schur_apply(dev(p)))                                            # one pass (fl_schur_var.hip; round 5)
    _close(got, want, tol=2e-10)
    from fluca_amd import capi
    capi.check(capi.lib.fl_tuning_set(b"schur_var_fused", 0))                    # ... and the composition of the seven kernels it replaces
    try:
        comp = host(M.schur_apply(dev(p)))
    finally:
        capi.check(capi.lib.fl_tuning_set(b"schur_var_fused", 1))
    _close(comp, want, tol=2e-10)
    _close(got, comp, tol=1e-12)
    M.set_ainv_types(schur=fo.AINV_ID)                                           # ID: the 7-point operator of the hot path
    _close(host(M.schur_apply(dev(p))), g.assemble_S().mult(p), tol=2e-13)
    _close(fo.abf_schur_apply(g, None, p), g.assemble_S().mult(p), tol=1e-9)     # and the oracle's composition cancels to it (abfpc.c:152-154,169)
    M.close()
    P.close()


@pytest.mark.parametrize("n,bc,nonuni", [((9, 8, 7), CAVITY, True), CASES[2], CASES[4]])
@pytest.mark.parametrize("schur,upper", [(fo.AINV_DIAG, fo.AINV_DIAG), (fo.AINV_ROWSUM, fo.AINV_ID), (fo.AINV_ID, fo.AINV_ROWSUM)])
def test_pcapply_abf_with_diag_and_rowsum_types(n, bc, nonuni, schur, upper):
    """PCApply_ABF with -pc_abf_schur_ainv_type / -pc_abf_upper_ainv_type != ID (abfpc.c:81-101, 151-171) against the oracle's
    composition with the assembled (dense) S of that type."""
    from fluca_amd import capi
    from fluca_amd.poisson import KspOptions
    P, M, g = _pair(n, bc, nonuni)
    A = _state(M, g)
    rng = np.random.default_rng(5)
    momrhs = rng.standard_normal(3 * g.ncell)
    interprhs = [1e-2 * rng.standard_normal(g.nface[d]) for d in range(3)]
    nullspace = O not in bc
    if nullspace:
        fshape = [(n[2], n[1], g.nf[0]), (n[2], g.nf[1], n[0]), (g.nf[2], n[1], n[0])]
        for d in range(3):
            if not g.periodic[d]:
                a = interprhs[d].reshape(fshape[d])
                sl = [slice(None)] * 3
                sl[2 - d] = [0, -1]
                a[tuple(sl)] = 0.0
    vs, _ = A.solve(momrhs, ksp=fo.KSP_BCGS, pc=fo.PC_JACOBI, nullspace=False, rtol=1e-12, maxit=500)
    Vs = g.apply_T(vs, interprhs)
    srhs = g.rhs(*Vs)
    if schur == fo.AINV_ID:
        po, _ = g.assemble_S().solve(srhs, ksp=fo.KSP_BCGS, pc=fo.PC_JACOBI, nullspace=nullspace, rtol=1e-12, maxit=5000)
    else:
        Sd = fo.abf_schur_dense(g, fo.abf_ainv(A, schur))
        po = np.linalg.lstsq(Sd, srhs, rcond=1e-12)[0] if nullspace else np.linalg.solve(Sd, srhs)
        assert np.linalg.norm(Sd @ po - srhs) <= 1e-9 * np.linalg.norm(srhs)       # the right-hand side is consistent
    kGp = np.concatenate(g.apply_G(po))
    w = kGp if upper == fo.AINV_ID else fo.abf_ainv(A, upper) * kGp               # :80-94
    Tw, TkGp, kGst = g.apply_T(w), g.apply_T(kGp), g.apply_gst(po)
    v_ref = vs - w                                                                 # :95
    V_ref = [Vs[d] - Tw[d] + TkGp[d] - kGst[d] for d in range(3)]                 # :96-101
    M.set_ainv_types(schur=schur, upper=upper)
    v, Vf, p, info = M.abf_apply(dev(momrhs), [dev(r) for r in interprhs], None,
                                 momentum=KspOptions(type=capi.KSP_BCGS, rtol=1e-12, maxit=500),
                                 schur=KspOptions(type=capi.KSP_BCGS, rtol=1e-11, maxit=5000, remove_nullspace=int(nullspace)))
    assert info[0]["reason"] > 0 and info[1]["reason"] > 0, info
    if schur != fo.AINV_ID:
        assert info[1]["iters"] <= 40, info[1]       # the constant-coefficient solve is a good preconditioner for the variable one
    pg = host(p)
    if nullspace:
        pg, po = pg - pg.mean(), po - po.mean()
    assert np.linalg.norm(pg - po) <= 1e-7 * np.linalg.norm(po)
    assert np.linalg.norm(host(v) - v_ref) <= 1e-7 * np.linalg.norm(v_ref)
    for d in range(3):
        assert np.linalg.norm(host(Vf[d]) - V_ref[d]) <= 1e-7 * max(np.linalg.norm(V_ref[d]), 1e-30)
    M.close()
    P.close()


@pytest.mark.parametrize("n,bc,nonuni", [CASES[0], CASES[1], CASES[3]])
def test_block_jacobian_mult(n, bc, nonuni):
    """fl_abf_jacobian_mult == the MatNest product of cnlinearcart3d.c:2885-2941 composed from the oracle's operators."""
    P, M, g = _pair(n, bc, nonuni)
    rng = np.random.default_rng(41)
    V0, W = _fields(g)
    rho, mu = 1.0, 0.02
    dt = g.kappa * rho
    M.set_state(dt, rho, mu, [dev(a) for a in V0], [dev(a) for a in W])
    A = g.assemble_momentum(1.0, dt, -0.5 * mu * dt / rho, V0, W)
    v = rng.standard_normal(3 * g.ncell)
    Vf = [rng.standard_normal(g.nface[d]) for d in range(3)]
    p = rng.standard_normal(g.ncell)
    kGp = np.concatenate(g.apply_G(p))
    w = v + kGp
    Tw = g.apply_T(w)
    kGst = g.apply_gst(p)
    fv_ref = A.mult(v) + kGp
    fV_ref = [Vf[d] - Tw[d] + kGst[d] for d in range(3)]
    fp_ref = -g.rhs(*Vf)
    fv, fV, fp = M.jacobian_mult(dev(v), [dev(a) for a in Vf], dev(p))
    _close(host(fv), fv_ref)
    _close(host(fp), fp_ref, 1e-12)
    for d in range(3):
        _close(host(fV[d]), fV_ref[d], 1e-12)
    M.close()
    P.close()


def test_abf_is_an_approximate_inverse_of_the_block_jacobian():
    """Everything at once, signs and scalings included: with A = I + O(dt) the approximate block factorisation with
    Ainv = I is an O(dt) perturbation of J^-1, so J (PCApply_ABF f) must come back to f up to a few per cent."""
    from fluca_amd import capi
    from fluca_amd.poisson import KspOptions
    n, bc = (17, 9, 11), CAVITY
    P, M, g = _pair(n, bc, False)
    rng = np.random.default_rng(51)
    V0, W = _fields(g)
    rho, mu = 1.0, 0.02
    dt = g.kappa * rho
    M.set_state(dt, rho, mu, [dev(a) for a in V0], [dev(a) for a in W])
    momrhs = rng.standard_normal(3 * g.ncell)
    v, Vf, p, info = M.abf_apply(dev(momrhs), momentum=KspOptions(type=capi.KSP_BCGS, rtol=1e-12, maxit=500), schur=KspOptions(rtol=1e-12, maxit=5000))
    assert info[0]["reason"] > 0 and info[1]["reason"] > 0
    fv, fV, fp = M.jacobian_mult(v, Vf, p)
    r = np.linalg.norm(host(fv) - momrhs) / np.linalg.norm(momrhs)
    assert r < 0.05, r
    # interprhs = contrhs = 0: the face row and the continuity row are reproduced to solver accuracy
    assert max(np.abs(host(fV[d])).max() for d in range(3)) <= 1e-8 * np.abs(host(Vf[0])).max()
    assert np.abs(host(fp)).max() <= 1e-7 * np.abs(host(Vf[0])).max() / min(np.diff(g.xf[0]).min(), np.diff(g.xf[2]).min())
    M.close()
    P.close()


@pytest.mark.parametrize("n,bc,nonuni", [CASES[0], CASES[1], CASES[3], CASES[4], CASES[5]])
def test_face_velocity_interpolation_B_matches_oracle(n, bc, nonuni):
    P, M, g = _pair(n, bc, nonuni)
    rng = np.random.default_rng(61)
    v = rng.standard_normal(3 * g.ncell)
    want = g.apply_B(v)
    got = M.interp_faces(dev(v))
    for q in range(9):
        assert np.abs(host(got[q]) - want[q]).max() <= 2e-13 * np.abs(v).max(), q
    vbc = [rng.standard_normal(g.nface[d]) for c in range(3) for d in range(3)]
    got = M.interp_faces(dev(v), [dev(a) for a in vbc])
    for q in range(9):
        assert np.abs(host(got[q]) - (want[q] + vbc[q])).max() <= 2e-13 * max(np.abs(v).max(), np.abs(vbc[q]).max()), q
    M.close()
    P.close()


def _boundary_vbc(g, rng):
    """vbc of cnlinearcart3d.c:1749-1932: values on the faces at the ends of a non-periodic axis, zero on every inner face"""
    out = []
    for c in range(3):
        for d in range(3):
            shape = [g.n[2], g.n[1], g.n[0]]
            shape[2 - d] = g.nf[d]
            a = np.zeros(shape)
            if not g.periodic[d]:
                ends = [slice(None)] * 3
                for f in (0, g.n[d]):
                    ends[2 - d] = f
                    a[tuple(ends)] = rng.standard_normal(a[tuple(ends)].shape)
            out.append(a.ravel())
    return out


@pytest.mark.parametrize("n,bc,nonuni", CASES + [((256, 21, 6), [O, V, V, SYM, PER, PER], True), ((129, 16, 9), [PER] * 6, False), ((3, 9, 4), CAVITY, False)])
def test_state_with_v0_forms_the_same_operator(n, bc, nonuni):
    """fl_momentum_set_state_v0 (v0interp on inner faces formed inside k_mom3 from v0) against the oracle's rows built from the stored
    v0interp = B v0 + vbc, against the stored-path kernel, and through a BiCGStab solve"""
    P, M, g = _pair(n, bc, nonuni)
    rng = np.random.default_rng(23)
    V0 = [rng.standard_normal(g.nface[d]) for d in range(3)]
    v0 = rng.standard_normal(3 * g.ncell)
    vbc = _boundary_vbc(g, rng)
    W = [b + c for b, c in zip(g.apply_B(v0), vbc)]
    hmin = min(np.diff(g.xf[d]).min() for d in range(3))
    dt, rho, mu = 0.5 * hmin, 1.3, 0.5 * hmin
    v = rng.standard_normal(3 * g.ncell)
    A = g.assemble_momentum(1.0, dt, -0.5 * mu * dt / rho, V0, W)
    Wd = M.interp_faces(dev(v0), [dev(a) for a in vbc])
    for q in range(9):
        assert np.abs(host(Wd[q]) - W[q]).max() <= 2e-13 * max(1.0, np.abs(W[q]).max())
    M.set_state(dt, rho, mu, [dev(a) for a in V0], Wd)
    y_stored, d_stored = host(M.apply(dev(v))), host(M.diagonal())
    M.set_state(dt, rho, mu, [dev(a) for a in V0], Wd, v0=dev(v0))
    y_fly, d_fly = host(M.apply(dev(v))), host(M.diagonal())
    _close(y_fly, A.mult(v))
    _close(d_fly, A.diag())
    # same table numbers, same two products per face value in the same order: the two kernels agree to the last rounding of the sums
    assert np.abs(y_fly - y_stored).max() <= 4e-15 * np.abs(y_stored).max()
    assert np.abs(d_fly - d_stored).max() <= 4e-15 * np.abs(d_stored).max()
    b = rng.standard_normal(3 * g.ncell)
    for pc in (fo.PC_JACOBI, fo.PC_NONE):
        xo, io = A.solve(b, ksp=fo.KSP_BCGS, pc=pc, nullspace=False, rtol=1e-8, maxit=500)
        xg, ig = M.solve(dev(b), history=True, pc=pc, rtol=1e-8, maxit=500)
        assert io["reason"] > 0 and ig["reason"] == io["reason"]
        assert np.allclose(ig["history"][:3], io["history"][:3], rtol=1e-9)
        assert abs(ig["iters"] - io["iters"]) <= max(2, io["iters"] // 6)
        assert np.linalg.norm(host(xg) - xo) <= 1e-5 * np.linalg.norm(xo)
    # the stored fields need only be right on the block-end faces then (fl_momentum_interp_faces_ends leaves the inner entries alone)
    junk = [torch.full((g.nface[d],), 1e300, dtype=torch.float64, device="cuda") for c in range(3) for d in range(3)]
    We = M.interp_faces(dev(v0), [dev(a) for a in vbc], ends_only=True, out=junk)
    M.set_state(dt, rho, mu, [dev(a) for a in V0], We, v0=dev(v0))
    _close(host(M.apply(dev(v))), A.mult(v))
    _close(host(M.diagonal()), A.diag())
    # a state without v0 afterwards goes back to the stored fields
    Wr = [rng.standard_normal(g.nface[d]) for c in range(3) for d in range(3)]
    M.set_state(dt, rho, mu, [dev(a) for a in V0], [dev(a) for a in Wr])
    _close(host(M.apply(dev(v))), g.assemble_momentum(1.0, dt, -0.5 * mu * dt / rho, V0, Wr).mult(v))
    M.close()
    P.close()


def test_momentum_and_multigrid_full_size_properties_512():
    """BASELINE size (512^3: 134 M cells, 403 M velocity unknowns): size-independent properties of the widened rows.
    Linearity of A, A with zero advecting fields and zero viscosity is the identity, diag(A) = A e summed the cheap way on
    a constant-coefficient state, BiCGStab solve-then-apply round trip, PCApply_ABF leaves a divergence-free face
    velocity, multigrid-PCG and Jacobi-PCG agree on the pressure."""
    import torch
    from fluca_amd import capi
    from fluca_amd.poisson import KspOptions, Momentum, Poisson
    n = (512, 512, 512)
    P = Poisson.uniform(n, [(0, 1)] * 3, CAVITY, 1.0 / 512 / 2)
    M = Momentum(P)
    N = P.ncell
    gen = torch.Generator(device="cuda").manual_seed(9)
    rnd = lambda m: torch.rand(m, dtype=torch.float64, device="cuda", generator=gen) * 2 - 1
    x, y = rnd(3 * N), rnd(3 * N)
    zeroF = [torch.zeros(P.nface[d], dtype=torch.float64, device="cuda") for d in range(3)]
    M.set_state(P.kappa, 1.0, 0.0, zeroF, [zeroF[d] for c in range(3) for d in range(3)])
    assert torch.equal(M.apply(x), x)                                          # A = I
    V0 = [rnd(P.nface[d]) for d in range(3)]
    W = [rnd(P.nface[d]) for c in range(3) for d in range(3)]
    M.set_state(P.kappa, 1.0, 1.0 / 512, V0, W)
    del V0, W
    ax, ay = M.apply(x), M.apply(y)
    lin = M.apply(2.0 * x - 3.0 * y) - (2.0 * ax - 3.0 * ay)
    assert float(lin.abs().max()) <= 1e-12 * float(ax.abs().max())
    dg = M.diagonal()
    assert float(dg.min()) > 1.0 and float(dg.max()) < 10.0                     # I + viscous diagonal (positive) + O(CFL) convection
    sol, info = M.solve(ax, rtol=1e-10, maxit=100)
    assert info["reason"] == 2 and info["iters"] < 40
    assert float((sol - x).norm()) <= 1e-7 * float(x.norm())
    del lin, ay, dg
    # PCApply_ABF with the multigrid pressure solve, then D V = 0 and agreement with the Jacobi-PCG pressure
    v, Vf, p, st = M.abf_apply(ax, momentum=KspOptions(type=capi.KSP_BCGS, rtol=1e-8, maxit=100), schur=KspOptions(pc=2, rtol=1e-9, maxit=60))
    assert st[0]["reason"] == 2 and st[1]["reason"] == 2 and st[1]["iters"] < 30
    div = P.rhs(*Vf)
    srhs_scale = float(P.rhs(*M.face_interp(v)).abs().max()) + 1.0
    assert float(div.abs().max()) <= 1e-6 * srhs_scale * 512
    M.close()
    P.close()


def test_vec_mdot_and_maxpy():
    """fl_vec_mdot / fl_vec_maxpy == VecMDot / VecMAXPY, for more vectors than one launch takes (8)."""
    import ctypes as C
    import torch
    from fluca_amd import capi
    from fluca_amd.poisson import Poisson
    P = Poisson.uniform((8, 8, 8), CAVITY_BOX, CAVITY, 1e-3)
    rng = np.random.default_rng(2)
    n, k = 10007, 11
    x = rng.standard_normal(n)
    Y = rng.standard_normal((k, n))
    xd, Yd = dev(x), [dev(y) for y in Y]
    ptrs = (C.c_void_p * k)(*[t.data_ptr() for t in Yd])
    out = (C.c_double * k)()
    capi.check(capi.lib.fl_vec_mdot(P.h, n, xd.data_ptr(), ptrs, k, out))
    assert np.allclose(np.array(out[:]), Y @ x, rtol=1e-12, atol=1e-12)
    a = rng.standard_normal(k)
    capi.check(capi.lib.fl_vec_maxpy(P.h, n, xd.data_ptr(), (C.c_double * k)(*a), ptrs, k))
    torch.cuda.synchronize()
    assert np.allclose(host(xd), x + a @ Y, rtol=1e-13, atol=1e-13)
    assert capi.lib.fl_vec_mdot(P.h, n, xd.data_ptr(), ptrs, -1, out) == -63
    P.close()


@pytest.mark.parametrize("n,bc,nonuni", [((20, 9, 6), CAVITY, True), ((12, 10, 9), [PER] * 6, False), ((17, 9, 11), [V, O, V, V, PER, PER], True)])
@pytest.mark.parametrize("pc,restart", [(fo.PC_JACOBI, 30), (fo.PC_JACOBI, 5), (fo.PC_NONE, 8)])
def test_gmres_matches_oracle(n, bc, nonuni, pc, restart):
    """KSPSolve(kspA) with the reference's default Krylov type (abfpc.c:72): restarted GMRES, classical Gram-Schmidt, left PC,
    against the oracle's restatement of the same algorithm -- iteration count, reason, the monitored-norm history and the answer."""
    from fluca_amd import capi
    P, M, g = _pair(n, bc, nonuni)
    V0, W = _fields(g)
    # CFL ~ 1 convection and a stiff viscous part, as in the BiCGStab test: a handful to a few dozen iterations
    hmin = min(np.diff(g.xf[d]).min() for d in range(3))
    dt, rho, mu = 0.5 * hmin, 1.0, 0.5 * hmin
    M.set_state(dt, rho, mu, [dev(a) for a in V0], [dev(a) for a in W])
    A = g.assemble_momentum(1.0, dt, -0.5 * mu * dt / rho, V0, W)
    b = np.random.default_rng(9).standard_normal(A.nrow)
    rtol = 1e-8
    xo, io = fo.gmres(A, b, pc=pc, rtol=rtol, restart=restart, maxit=300)
    assert io["iters"] > restart or restart == 30            # the short restarts are really exercised
    xg, ig = M.solve(dev(b), history=True, type=capi.KSP_GMRES, pc=pc, rtol=rtol, maxit=300, gmres_restart=restart)
    assert ig["reason"] == io["reason"] == 2
    assert abs(ig["iters"] - io["iters"]) <= 1, (ig["iters"], io["iters"])
    m = min(len(ig["history"]), len(io["history"]))
    assert np.allclose(ig["history"][:m], io["history"][:m], rtol=1e-6, atol=1e-12 * io["history"][0])
    assert np.linalg.norm(host(xg) - xo) <= 1e-6 * np.linalg.norm(xo)
    M.close()
    P.close()


def test_gmres_edge_cases():
    from fluca_amd import capi
    P, M, g = _pair((10, 8, 6), CAVITY, False)
    b = dev(np.zeros(3 * g.ncell))
    x, info = M.solve(b, type=capi.KSP_GMRES)                     # zero right-hand side: converged at once (atol), x = 0
    assert info["iters"] == 0 and info["reason"] == 3 and float(x.abs().max()) == 0.0
    b = dev(np.random.default_rng(1).standard_normal(3 * g.ncell))
    x, info = M.solve(b, type=capi.KSP_GMRES, maxit=3, rtol=1e-30)   # before set_state A = I: exact after one step (happy breakdown)
    assert info["reason"] in (2, 3) and info["iters"] == 1 and torch.allclose(x, b)   # residual exactly 0: below atol
    with pytest.raises(capi.FlucaError):
        P.solve(b[:g.ncell].contiguous(), type=capi.KSP_GMRES)        # kspS: gmres is not offered there
    M.close()
    P.close()
