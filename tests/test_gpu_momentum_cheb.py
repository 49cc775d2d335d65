"""-m gpu: KSPCHEBYSHEV on the momentum block (fl_momentum_solve with FL_KSP_CHEBYSHEV; -ns_abf_momentum_ksp_type chebyshev) and the
Gershgorin bound its default interval comes from, against the oracle: the assembled A (oracle/fluca_oracle.c, cnlinearcart3d.c:425-646,
873-1294, 2930-2941 restated row by row), its fo_gershgorin_dinvA, and the KSPCHEBYSHEV + PCJACOBI restatement (PETSc's three-term
recurrence; PARITY UNPINNED like every Krylov restatement here).  Two code paths: the step fused into the product (k_mom3, state handed
over with v0, ny > 8) and the product followed by a vector update (every other state)."""
import numpy as np
import pytest

from oracle import fluca_oracle as fo
from tests.gpu_common import CAVITY, O, PER, SYM, V, dev, host
from tests.test_gpu_momentum import _pair

pytestmark = pytest.mark.gpu

CASES = [
    ((17, 9, 11), CAVITY, False, 0.4),                       # fused: one tile, all walls
    ((12, 10, 9), [PER] * 6, False, 0.6),                    # fused: every ring cell through a periodic seam
    ((9, 12, 7), [V, O, V, V, PER, PER], True, 0.3),         # fused: channel boundary types, stretched
    ((130, 37, 20), CAVITY, True, 0.5),                      # fused: several tiles, ragged in x and y
    ((70, 5, 3), [PER, PER, V, V, O, SYM], False, 0.5),      # ny <= 8: the stored path (k_mom2 + vector update)
    ((64, 48, 40), [V, O, V, V, PER, PER], False, 0.8),      # fused: several z chunks
]


def _state(g, vmag, seed=23):
    """A smooth-ish state: CFL-sized convection beside a viscous part of the same size, so that D^-1 A is diagonally dominant."""
    rng = np.random.default_rng(seed)
    V0 = [vmag * rng.uniform(-1, 1, g.nface[d]) for d in range(3)]
    v0 = vmag * rng.uniform(-1, 1, 3 * g.ncell)
    hmin = min(np.diff(g.xf[d]).min() for d in range(3))
    dt, rho = 0.4 * hmin, 1.3
    mu = 0.8 * rho * hmin * hmin / dt                          # mu dt / (rho h^2) = 0.8
    return V0, v0, dt, rho, mu


@pytest.mark.parametrize("n,bc,nonuni,vmag", CASES)
@pytest.mark.parametrize("with_v0", [True, False])
def test_gershgorin_and_chebyshev_match_the_oracle(n, bc, nonuni, vmag, with_v0):
    P, M, g = _pair(n, bc, nonuni)
    V0, v0, dt, rho, mu = _state(g, vmag)
    W = g.apply_B(v0)
    A = g.assemble_momentum(1.0, dt, -0.5 * mu * dt / rho, V0, W)
    Wd = M.interp_faces(dev(v0))
    M.set_state(dt, rho, mu, [dev(a) for a in V0], Wd, v0=dev(v0) if with_v0 else None)
    G = A.gershgorin(fo.PC_JACOBI)                             # max_i sum_j |a_ij| / |a_ii|, the diagonal included
    radius = M.gershgorin()
    # a periodic axis of two cells folds the m and p columns of a row into one matrix entry (|am + ap| <= |am| + |ap|): not among these cases
    assert abs((1.0 + radius) - G) <= 1e-12 * G, (radius, G)
    b = np.random.default_rng(5).standard_normal(3 * g.ncell)
    emin, emax = M.chebyshev_interval()
    assert emax == 1.0 + radius and abs(emin - max(1.0 - radius, 0.9 / A.diag().mean())) <= 1e-12
    for norm, rtol in ((fo.NORM_PRECONDITIONED, 1e-8), (fo.NORM_UNPRECONDITIONED, 1e-6)):
        xo, io = A.solve(b, ksp=fo.KSP_CHEBYSHEV, pc=fo.PC_JACOBI, norm=norm, nullspace=False, rtol=rtol, maxit=400, emin=emin, emax=emax)
        # the default interval IS (emin, emax): nothing is passed
        xg, ig = M.solve(dev(b), type=fo.KSP_CHEBYSHEV, pc=fo.PC_JACOBI, norm_type=norm, rtol=rtol, maxit=400, history=True, check_every=5)
        assert io["reason"] > 0 and ig["reason"] == io["reason"], (ig, io["reason"], io["iters"])
        assert ig["iters"] == io["iters"]
        assert np.allclose(ig["history"][:io["iters"] + 1], io["history"][:io["iters"] + 1], rtol=1e-8, atol=1e-13 * io["history"][0])
        assert np.linalg.norm(host(xg) - xo) <= 1e-9 * np.linalg.norm(xo)
    # a fixed number of steps without a norm (the smoother's way of calling), explicit interval, no preconditioner
    lam = A.gershgorin(fo.PC_NONE)
    xo, io = A.solve(b, ksp=fo.KSP_CHEBYSHEV, pc=fo.PC_NONE, norm=fo.NORM_NONE, nullspace=False, maxit=7, emin=0.1 * lam, emax=1.1 * lam)
    xg, ig = M.solve(dev(b), type=fo.KSP_CHEBYSHEV, pc=fo.PC_NONE, norm_type=fo.NORM_NONE, maxit=7, emin=0.1 * lam, emax=1.1 * lam)
    assert ig["iters"] == io["iters"] == 7 and ig["reason"] == io["reason"] == 4
    assert np.linalg.norm(host(xg) - xo) <= 1e-10 * np.linalg.norm(xo)
    M.close()
    P.close()


def test_fused_step_equals_product_plus_update():
    """The same state with and without v0: k_mom3's fused step against k_mom2 + the vector update -- the same recurrence on operators that agree
    to the rounding of the row sums."""
    P, M, g = _pair((130, 37, 20), [V, O, V, V, PER, PER], True)
    V0, v0, dt, rho, mu = _state(g, 0.5)
    Wd = M.interp_faces(dev(v0))
    b = dev(np.random.default_rng(9).standard_normal(3 * g.ncell))
    out = {}
    for fused in (True, False):
        M.set_state(dt, rho, mu, [dev(a) for a in V0], Wd, v0=dev(v0) if fused else None)
        x, info = M.solve(b, type=fo.KSP_CHEBYSHEV, pc=fo.PC_JACOBI, rtol=1e-10, maxit=300, history=True)
        assert info["reason"] == 2
        out[fused] = (host(x), info)
    assert out[True][1]["iters"] == out[False][1]["iters"]
    assert np.allclose(out[True][1]["history"], out[False][1]["history"], rtol=1e-9)
    assert np.linalg.norm(out[True][0] - out[False][0]) <= 1e-11 * np.linalg.norm(out[False][0])
    M.close()
    P.close()


def test_chebyshev_refuses_what_it_cannot_bound():
    from fluca_amd import capi
    P, M, g = _pair((12, 10, 9), CAVITY, False)
    V0, v0, dt, rho, mu = _state(g, 0.3)
    M.set_state(dt, rho, mu, [dev(a) for a in V0], M.interp_faces(dev(v0)), v0=dev(v0))
    b = dev(np.ones(3 * g.ncell))
    with pytest.raises(capi.FlucaError):                         # no preconditioner and no interval: PETSC_ERR_SUP
        M.solve(b, type=fo.KSP_CHEBYSHEV, pc=fo.PC_NONE)
    with pytest.raises(capi.FlucaError):                         # emin <= 0
        M.solve(b, type=fo.KSP_CHEBYSHEV, pc=fo.PC_JACOBI, emin=-1.0, emax=2.0)
    with pytest.raises(capi.FlucaError):
        M.solve(b, type=fo.KSP_CHEBYSHEV, pc=fo.PC_JACOBI, norm_type=fo.NORM_NATURAL)
    M.close()
    P.close()


def test_an_interval_that_is_too_short_ends_in_dtol_and_the_handle_recovers():
    """emax below the spectrum: the iteration grows until KSPConvergedDefault's divergence test stops it (dtol 1e5), as the oracle's restatement does;
    the next solve on the same handle is not affected by what the diverged one left in the work vectors."""
    P, M, g = _pair((24, 20, 12), CAVITY, False)
    V0, v0, dt, rho, mu = _state(g, 0.4)
    W = g.apply_B(v0)
    A = g.assemble_momentum(1.0, dt, -0.5 * mu * dt / rho, V0, W)
    M.set_state(dt, rho, mu, [dev(a) for a in V0], M.interp_faces(dev(v0)), v0=dev(v0))
    b = np.random.default_rng(2).standard_normal(3 * g.ncell)
    lam = A.gershgorin(fo.PC_JACOBI)
    xo, io = A.solve(b, ksp=fo.KSP_CHEBYSHEV, pc=fo.PC_JACOBI, nullspace=False, rtol=1e-8, maxit=400, emin=0.05, emax=0.45 * lam)
    xg, ig = M.solve(dev(b), type=fo.KSP_CHEBYSHEV, pc=fo.PC_JACOBI, rtol=1e-8, maxit=400, emin=0.05, emax=0.45 * lam, check_every=1, history=True)
    assert io["reason"] == -4 and ig["reason"] == -4, (io["reason"], ig["reason"])          # KSP_DIVERGED_DTOL
    assert abs(ig["iters"] - io["iters"]) <= 1, (ig["iters"], io["iters"])
    x2, i2 = M.solve(dev(b), type=fo.KSP_CHEBYSHEV, pc=fo.PC_JACOBI, rtol=1e-8, maxit=400)
    xr, ir = A.solve(b, ksp=fo.KSP_CHEBYSHEV, pc=fo.PC_JACOBI, nullspace=False, rtol=1e-8, maxit=400, emin=M.chebyshev_interval()[0], emax=M.chebyshev_interval()[1])
    assert i2["reason"] == ir["reason"] == 2 and i2["iters"] == ir["iters"]
    assert np.linalg.norm(host(x2) - xr) <= 1e-9 * np.linalg.norm(xr)
    M.close()
    P.close()


@pytest.mark.parametrize("ksp", [fo.KSP_BCGS, fo.KSP_CHEBYSHEV, 3])          # 3 = FL_KSP_GMRES
@pytest.mark.parametrize("n,bc,nonuni", [((17, 9, 11), CAVITY, False), ((64, 48, 40), [V, O, V, V, PER, PER], False), ((9, 12, 7), [V, O, V, V, PER, PER], True)])
def test_solve_from_a_nonzero_initial_guess(n, bc, nonuni, ksp):
    """fl_ksp_opts.initial_guess_nonzero (-ksp_initial_guess_nonzero on kspA, round 5): x holds the guess; the convergence test compares with the norm of
    the RIGHT-HAND SIDE (KSPConvergedDefault's rule for a non-zero guess).  (a) a zero guess reproduces the plain solve; (b) a guess that is close
    needs fewer iterations, and its answer meets the same test -- the true preconditioned residual is below rtol || M b || -- and agrees with the
    oracle's converged solution; (c) a guess that already meets the test costs no iteration."""
    P, M, g = _pair(n, bc, nonuni)
    V0, v0, dt, rho, mu = _state(g, 0.4)
    W = g.apply_B(v0)
    A = g.assemble_momentum(1.0, dt, -0.5 * mu * dt / rho, V0, W)
    M.set_state(dt, rho, mu, [dev(a) for a in V0], M.interp_faces(dev(v0)), v0=dev(v0))
    rng = np.random.default_rng(9)
    b = rng.standard_normal(3 * g.ncell)
    rtol = 1e-7
    kw = dict(type=ksp, pc=fo.PC_JACOBI, rtol=rtol, maxit=400, check_every=3)
    xo, _ = A.solve(b, ksp=fo.KSP_BCGS, pc=fo.PC_JACOBI, nullspace=False, rtol=1e-13, maxit=1000)     # the converged answer
    dinv = 1.0 / A.diag()
    bnorm = np.linalg.norm(dinv * b)

    def pres(x):            # || M (b - A x) || / || M b ||
        return np.linalg.norm(dinv * (b - A.mult(x))) / bnorm

    xp, ip = M.solve(dev(b), **kw)                                                   # plain: zero guess
    assert ip["reason"] == 2 and pres(host(xp)) <= 1.01 * rtol
    x0, i0 = M.solve(dev(b), x=dev(np.zeros_like(b)), initial_guess_nonzero=1, **kw)  # (a)
    assert i0["reason"] == 2 and i0["iters"] == ip["iters"]
    assert np.linalg.norm(host(x0) - host(xp)) <= 1e-12 * np.linalg.norm(host(xp))
    guess = xo + 1e-3 * np.linalg.norm(xo) / np.sqrt(xo.size) * rng.standard_normal(xo.size)
    xg, ig = M.solve(dev(b), x=dev(guess), initial_guess_nonzero=1, **kw)              # (b)
    assert ig["reason"] == 2 and 0 < ig["iters"] < ip["iters"], (ig["iters"], ip["iters"])
    assert pres(host(xg)) <= 1.01 * rtol
    assert np.linalg.norm(host(xg) - xo) <= 50 * rtol * np.linalg.norm(xo)
    assert abs(ig["rnorm0"] - bnorm) <= 1e-10 * bnorm                                 # the reference norm of the test is || M b ||
    xe, ie = M.solve(dev(b), x=dev(xo), initial_guess_nonzero=1, **kw)                # (c)
    assert ie["iters"] == 0 and ie["reason"] in (2, 3) and np.linalg.norm(host(xe) - xo) <= 1e-12 * np.linalg.norm(xo)
    # the Schur solvers start from zero, as the reference's kspS does: the flag is refused there
    from fluca_amd.capi import FlucaError
    with pytest.raises(FlucaError):
        P.solve(dev(np.ones(g.ncell)), initial_guess_nonzero=1)
    M.close()
    P.close()
