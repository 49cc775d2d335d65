"""-m gpu: the multigrid-preconditioned CG (FL_PC_MG, fl_mg.hip) through the C-ABI against the CPU restatement of the same
algorithm (oracle MgOracle).  No reference function exists for it."""
import ctypes as C

import numpy as np
import pytest

from oracle import fluca_oracle as fo
from tests.gpu_common import CAVITY, CAVITY_BOX, O, PER, SYM, V, dev, host, make_pair, stretched

pytestmark = pytest.mark.gpu


def _bounds(mg):
    """the eigenvalue bound the product uses on every level (its 1-D separable Gershgorin bound)"""
    from fluca_amd import capi
    from fluca_amd.poisson import Poisson
    out = []
    for g in mg.grids:
        P = Poisson(g.n, g.xf, g.bc, g.kappa)
        lam = C.c_double()
        capi.check(capi.lib.fl_poisson_gershgorin(P.h, capi.PC_JACOBI, C.byref(lam)))
        out.append(lam.value)
        P.close()
    return out


@pytest.mark.parametrize("n,bc,nonuni,nullspace", [
    ((32, 32, 16), CAVITY, False, True),
    ((32, 16, 16), [PER] * 6, False, True),
    ((24, 20, 16), [V, O, V, V, PER, PER], False, False),       # 20 -> 10 -> stops (10 = 2*5 but 5 is odd): semi-coarsening
    ((32, 24, 16), CAVITY, True, True),
])
@pytest.mark.parametrize("prolong", ["constant", "linear"])
def test_mg_pcg_matches_oracle(n, bc, nonuni, nullspace, prolong):
    from fluca_amd import capi
    capi.check(capi.lib.fl_tuning_set(b"mg_prolong", 1 if prolong == "linear" else 0))
    try:
        _mg_pcg_matches_oracle(n, bc, nonuni, nullspace, prolong)
    finally:
        capi.check(capi.lib.fl_tuning_set(b"mg_prolong", 1))   # the default


def _mg_pcg_matches_oracle(n, bc, nonuni, nullspace, prolong):
    P, g = make_pair(n, bc, kappa=1e-3, nonuniform=nonuni)
    S = g.assemble_S()
    rng = np.random.default_rng(3)
    p = rng.standard_normal(g.ncell)
    if nullspace:
        p -= p.mean()
    b = S.mult(p)
    mg = fo.MgOracle(g, nullspace=nullspace)
    mg = fo.MgOracle(g, nullspace=nullspace, bounds=_bounds(mg), prolong=prolong)
    xo, io = mg.pcg(b, rtol=1e-8, maxit=50)
    xg, ig = P.solve(dev(b), history=True, type=0, pc=2, remove_nullspace=int(nullspace), rtol=1e-8, maxit=50)
    assert ig["reason"] == io["reason"] == 2
    assert abs(ig["iters"] - io["iters"]) <= 1, (ig["iters"], io["iters"])
    m = min(len(ig["history"]), len(io["history"]))
    # the coarsest-level CG stops on a tolerance: a one-iteration difference there perturbs the cycle at the 1e-3 level
    assert np.allclose(ig["history"][:3], io["history"][:3], rtol=1e-6)
    assert np.allclose(ig["history"][:m], io["history"][:m], rtol=5e-2)
    xg = host(xg)
    res = np.linalg.norm(b - S.mult(xg)) / np.linalg.norm(b)
    assert res < 1e-6
    if nullspace:
        xg = xg - xg.mean()
    assert np.linalg.norm(xg - xo) <= 1e-6 * np.linalg.norm(xo)
    P.close()


def test_mg_vs_jacobi_iteration_counts_and_reuse():
    P, g = make_pair((64, 64, 32), CAVITY, kappa=1e-3)
    S = g.assemble_S()
    rng = np.random.default_rng(4)
    p = rng.standard_normal(g.ncell)
    p -= p.mean()
    b = dev(S.mult(p))
    xj, ij = P.solve(b, type=0, pc=1, rtol=1e-8, maxit=5000)
    xm, im = P.solve(b, type=0, pc=2, rtol=1e-8, maxit=100)
    xm2, im2 = P.solve(b, type=0, pc=2, rtol=1e-8, maxit=100)         # hierarchy is reused
    assert ij["reason"] == im["reason"] == 2 and im2["iters"] == im["iters"]
    assert im["iters"] * 8 < ij["iters"], (im["iters"], ij["iters"])
    a, c = host(xj), host(xm)
    assert np.linalg.norm((a - a.mean()) - (c - c.mean())) <= 1e-5 * np.linalg.norm(a - a.mean())
    with pytest.raises(RuntimeError):
        P.solve(b, type=1, pc=2)                                      # BiCGStab + MG: not built
    # -pc_mg_levels follows the request in BOTH directions: deep, shallow, deep again must give deep's history again (round 5: a hierarchy once
    # built for fewer levels used to answer a later request for more)
    hist = {}
    for step, lev in enumerate((0, 2, 0, 3, 2)):
        _, info = P.solve(b, type=0, pc=2, rtol=1e-8, maxit=100, mg_levels=lev, history=True)
        assert info["reason"] == 2
        if lev in hist:
            assert np.array_equal(hist[lev], info["history"]), (step, lev)
        hist[lev] = info["history"]
    assert hist[2][0] != hist[0][0] and hist[3][0] != hist[0][0] and hist[2][0] != hist[3][0]     # three different cycles: ||M^-1 b|| differs
    P.close()


@pytest.mark.parametrize("n,levels", [((15, 15, 15), 0), ((32, 32, 16), 1)])
def test_one_level_hierarchy_keeps_the_outer_residual(n, levels):
    """No axis can be coarsened (odd cell counts) or -pc_mg_levels 1: the "cycle" is the coarse solve on the fine handle
    itself, whose work vectors are the outer CG's too.  The outer residual must survive every preconditioner apply."""
    P, g = make_pair(n, CAVITY, kappa=1e-3)
    S = g.assemble_S()
    rng = np.random.default_rng(3)
    p = rng.standard_normal(g.ncell)
    p -= p.mean()
    b = S.mult(p)
    mg = fo.MgOracle(g, max_levels=levels, nullspace=True)
    assert mg.nlevels == 1
    xo, io = mg.pcg(b, rtol=1e-8, maxit=100)
    xg, ig = P.solve(dev(b), history=True, type=0, pc=2, remove_nullspace=1, rtol=1e-8, maxit=100, mg_levels=levels)
    assert ig["reason"] == io["reason"] == 2
    assert abs(ig["iters"] - io["iters"]) <= 1, (ig["iters"], io["iters"])
    xg = host(xg)
    assert np.linalg.norm(b - S.mult(xg)) / np.linalg.norm(b) < 1e-6      # a clobbered residual reports convergence with a wrong x
    assert np.linalg.norm((xg - xg.mean()) - xo) <= 1e-5 * np.linalg.norm(xo)
    P.close()


@pytest.mark.parametrize("n,bc,nonuni,nullspace,nu", [
    ((40, 24, 16), [O, V, V, V, V, V], True, False, 2),            # stretched + outlet, two smoothing steps: KSPCG's beta needs 80 iterations here
    ((48, 48, 24), [V, O, V, V, PER, PER], True, False, 2),        # the channel of BASELINE config 3, stretched: 100 with KSPCG's beta
    ((64, 32, 32), [PER, PER, V, V, SYM, V], True, True, 3),       # stretched, two periodic axes, null space
    ((32, 16, 16), [PER] * 6, False, True, 3),
])
def test_flexible_beta_is_the_default_and_matches_the_oracle(n, bc, nonuni, nullspace, nu):
    """ADVICE (round 3): the tri-linear prolongation against the volume-weighted restriction makes the cycle a non-symmetric operator.  The
    outer CG therefore takes beta in the Polak-Ribiere form ("mg_flexible" = 1, the default): same iteration counts as the oracle's
    restatement of that form, and never more than with the piecewise-constant transfer."""
    from fluca_amd import capi
    P, g = make_pair(n, bc, kappa=1e-3, nonuniform=nonuni)
    S = g.assemble_S()
    p = np.random.default_rng(3).standard_normal(g.ncell)
    if nullspace:
        p -= p.mean()
    b = S.mult(p)
    mg0 = fo.MgOracle(g, nullspace=nullspace, nu=nu)
    bounds = _bounds(mg0)
    its = {}
    try:
        for prolong in ("linear", "constant"):
            capi.check(capi.lib.fl_tuning_set(b"mg_prolong", 1 if prolong == "linear" else 0))
            xo, io = fo.MgOracle(g, nullspace=nullspace, nu=nu, bounds=bounds, prolong=prolong).pcg(b, rtol=1e-8, maxit=200)
            xg, ig = P.solve(dev(b), history=True, type=0, pc=2, remove_nullspace=int(nullspace), rtol=1e-8, maxit=200, mg_smooth_its=nu)
            assert ig["reason"] == io["reason"] == 2, (prolong, ig["iters"], io["iters"], ig["reason"], io["reason"])
            assert abs(ig["iters"] - io["iters"]) <= max(1, io["iters"] // 10), (prolong, ig["iters"], io["iters"])
            assert np.allclose(ig["history"][:3], io["history"][:3], rtol=1e-6)
            xg = host(xg)
            assert np.linalg.norm(b - S.mult(xg)) <= 1e-6 * np.linalg.norm(b)
            its[prolong] = ig["iters"]
        assert its["linear"] <= its["constant"], its
        # and with KSPCG's beta the tri-linear cycle is what the advisor feared on the first two grids
        if nu == 2:
            capi.check(capi.lib.fl_tuning_set(b"mg_prolong", 1))
            capi.check(capi.lib.fl_tuning_set(b"mg_flexible", 0))
            _, i0 = P.solve(dev(b), type=0, pc=2, remove_nullspace=int(nullspace), rtol=1e-8, maxit=200, mg_smooth_its=nu)
            assert i0["iters"] >= 2 * its["linear"], (i0["iters"], its["linear"])
    finally:
        capi.check(capi.lib.fl_tuning_set(b"mg_prolong", 1))
        capi.check(capi.lib.fl_tuning_set(b"mg_flexible", 1))
    P.close()


@pytest.mark.parametrize("n,bc,nonuni,nullspace,nu", [
    ((160, 40, 36), CAVITY, False, True, 3),                       # two tiles in x (one partial), three in y, two z chunks
    ((64, 48, 40), [V, O, V, V, PER, PER], True, False, 3),        # outlet, two periodic axes, stretched
    ((32, 32, 32), [PER] * 6, False, True, 3),                     # the ring wraps on every axis
    ((130, 18, 34), [V, V, PER, PER, SYM, V], True, True, 4),      # two cells past a tile in x and y; four steps: the sweep + a single step
    ((34, 10, 12), CAVITY, False, True, 5),                        # a tile that is mostly ring; five steps: the sweep + a pair
])
def test_three_smoothing_steps_from_zero_in_one_sweep(n, bc, nonuni, nullspace, nu):
    """Round 5: the pre-smoother of a cycle (zero initial guess) runs its first three steps -- the stencil-free first one included, and the outer CG's
    r -= alpha q on the fine level -- in ONE sweep of the fused kernel (fl_cheb2.hip, Z; knob "cheb_zero3").  Same numbers as the separate first
    step + fused pair: histories of a whole solve agree to rounding (the arithmetic is the same, only the order of the passes changed), and both
    agree with the oracle."""
    from fluca_amd import capi
    P, g = make_pair(n, bc, kappa=1e-3, nonuniform=nonuni)
    S = g.assemble_S()
    p = np.random.default_rng(11).standard_normal(g.ncell)
    if nullspace:
        p -= p.mean()
    b = S.mult(p)
    mg0 = fo.MgOracle(g, nullspace=nullspace, nu=nu)
    xo, io = fo.MgOracle(g, nullspace=nullspace, nu=nu, bounds=_bounds(mg0), prolong="linear").pcg(b, rtol=1e-9, maxit=100)
    out = {}
    try:
        capi.check(capi.lib.fl_tuning_set(b"cheb_fuse", 2))          # fused wherever legal: these grids are below the size where it pays
        for z3 in (1, 0):
            capi.check(capi.lib.fl_tuning_set(b"cheb_zero3", z3))
            xg, ig = P.solve(dev(b), history=True, type=0, pc=2, remove_nullspace=int(nullspace), rtol=1e-9, maxit=100, mg_smooth_its=nu)
            assert ig["reason"] == 2, (z3, ig)
            out[z3] = (host(xg), ig)
    finally:
        capi.check(capi.lib.fl_tuning_set(b"cheb_fuse", 1))
        capi.check(capi.lib.fl_tuning_set(b"cheb_zero3", 1))
    (x1, i1), (x0, i0) = out[1], out[0]
    assert i1["iters"] == i0["iters"]
    # (rounding differs in the last place -- other contractions of the same expressions -- and a Krylov iteration amplifies that: 6e-9 after 50 steps)
    assert np.allclose(i1["history"][:8], i0["history"][:8], rtol=1e-10, atol=0.0)
    assert np.allclose(i1["history"], i0["history"], rtol=1e-6, atol=0.0)
    assert np.linalg.norm(x1 - x0) <= 1e-8 * np.linalg.norm(x0)
    assert abs(i1["iters"] - io["iters"]) <= 1, (i1["iters"], io["iters"])
    assert np.allclose(i1["history"][:3], io["history"][:3], rtol=1e-6)
    assert np.linalg.norm(b - S.mult(x1)) <= 1e-7 * np.linalg.norm(b)
    P.close()
