"""CPU: the oracle's multigrid restatement behaves like a multigrid method (no reference golden exists: the algorithm is
specified by this build, fl_mg.hip / DESIGN.md section 10)."""
import numpy as np
import pytest

from oracle import fluca_oracle as fo

V, O, PER, SYM = fo.BC_VELOCITY, fo.BC_PRESSURE_OUTLET, fo.BC_PERIODIC, fo.BC_SYMMETRY


@pytest.mark.parametrize("bc,nullspace", [([V, V, V, V, SYM, V], True), ([PER] * 6, True), ([V, O, V, V, PER, PER], False)])
def test_mg_pcg_converges_fast_and_grid_independently(bc, nullspace):
    its = []
    for n in ((16, 16, 8), (32, 32, 16)):
        g = fo.Grid.uniform(n, [(0, 1), (0, 1), (0, 0.5)], bc, 1e-3)
        mg = fo.MgOracle(g, nullspace=nullspace)
        assert mg.nlevels >= 2
        S = g.assemble_S()
        rng = np.random.default_rng(2)
        p = rng.standard_normal(g.ncell)
        p -= p.mean() if nullspace else 0.0
        b = S.mult(p)
        x, info = mg.pcg(b, rtol=1e-8, maxit=60)
        assert info["reason"] == 2
        res = np.linalg.norm(b - S.mult(x)) / np.linalg.norm(b)
        assert res < 1e-6
        xo, io = S.solve(b, nullspace=nullspace, rtol=1e-8, maxit=5000)
        assert info["iters"] * 4 < io["iters"]              # far fewer iterations than Jacobi-PCG
        its.append(info["iters"])
    assert its[1] <= its[0] + 4                               # iteration count hardly grows with the grid


def test_transfer_operators():
    g = fo.Grid(np.array([8, 8, 8]), [np.linspace(0, 1, 9) ** 1.5, np.linspace(0, 2, 9), np.linspace(0, 1, 9)], [V] * 6, 1.0)
    mg = fo.MgOracle(g, max_levels=2)
    assert mg.nlevels == 2 and mg.ratio[0] == [2, 2, 2]
    one = np.ones(g.ncell)
    assert np.allclose(mg._restrict(0, one), 1.0)            # volume-weighted average preserves constants
    assert np.allclose(mg._prolong(0, np.ones(mg.grids[1].ncell)), 1.0)
    # restriction conserves the volume integral
    vol = np.einsum("k,j,i->kji", *[np.diff(g.xf[d]) for d in (2, 1, 0)]).ravel()
    volc = np.einsum("k,j,i->kji", *[np.diff(mg.grids[1].xf[d]) for d in (2, 1, 0)]).ravel()
    f = np.random.default_rng(0).standard_normal(g.ncell)
    assert (mg._restrict(0, f) * volc).sum() == pytest.approx((f * vol).sum())


def test_trilinear_prolongation_is_exact_on_linear_fields_and_cuts_iterations():
    """prolong = "linear" (the product's default, tuning knob "mg_prolong" = 1): constants are preserved; a field that is linear in the
    coordinates is reproduced exactly on every fine cell whose parents' neighbours all exist (periodic box: every cell, up to the seam,
    which is excluded by taking a field periodic in nothing -- so walls, interior cells only); fewer iterations than piecewise constant."""
    xf = [np.linspace(0, 1, 17) ** 1.3, np.linspace(0, 2, 17), np.linspace(0, 1, 9)]
    g = fo.Grid(np.array([16, 16, 8]), xf, [V] * 6, 1.0)
    mg = fo.MgOracle(g, max_levels=2, prolong="linear")
    gc = mg.grids[1]
    assert np.allclose(mg._prolong(0, np.ones(gc.ncell)), 1.0)
    cen = lambda gg, d: 0.5 * (np.asarray(gg.xf[d])[:-1] + np.asarray(gg.xf[d])[1:])
    lin = lambda gg: (2.0 * cen(gg, 0)[None, None, :] - 0.7 * cen(gg, 1)[None, :, None] + 1.3 * cen(gg, 2)[:, None, None])
    fine = mg._prolong(0, lin(gc).ravel()).reshape(8, 16, 16)
    inner = (slice(1, -1),) * 3                                 # the first / last cell of an axis has no neighbour on its wall side
    assert np.allclose(fine[inner], lin(g)[inner], rtol=0, atol=1e-13)
    # fewer outer iterations than with the piecewise-constant prolongation, markedly so with fewer smoothing steps
    g2 = fo.Grid.uniform((32, 32, 16), [(0, 1), (0, 1), (0, 0.5)], [V, V, V, V, SYM, V], 1e-3)
    S = g2.assemble_S()
    p = np.random.default_rng(3).standard_normal(g2.ncell)
    p -= p.mean()
    b = S.mult(p)
    for nu in (3, 1):
        its = [fo.MgOracle(g2, nu=nu, prolong=pr).pcg(b, rtol=1e-8, maxit=200)[1]["iters"] for pr in ("constant", "linear")]
        assert its[1] < its[0], (nu, its)


def _stretched(n, lo, hi, beta):
    s = np.linspace(-1.0, 1.0, n + 1)
    return lo + (hi - lo) * 0.5 * (1.0 + np.tanh(beta * s) / np.tanh(beta))


def test_flexible_beta_keeps_a_nonsymmetric_cycle_converging():
    """The V-cycle is not a symmetric operator once R is not a multiple of P^T (tri-linear P against volume-weighted R, or any pair on a
    stretched grid), and KSPCG's beta = r_new.z_new / r_old.z_old assumes one: on a stretched channel with an outlet and two smoothing steps
    it needs several times the iterations of the Polak-Ribiere form (the product's default, "mg_flexible" = 1), which also is never worse
    where the cycle IS symmetric (uniform walls, piecewise-constant transfer: the two forms agree there up to round-off)."""
    n = (40, 24, 16)
    xf = [_stretched(n[d], 0.0, (1.0, 1.0, 0.5)[d], 1.1 + 0.2 * d) for d in range(3)]
    g = fo.Grid(n, xf, [O, V, V, V, V, V], 1e-3)
    S = g.assemble_S()
    b = S.mult(np.random.default_rng(3).standard_normal(g.ncell))
    its = {fl: fo.MgOracle(g, nullspace=False, nu=2, prolong="linear", flexible=fl).pcg(b, rtol=1e-8, maxit=200)[1] for fl in (False, True)}
    assert its[True]["reason"] == 2 and its[True]["iters"] <= 30, its[True]["iters"]
    assert its[False]["iters"] >= 2 * its[True]["iters"], (its[False]["iters"], its[True]["iters"])
    gu = fo.Grid.uniform((32, 32, 16), [(0, 1), (0, 1), (0, 0.5)], [V, V, V, V, SYM, V], 1e-3)
    Su = gu.assemble_S()
    p = np.random.default_rng(3).standard_normal(gu.ncell)
    bu = Su.mult(p - p.mean())
    same = [fo.MgOracle(gu, nu=3, prolong="constant", flexible=fl).pcg(bu, rtol=1e-8, maxit=100)[1] for fl in (False, True)]
    assert same[0]["iters"] == same[1]["iters"] and np.allclose(same[0]["history"], same[1]["history"], rtol=2e-3)   # the coarsest level is solved to a tolerance: not exactly a fixed operator
