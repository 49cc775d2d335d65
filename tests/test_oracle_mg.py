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
