"""The oracle's KSPCG with -ksp_cg_single_reduction against its own two-reduction KSPCG and against SciPy's CG: same iterates in
exact arithmetic, so the histories agree to round-off over the first iterations and both stop at the same iteration (+-1)."""
import numpy as np
import pytest
import scipy.sparse as sp
import scipy.sparse.linalg as spla

from oracle import fluca_oracle as fo

V, O, PER, SYM = fo.BC_VELOCITY, fo.BC_PRESSURE_OUTLET, fo.BC_PERIODIC, fo.BC_SYMMETRY
# symmetric S only (uniform grids without an outlet): the rearrangement uses (z, r_old) = 0, which is the symmetry of A and B; with the
# one-sided outlet rows S is not symmetric, CG itself is no longer a valid method there and the two forms part ways (104 against 136
# iterations on the 9 x 12 x 7 channel) -- PETSc's would too
CASES = [((12, 10, 9), [V, V, V, V, SYM, V], True), ((9, 12, 7), [V, V, SYM, SYM, PER, PER], True), ((16, 8, 8), [PER] * 6, True)]


@pytest.mark.parametrize("n,bc,nullspace", CASES)
@pytest.mark.parametrize("pc", [fo.PC_JACOBI, fo.PC_NONE])
@pytest.mark.parametrize("norm", [fo.NORM_PRECONDITIONED, fo.NORM_UNPRECONDITIONED, fo.NORM_NATURAL])
def test_single_reduction_equals_two_reductions(n, bc, nullspace, pc, norm):
    g = fo.Grid.uniform(n, [(0, 1), (0, 1), (0, 0.5)], bc, 1e-3)
    S = g.assemble_S()
    rng = np.random.default_rng(3)
    p = rng.uniform(-1, 1, g.ncell)
    if nullspace:
        p -= p.mean()
    b = S.mult(p)
    kw = dict(pc=pc, norm=norm, nullspace=nullspace, rtol=1e-9, maxit=400)
    x2, i2 = S.solve(b, **kw)
    x1, i1 = S.solve(b, single_reduction=True, **kw)
    assert i1["reason"] == i2["reason"] and abs(i1["iters"] - i2["iters"]) <= 1
    m = min(i1["iters"], i2["iters"], 25)
    assert np.allclose(i1["history"][:m], i2["history"][:m], rtol=1e-8)
    assert np.linalg.norm(x1 - x2) <= 1e-7 * np.linalg.norm(x2)


def test_single_reduction_against_scipy_cg():
    n, bc = (10, 9, 8), [V, V, V, V, PER, PER]      # singular but consistent: b = S p
    g = fo.Grid.uniform(n, [(0, 1), (0, 1), (0, 0.5)], bc, 1e-3)
    S = g.assemble_S()
    A = S.to_scipy()
    b = S.mult(np.random.default_rng(1).standard_normal(g.ncell))
    d = A.diagonal()
    res = []
    xs, info = spla.cg(A, b, rtol=1e-10, maxiter=500, M=sp.diags(1.0 / d), callback=lambda xk: res.append(np.linalg.norm(b - A @ xk)))
    assert info == 0
    x1, i1 = S.solve(b, single_reduction=True, pc=fo.PC_JACOBI, norm=fo.NORM_UNPRECONDITIONED, nullspace=False, rtol=1e-10, maxit=500)
    assert i1["reason"] == 2
    k = min(len(res), i1["iters"], 20)
    assert np.allclose(i1["history"][1:k + 1], res[:k], rtol=1e-6)
    assert np.linalg.norm((x1 - x1.mean()) - (xs - xs.mean())) <= 1e-6 * np.linalg.norm(xs - xs.mean())
