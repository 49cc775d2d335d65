"""CPU: independent cross-check of the oracle's Krylov restatements against SciPy's CG / BiCGStab.

This does NOT pin parity with PETSc (PETSc is absent) -- SciPy is a third implementation of the same
published algorithms.  What it rules out is an algorithmic slip in the oracle: in exact arithmetic preconditioned CG
produces the same iterates whatever library runs it, so the residual histories must agree to round-off for the first
iterations; BiCGStab likewise (SciPy runs the right-preconditioned form, so only the unpreconditioned case is compared).
"""
import numpy as np
import pytest
import scipy.sparse as sp
import scipy.sparse.linalg as spla

from oracle import fluca_oracle as fo

V, O, PER, SYM = fo.BC_VELOCITY, fo.BC_PRESSURE_OUTLET, fo.BC_PERIODIC, fo.BC_SYMMETRY


def _csr(A):
    rp, col, val = A.arrays()
    return sp.csr_matrix((val, col, rp), shape=(A.nrow, A.nrow))


@pytest.mark.parametrize("bc", [[V, V, V, V, PER, PER], [V, V, SYM, V, V, V]])
def test_cg_iterates_match_scipy(bc):
    n = (12, 10, 8)
    g = fo.Grid.uniform(n, [(0, 1)] * 3, bc, 1e-3)       # uniform grid, no outlet: S symmetric (and singular)
    S = g.assemble_S()
    M = _csr(S)
    assert abs(M - M.T).max() < 1e-12 * abs(M).max()
    p = np.random.default_rng(3).standard_normal(g.ncell)
    b = S.mult(p - p.mean())                              # consistent right-hand side: plain CG stays in the range of S
    d = S.diag()
    # oracle: PCJACOBI, natural norm sqrt(r.z) would need z; compare the UNPRECONDITIONED residual norms instead
    xo, io = S.solve(b, ksp=fo.KSP_CG, pc=fo.PC_JACOBI, norm=fo.NORM_UNPRECONDITIONED, nullspace=False, rtol=1e-10, maxit=400)
    hist = []
    Minv = spla.LinearOperator(M.shape, matvec=lambda r: r / d)
    xs, info = spla.cg(M, b, rtol=1e-10, atol=0.0, maxiter=400, M=Minv, callback=lambda xk: hist.append(np.linalg.norm(b - M @ xk)))
    assert info == 0 and io["reason"] > 0
    assert abs(len(hist) - io["iters"]) <= 1
    m = min(len(hist), io["iters"], 25)
    # history[0] is ||b||; SciPy's callback fires after every update
    assert np.allclose(io["history"][1:m + 1], hist[:m], rtol=1e-6)
    xo, xs = xo - xo.mean(), xs - xs.mean()
    assert np.linalg.norm(xo - xs) <= 1e-7 * np.linalg.norm(xs)


def test_bicgstab_matches_scipy_without_preconditioner():
    n = (10, 9, 8)
    xf = [np.linspace(0, 1, m + 1) ** 1.4 for m in n]
    g = fo.Grid(n, xf, [V, O, V, V, O, V], 1e-3)          # stretched: S is non-symmetric
    S = g.assemble_S()
    M = _csr(S)
    b = np.random.default_rng(4).standard_normal(g.ncell)
    xo, io = S.solve(b, ksp=fo.KSP_BCGS, pc=fo.PC_NONE, nullspace=False, rtol=1e-9, maxit=500)
    hist = []
    xs, info = spla.bicgstab(M, b, rtol=1e-9, atol=0.0, maxiter=500, callback=lambda xk: hist.append(np.linalg.norm(b - M @ xk)))
    assert info == 0 and io["reason"] > 0
    m = min(len(hist), io["iters"], 10)
    assert np.allclose(io["history"][1:m + 1], hist[:m], rtol=1e-5)
    assert abs(len(hist) - io["iters"]) <= max(3, io["iters"] // 8)
    assert np.linalg.norm(xo - xs) <= 1e-6 * np.linalg.norm(xs)


def test_momentum_bicgstab_matches_scipy():
    n = (8, 7, 6)
    xf = [np.linspace(0, 1, m + 1) ** 1.3 for m in n]
    g = fo.Grid(n, xf, [V, V, V, V, SYM, V], 1e-3)
    rng = np.random.default_rng(6)
    V0 = [rng.standard_normal(g.nface[d]) for d in range(3)]
    W = [rng.standard_normal(g.nface[d]) for c in range(3) for d in range(3)]
    A = g.assemble_momentum(1.0, 0.03, -0.5 * 0.05 * 0.03, V0, W)
    M = _csr(A)
    b = rng.standard_normal(3 * g.ncell)
    xo, io = A.solve(b, ksp=fo.KSP_BCGS, pc=fo.PC_NONE, nullspace=False, rtol=1e-10, maxit=300)
    hist = []
    xs, info = spla.bicgstab(M, b, rtol=1e-10, atol=0.0, maxiter=300, callback=lambda xk: hist.append(np.linalg.norm(b - M @ xk)))
    assert info == 0 and io["reason"] > 0
    m = min(len(hist), io["iters"], 8)
    assert np.allclose(io["history"][1:m + 1], hist[:m], rtol=1e-5)
    assert np.linalg.norm(xo - xs) <= 1e-7 * np.linalg.norm(xs)


def test_gmres_restatement_against_scipy():
    """The oracle's KSPGMRES restatement (oracle/fluca_oracle.py::gmres) on the assembled momentum matrix: same restarted,
    left-preconditioned algorithm as SciPy's gmres when SciPy is handed the Jacobi-scaled system -- iterates after every restart
    cycle and the final answer must agree (independent implementation of the same published algorithm)."""
    import scipy.sparse as sp
    import scipy.sparse.linalg as spla
    rng = np.random.default_rng(4)
    n = (9, 8, 7)
    xf = [np.linspace(0.0, 1.0, m + 1) ** (1.0 + 0.2 * d) for d, m in enumerate(n)]
    g = fo.Grid(n, xf, [fo.BC_VELOCITY] * 4 + [fo.BC_PERIODIC] * 2, kappa=1e-2)
    V0 = [rng.standard_normal(nf) for nf in g.nface]
    W = [rng.standard_normal(g.nface[d]) for c in range(3) for d in range(3)]
    A = g.assemble_momentum(1.0, 0.05, -0.02, V0, W)
    b = rng.standard_normal(A.nrow)
    x, info = fo.gmres(A, b, rtol=1e-9, restart=7, maxit=400)
    assert info["reason"] == 2
    As = A.to_scipy()
    Dinv = sp.diags(1.0 / A.diag())
    MA, Mb = (Dinv @ As).tocsr(), Dinv @ b
    xs, flag = spla.gmres(MA, Mb, rtol=1e-9, atol=0.0, restart=7, maxiter=400)
    assert flag == 0
    assert np.linalg.norm(x - xs) <= 1e-7 * np.linalg.norm(xs)
    assert np.linalg.norm(Mb - MA @ x) <= 1.01e-9 * np.linalg.norm(Mb)
    # the recurrence norm is the true preconditioned residual norm (in exact arithmetic): check at the end of the first cycle
    x7, i7 = fo.gmres(A, b, rtol=1e-30, restart=7, maxit=7)
    assert i7["iters"] == 7 and i7["reason"] == -3
    assert abs(np.linalg.norm(Mb - MA @ x7) - i7["history"][-1]) <= 1e-10 * i7["history"][0]
    assert np.all(np.diff(i7["history"]) <= 1e-14 * i7["history"][0])     # GMRES residuals never grow inside a cycle
