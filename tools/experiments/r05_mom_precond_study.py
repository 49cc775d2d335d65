"""Round 5, VERDICT item 3 (fallback branch): would a stronger preconditioner for the momentum block A = I + dt C - (mu dt / 2 rho) L pay?
CPU study on the oracle's assembled A (channel state: parabolic inflow everywhere, the flow configurations' nu dt / h^2): extreme eigenvalues of
M^-1 A (ARPACK) for M = diag(A) (PCJACOBI, what k_mom3<OUT=4> applies), M = the z-line blocks of A (tridiagonal solves along the kernel's march
direction), M = y-lines, M = the ADI product of the three line operators; Chebyshev steps to rtol 1e-5 predicted from kappa and counted by running
the recurrence.  Usage: python tools/experiments/r05_mom_precond_study.py [n] [nu_dt_over_h2] [hz_factor]"""
import sys

import numpy as np
import scipy.sparse as sp
import scipy.sparse.linalg as spla

sys.path.insert(0, ".")
from oracle import fluca_oracle as fo  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 32
ratio = float(sys.argv[2]) if len(sys.argv) > 2 else 2.56
hzf = float(sys.argv[3]) if len(sys.argv) > 3 else 1.0       # box height in z relative to x, y (0.5 = the bench's cavity box)
bc = [1, 2, 1, 1, 3, 3]
g = fo.Grid.uniform((n, n, n), [(0, 1), (0, 1), (0, hzf)], bc, 1e-3)
h = 1.0 / n
dt = 0.5 * h                     # CFL 0.5 on the unit inflow
mu = ratio * h * h / dt          # nu dt / h^2 = ratio (rho = 1)
N = g.ncell
yc = (np.arange(n) + 0.5) * h
u = np.broadcast_to((4 * yc * (1 - yc))[None, :, None], (n, n, n)).ravel()
v0 = np.concatenate([u, np.zeros(N), np.zeros(N)])
W = g.apply_B(v0)
V0 = g.apply_T(v0)
A = g.assemble_momentum(1.0, dt, -0.5 * mu * dt, V0, W).to_scipy().tocsr()
print(f"n = {n}^3, nu dt / h^2 = {ratio}, hz = {hzf} h, rows {A.shape[0]}, nnz {A.nnz}")
D = A.diagonal()
rows, cols = A.nonzero()
vals = np.asarray(A[rows, cols]).ravel()
comp_r, comp_c = rows // N, cols // N
cell_r, cell_c = rows % N, cols % N
same = comp_r == comp_c
off = cell_c - cell_r


def line_matrix(strides):
    keep = same & np.isin(off, [0] + [s for st in strides for s in (st, -st)] + [w for st, wrap in strides_wrap(strides) for w in wrap])
    return sp.csc_matrix((vals[keep], (rows[keep], cols[keep])), shape=A.shape)


def strides_wrap(strides):
    out = []
    for st in strides:
        if st == n * n:      # periodic z: the wrap entries of a z line
            out.append((st, [st * (n - 1), -st * (n - 1)]))
        else:
            out.append((st, []))
    return out


def extreme(apply_minv):
    op = spla.LinearOperator(A.shape, matvec=lambda x: apply_minv(A @ x), dtype=float)
    lmax = spla.eigs(op, k=1, which="LM", return_eigenvectors=False, tol=1e-6)[0]
    sh = spla.LinearOperator(A.shape, matvec=lambda x: lmax.real * x - apply_minv(A @ x), dtype=float)
    lmin = lmax.real - spla.eigs(sh, k=1, which="LM", return_eigenvectors=False, tol=1e-6)[0].real
    return lmin, lmax.real


def cheb_steps(apply_minv, emin, emax, tol=1e-5, maxit=400):
    rng = np.random.default_rng(3)
    b = rng.standard_normal(A.shape[0])
    x = np.zeros_like(b)
    z0 = apply_minv(b)
    ref = np.linalg.norm(z0)
    # PETSc's KSPCHEBYSHEV three-term form (oracle/fluca_oracle.c): first step p = scale * z, then the recurrence
    scale = 2.0 / (emax + emin)
    alpha = 1.0 - scale * emin
    mu_ = 1.0 / alpha
    omegaprod = 2.0 / alpha
    c_km1, c_k = 1.0, mu_
    xm = x.copy()
    x = scale * z0
    for it in range(1, maxit):
        z = apply_minv(b - A @ x)
        if np.linalg.norm(z) <= tol * ref:
            return it
        c_kp1 = 2.0 * mu_ * c_k - c_km1
        omega = omegaprod * c_k / c_kp1
        xn = (1.0 - omega) * xm + omega * (x + scale * z)
        xm, x = x, xn
        c_km1, c_k = c_k, c_kp1
    return maxit


cases = {"jacobi": lambda x: x / D}
for name, st in (("z-line", [n * n]), ("y-line", [n]), ("x-line", [1])):
    lu = spla.splu(line_matrix([st[0]]))
    cases[name] = lu.solve
lus = [spla.splu(line_matrix([s])) for s in (1, n, n * n)]
cases["ADI (x-line, y-line, z-line solves chained through D)"] = lambda x: lus[2].solve(D * lus[1].solve(D * lus[0].solve(x)))
for name, minv in cases.items():
    lmin, lmax = extreme(minv)
    kappa = lmax / lmin
    pred = np.log(2 / 1e-5) / np.log((np.sqrt(kappa) + 1) / (np.sqrt(kappa) - 1))
    steps = cheb_steps(minv, lmin, lmax)
    print(f"{name:58s} lambda in [{lmin:.4f}, {lmax:.4f}]  kappa {kappa:6.2f}  Chebyshev steps to 1e-5: predicted {pred:5.1f}, counted {steps}")
