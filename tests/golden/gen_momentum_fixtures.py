#!/usr/bin/env python3
"""Generates tests/golden/momentum_fixtures.json from the CPU oracle (oracle/fluca_oracle.c, fluca_oracle.py).

PARITY UNPINNED for the Krylov parts, as in gen_solve_fixtures.py: these freeze the oracle's own numbers for the widened
rows (momentum block, multigrid) so that it cannot drift silently and the GPU path has a committed, host-independent
target.  Inputs are analytic (no random generator), so the fixture stores only results:
  cavity_flow_3d box and BCs, n^3/2... cells; V0_d = sin(2 pi x) cos(pi y) + 0.3 d, v0interp_{c,d} = cos(pi x) sin(2 pi z) + 0.1 (c - d),
  v_c = cos(pi x) cos(2 pi y) cos(pi z) (c + 1), b_c = sin(pi x) sin(pi y) + 0.2 c, all evaluated at the face / cell centres.

usage: python tests/golden/gen_momentum_fixtures.py   (single-threaded for a reproducible summation order)
"""
import json
import os
import sys

os.environ.setdefault("OMP_NUM_THREADS", "1")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np

from oracle import fluca_oracle as fo

BOX = [(0.0, 1.0), (0.0, 1.0), (0.0, 0.5)]
CAVITY = [1, 1, 1, 1, 4, 1]
DT, RHO, MU = 4e-3, 1.0, 0.02


def fields(g):
    """analytic inputs on the grid g: (V0[3], W[9], v, b)"""
    xc = [0.5 * (a[1:] + a[:-1]) for a in g.xf]

    def at(d):      # coordinates of the d-faces (or of the cells for d = None), arrays shaped (k, j, i)
        ax = [g.xf[a][:g.nf[a]] if a == d else xc[a] for a in range(3)]
        return np.meshgrid(ax[2], ax[1], ax[0], indexing="ij")[::-1]

    V0 = [(np.sin(2 * np.pi * at(d)[0]) * np.cos(np.pi * at(d)[1]) + 0.3 * d).ravel() for d in range(3)]
    W = [(np.cos(np.pi * at(d)[0]) * np.sin(2 * np.pi * at(d)[2]) + 0.1 * (c - d)).ravel() for c in range(3) for d in range(3)]
    X, Y, Z = at(None)
    v = np.concatenate([(np.cos(np.pi * X) * np.cos(2 * np.pi * Y) * np.cos(np.pi * Z) * (c + 1)).ravel() for c in range(3)])
    b = np.concatenate([(np.sin(np.pi * X) * np.sin(np.pi * Y) + 0.2 * c).ravel() for c in range(3)])
    return V0, W, v, b


def main():
    out = []
    for n in ((16, 12, 8), (32, 32, 16)):
        g = fo.Grid.uniform(n, BOX, CAVITY, DT / RHO)
        V0, W, v, b = fields(g)
        A = g.assemble_momentum(1.0, DT, -0.5 * MU * DT / RHO, V0, W)
        y = A.mult(v)
        x, info = A.solve(b, ksp=fo.KSP_BCGS, pc=fo.PC_JACOBI, nullspace=False, rtol=1e-10, maxit=200)
        # multigrid-PCG on S with the manufactured pressure of gen_solve_fixtures.py
        S = g.assemble_S()
        xc = [0.5 * (a[1:] + a[:-1]) for a in g.xf]
        Z, Y, X = np.meshgrid(xc[2], xc[1], xc[0], indexing="ij")
        p = (np.cos(np.pi * X) * np.cos(np.pi * Y) * np.cos(2 * np.pi * Z)).ravel()
        p -= p.mean()
        mg = fo.MgOracle(g)                      # piecewise-constant prolongation (the round-2 cycle)
        xm, im = mg.pcg(S.mult(p), rtol=1e-8, maxit=50)
        mgl = fo.MgOracle(g, prolong="linear")   # tri-linear prolongation (the library's default since round 3)
        xl, il = mgl.pcg(S.mult(p), rtol=1e-8, maxit=50)
        out.append(dict(n=list(n), dt=DT, rho=RHO, mu=MU, bc=CAVITY,
                        apply=dict(norm2=float(np.linalg.norm(y)), sum=float(y.sum()), absmax=float(np.abs(y).max()),
                                   samples=[float(y[i]) for i in (0, 7, len(y) // 3, len(y) // 2 + 5, len(y) - 1)]),
                        diag=dict(min=float(A.diag().min()), max=float(A.diag().max()), sum=float(A.diag().sum())),
                        bcgs=dict(iters=info["iters"], reason=info["reason"], history=[float(h) for h in info["history"]],
                                  x_norm2=float(np.linalg.norm(x))),
                        mg=dict(levels=mg.nlevels, bounds=[float(v) for v in mg.bounds], iters=im["iters"], reason=im["reason"],
                                history=[float(h) for h in im["history"]], err_inf=float(np.abs(xm - p).max())),
                        mg_linear=dict(iters=il["iters"], reason=il["reason"], history=[float(h) for h in il["history"]],
                                       err_inf=float(np.abs(xl - p).max()))))
        print(n, info["iters"], im["iters"], out[-1]["mg"]["err_inf"])
    with open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "momentum_fixtures.json"), "w") as fh:
        json.dump(out, fh, indent=0)


if __name__ == "__main__":
    main()
