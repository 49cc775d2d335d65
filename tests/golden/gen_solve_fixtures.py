#!/usr/bin/env python3
"""Generates tests/golden/solve_fixtures.json from the CPU oracle (oracle/fluca_oracle.c).

PARITY UNPINNED: these are NOT outputs of the reference (its solve runs inside PETSc, absent here); they freeze the
oracle's own Krylov behaviour so that (a) the oracle cannot drift silently and (b) the GPU path has a committed target
that does not depend on the host it is tested on.  Problem: SURVEY 8c manufactured solve, p = cos(pi x) cos(pi y)
cos(2 pi z) on the cavity_flow_3d box [0,1]^2 x [0,1/2], b = S p, kappa = 1e-3, cavity BCs
(fluca/tests/cavity_flow/cavity_flow_3d.c:42,72-77), PETSc-default tolerances unless stated.

usage: python tests/golden/gen_solve_fixtures.py   (single-threaded for a reproducible summation order)
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


def manufactured(g):
    xc = [0.5 * (a[1:] + a[:-1]) for a in g.xf]
    Z, Y, X = np.meshgrid(xc[2], xc[1], xc[0], indexing="ij")
    p = (np.cos(np.pi * X) * np.cos(np.pi * Y) * np.cos(2 * np.pi * Z)).ravel()
    return p - p.mean()


def main():
    out = []
    for n in (16, 32, 64):
        for ksp, name, kw in ((fo.KSP_CG, "cg", dict(rtol=1e-5)), (fo.KSP_BCGS, "bcgs", dict(rtol=1e-5)),
                              (fo.KSP_CHEBYSHEV, "chebyshev", dict(rtol=1e-2, maxit=200, emin=0.2, emax=2.2))):
            g = fo.Grid.uniform((n, n, n), BOX, CAVITY, 1e-3)
            S = g.assemble_S()
            p = manufactured(g)
            b = S.mult(p)
            x, info = S.solve(b, ksp=ksp, **kw)
            x -= x.mean()
            out.append(dict(n=n, ksp=name, kappa=1e-3, bc=CAVITY, opts=kw, iters=info["iters"], reason=info["reason"],
                            rnorm0=info["rnorm0"], history=[float(v) for v in info["history"]],
                            err_inf=float(abs(x - p).max()), b_norm=float(np.linalg.norm(b))))
            print(n, name, info["iters"], info["reason"], out[-1]["err_inf"])
    with open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "solve_fixtures.json"), "w") as fh:
        json.dump(out, fh, indent=0)


if __name__ == "__main__":
    main()
