"""Random small grids / boundary types / stretching: the matrix-free momentum block (apply, diagonal, BiCGStab and GMRES solves) and the
face interpolations against the oracle's assembled rows; every case also with the state handed over together with v0 (k_mom3: v0interp formed in
the kernel on inner faces, stored values with a random boundary-only vbc on the block-end faces).  usage: python tools/fuzz_momentum.py [seed] [cases]"""
import os
import sys
import traceback

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np

from oracle import fluca_oracle as fo
from tests.gpu_common import CAVITY_BOX, O, PER, SYM, V, dev, host, stretched

seed = int(sys.argv[1]) if len(sys.argv) > 1 else 1
ncase = int(sys.argv[2]) if len(sys.argv) > 2 else 40
rng = np.random.default_rng(seed)
SIZES = [2, 3, 4, 5, 7, 8, 9, 15, 16, 17, 33, 63, 64, 65, 66, 129]
bad = 0
for case in range(ncase):
    n = tuple(int(rng.choice(SIZES[: (len(SIZES) if d == 0 else (11 if d == 1 else 10))])) for d in range(3))
    if n[0] * n[1] * n[2] > 60000:
        continue
    bc = []
    for d in range(3):
        if rng.integers(0, 4) == 0:
            bc += [PER, PER]
        else:
            lo, hi = int(rng.choice([V, SYM, O])), int(rng.choice([V, SYM, O]))
            if n[d] < 3:
                lo = V if lo == O else lo
                hi = V if hi == O else hi
            bc += [lo, hi]
    nonuni = bool(rng.integers(0, 2))
    tag = f"n={n} bc={bc} nonuni={nonuni}"
    from fluca_amd.poisson import Momentum, Poisson
    try:
        if nonuni:
            xf = [stretched(n[d], CAVITY_BOX[d][0], CAVITY_BOX[d][1], 1.1 + 0.2 * d) for d in range(3)]
            P, g = Poisson(n, xf, bc, 1e-3), fo.Grid(n, xf, bc, 1e-3)
        else:
            P, g = Poisson.uniform(n, CAVITY_BOX, bc, 1e-3), fo.Grid.uniform(n, CAVITY_BOX, bc, 1e-3)
        M = Momentum(P)
    except Exception as e:  # noqa: BLE001
        print("CREATE", tag, repr(e))
        continue
    try:
        V0 = [rng.standard_normal(g.nface[d]) for d in range(3)]
        W = [rng.standard_normal(g.nface[d]) for c in range(3) for d in range(3)]
        dt, rho, mu = 0.013, 1.7, 0.031
        v = rng.standard_normal(3 * g.ncell)
        M.set_state(dt, rho, mu, [dev(a) for a in V0], [dev(a) for a in W])
        A = g.assemble_momentum(1.0, dt, -0.5 * mu * dt / rho, V0, W)
        want = A.mult(v)
        got = host(M.apply(dev(v)))
        if np.abs(got - want).max() > 2e-13 * np.abs(want).max():
            print("APPLY ", tag, np.abs(got - want).max() / np.abs(want).max()); bad += 1
        dg = host(M.diagonal())
        if np.abs(dg - A.diag()).max() > 2e-13 * np.abs(A.diag()).max():
            print("DIAG  ", tag, np.abs(dg - A.diag()).max()); bad += 1
        # the same operator from (V0, v0, vbc on boundary faces): fl_momentum_set_state_v0
        v0 = rng.standard_normal(3 * g.ncell)
        Wb = g.apply_B(v0)
        for c in range(3):
            for d in range(3):
                if not g.periodic[d]:
                    shape = [g.n[2], g.n[1], g.n[0]]
                    shape[2 - d] = g.nf[d]
                    a = Wb[c * 3 + d].reshape(shape)
                    for f in (0, g.n[d]):
                        sl = [slice(None)] * 3
                        sl[2 - d] = f
                        a[tuple(sl)] += rng.standard_normal(a[tuple(sl)].shape)
        M.set_state(dt, rho, mu, [dev(a) for a in V0], [dev(a) for a in Wb], v0=dev(v0))
        Ab = g.assemble_momentum(1.0, dt, -0.5 * mu * dt / rho, V0, Wb)
        want = Ab.mult(v)
        got = host(M.apply(dev(v)))
        if np.abs(got - want).max() > 2e-13 * np.abs(want).max():
            print("APPLY0", tag, np.abs(got - want).max() / np.abs(want).max()); bad += 1
        dg = host(M.diagonal())
        if np.abs(dg - Ab.diag()).max() > 2e-13 * np.abs(Ab.diag()).max():
            print("DIAG0 ", tag, np.abs(dg - Ab.diag()).max()); bad += 1
        xo0, io0 = Ab.solve(v, ksp=fo.KSP_BCGS, pc=fo.PC_JACOBI, nullspace=False, rtol=1e-9, maxit=300)
        xg0, ig0 = M.solve(dev(v), type=1, rtol=1e-9, maxit=300)
        if io0["reason"] > 0 and (ig0["reason"] != io0["reason"] or np.abs(host(xg0) - xo0).max() > 1e-5 * np.abs(xo0).max()):
            print("BCGS0 ", tag, "reason", ig0["reason"], io0["reason"], "iters", ig0["iters"], io0["iters"]); bad += 1
        # a well-conditioned system (small dt): both Krylov types to the oracle's answer
        dt2 = 1e-3
        M.set_state(dt2, 1.0, 0.01, [dev(a) for a in V0], [dev(a) for a in W])
        A2 = g.assemble_momentum(1.0, dt2, -0.5 * 0.01 * dt2, V0, W)
        b = rng.standard_normal(3 * g.ncell)
        xo, io = A2.solve(b, ksp=fo.KSP_BCGS, pc=fo.PC_JACOBI, nullspace=False, rtol=1e-9, maxit=200)
        for typ, name in ((1, "BCGS"), (3, "GMRES")):
            xg, ig = M.solve(dev(b), type=typ, rtol=1e-9, maxit=200)
            d = np.abs(host(xg) - xo).max() / np.abs(xo).max()
            if ig["reason"] != 2 or d > 1e-6:
                print(name.ljust(6), tag, "reason", ig["reason"], "iters", ig["iters"], io["iters"], "diff", d); bad += 1
        # round 4: the Gershgorin radius of D^-1 A from the row coefficients (k_mom2 here, k_mom3 below) and KSPCHEBYSHEV on the default interval
        for with_v0 in (False, True):
            if with_v0:
                M.set_state(dt2, 1.0, 0.01, [dev(a) for a in V0], [dev(a) for a in Wb], v0=dev(v0))
                Ac = g.assemble_momentum(1.0, dt2, -0.5 * 0.01 * dt2, V0, Wb)
            else:
                Ac = A2
            G, rad = Ac.gershgorin(fo.PC_JACOBI), M.gershgorin()
            # (a periodic axis of two cells folds two columns of a row into one matrix entry: the kernel's sum of absolute values is then an upper bound)
            two = any(g.periodic[d] and g.n[d] == 2 for d in range(3))
            if (abs(1.0 + rad - G) > 1e-11 * G) if not two else (1.0 + rad < G * (1 - 1e-11)):
                print("GERSH" + "0 "[with_v0], tag, 1.0 + rad, G); bad += 1
            emin, emax = M.chebyshev_interval()
            xo, io = Ac.solve(b, ksp=fo.KSP_CHEBYSHEV, pc=fo.PC_JACOBI, nullspace=False, rtol=1e-9, maxit=400, emin=emin, emax=emax)
            xg, ig = M.solve(dev(b), type=2, rtol=1e-9, maxit=400)
            d = np.abs(host(xg) - xo).max() / np.abs(xo).max()
            if ig["reason"] != io["reason"] or ig["iters"] != io["iters"] or d > 1e-7:
                print("CHEB" + "0 "[with_v0] + " ", tag, "reason", ig["reason"], io["reason"], "iters", ig["iters"], io["iters"], "diff", d, "interval", emin, emax); bad += 1
    except Exception:  # noqa: BLE001
        print("EXC   ", tag)
        traceback.print_exc()
        bad += 1
    finally:
        M.close()
        P.close()
print("cases", ncase, "deviations", bad)
