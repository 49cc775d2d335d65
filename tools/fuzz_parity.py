"""Random small grids / boundary types / stretching: HIP path against the oracle for apply, rhs, projection, CG, BiCGStab, Chebyshev.
Prints every case that deviates; the interesting ones go into tests/ as named cases.  usage: python tools/fuzz_parity.py [seed] [cases]"""
import os
import sys
import traceback

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np

from oracle import fluca_oracle as fo
from tests.gpu_common import O, PER, SYM, V, dev, host, make_pair

seed = int(sys.argv[1]) if len(sys.argv) > 1 else 1
ncase = int(sys.argv[2]) if len(sys.argv) > 2 else 60
rng = np.random.default_rng(seed)
SIZES = [1, 2, 3, 4, 5, 7, 8, 9, 15, 16, 17, 31, 33, 64, 65, 127, 129, 130, 257]
bad = 0
for case in range(ncase):
    n = tuple(int(rng.choice(SIZES[: (len(SIZES) if d == 0 else 14)])) for d in range(3))
    if n[0] * n[1] * n[2] > 400000:
        continue
    bc = []
    for d in range(3):
        kind = rng.integers(0, 4)
        if kind == 0:
            bc += [PER, PER]
        else:
            pair = [V, SYM, O]
            lo, hi = int(rng.choice(pair)), int(rng.choice(pair))
            if n[d] < 3:  # one-sided outlet rows need cells
                lo = V if lo == O else lo
                hi = V if hi == O else hi
            bc += [lo, hi]
    nonuni = bool(rng.integers(0, 2)) and min(n) >= 2
    singular = O not in bc
    tag = f"n={n} bc={bc} nonuni={nonuni}"
    try:
        P, g = make_pair(n, bc, kappa=float(rng.choice([1e-3, 0.5])), nonuniform=nonuni)
    except Exception as e:  # noqa: BLE001
        print("CREATE", tag, repr(e))
        continue
    try:
        S = g.assemble_S()
        p = rng.uniform(-1, 1, g.ncell)
        if singular:
            p -= p.mean()
        b = S.mult(p)
        y = host(P.apply(dev(p)))
        sc = max(np.abs(b).max(), 1e-300)
        if np.abs(y - b).max() > 1e-12 * sc:
            print("APPLY ", tag, np.abs(y - b).max() / sc); bad += 1
        Vf = [rng.standard_normal(nf) for nf in g.nface]
        r1 = host(P.rhs(*[dev(a) for a in Vf]))
        r0 = g.rhs(*Vf)
        if np.abs(r1 - r0).max() > 1e-12 * max(1.0, np.abs(r0).max()):
            print("RHS   ", tag, np.abs(r1 - r0).max()); bad += 1
        Vd = [dev(a) for a in Vf]
        P.project(dev(p), V=Vd)
        Gst = g.apply_gst(p)
        for d in range(3):
            ref = Vf[d] - Gst[d]
            if np.abs(host(Vd[d]) - ref).max() > 1e-12 * max(1.0, np.abs(ref).max()):
                print("PROJ  ", tag, d, np.abs(host(Vd[d]) - ref).max()); bad += 1
        # all six arrays at once (k_project_six on even rows) against the subset calls (k_project_all), bit for bit, and against the oracle
        vs = [rng.standard_normal(g.ncell) for _ in range(3)]
        v6, V6 = [dev(a) for a in vs], [dev(a) for a in Vf]
        P.project(dev(p), v=v6, V=V6)
        v3 = [dev(a) for a in vs]
        P.project(dev(p), v=v3)
        for d in range(3):
            if not np.array_equal(host(V6[d]), host(Vd[d])) or not np.array_equal(host(v6[d]), host(v3[d])):
                print("PROJ6 ", tag, d, np.abs(host(V6[d]) - host(Vd[d])).max(), np.abs(host(v6[d]) - host(v3[d])).max()); bad += 1
        if min(n) >= 3:
            Gc = g.apply_G(p)
            for d in range(3):
                ref = vs[d] - Gc[d]
                if np.abs(host(v6[d]) - ref).max() > 1e-12 * max(1.0, np.abs(ref).max()):
                    print("PROJ6v", tag, d, np.abs(host(v6[d]) - ref).max()); bad += 1
        symmetric = not nonuni
        for ksp, name in ((fo.KSP_CG, "CG"), (fo.KSP_BCGS, "BCGS")):
            if ksp == fo.KSP_CG and not symmetric:
                continue
            xo, io = S.solve(b, ksp=ksp, nullspace=singular, rtol=1e-7, maxit=400)
            xg, ig = P.solve(dev(b), type=ksp, remove_nullspace=int(singular), rtol=1e-7, maxit=400, history=True, check_every=5)
            m = min(len(ig["history"]), len(io["history"]), 4)
            hdev = np.abs(ig["history"][:m] / np.maximum(io["history"][:m], 1e-300) - 1).max() if m else 0.0
            tol_it = 2 if ksp == fo.KSP_CG else max(3, io["iters"] // 4)
            if ig["reason"] != io["reason"] or abs(ig["iters"] - io["iters"]) > tol_it or hdev > 1e-8:
                print(name.ljust(6), tag, "reason", ig["reason"], io["reason"], "iters", ig["iters"], io["iters"], "hist dev", hdev); bad += 1
        xo, io = S.solve(b, ksp=fo.KSP_CHEBYSHEV, nullspace=singular, norm=fo.NORM_NONE, maxit=7)
        xg, ig = P.solve(dev(b), type=2, remove_nullspace=int(singular), norm_type=fo.NORM_NONE, maxit=7)
        d = np.abs(host(xg) - xo).max() / max(np.abs(xo).max(), 1e-300)
        if d > 1e-9:
            print("CHEB  ", tag, d); bad += 1
    except Exception:  # noqa: BLE001
        print("EXC   ", tag)
        traceback.print_exc()
        bad += 1
    finally:
        P.close()
print("cases", ncase, "deviations", bad)
