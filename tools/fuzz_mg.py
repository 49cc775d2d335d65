"""Random grid sizes / boundary types / stretching: FL_PC_MG (fl_mg.hip) against the oracle's restatement of the same cycle.
usage: python tools/fuzz_mg.py [seed] [cases]"""
import ctypes as C
import os
import sys
import traceback

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np

from oracle import fluca_oracle as fo
from tests.gpu_common import O, PER, SYM, V, dev, host, make_pair

seed = int(sys.argv[1]) if len(sys.argv) > 1 else 1
ncase = int(sys.argv[2]) if len(sys.argv) > 2 else 16
rng = np.random.default_rng(seed)
SIZES = [6, 8, 10, 12, 15, 16, 18, 20, 24, 30, 32, 36, 40, 48]


def bounds(mg):
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


bad = 0
for case in range(ncase):
    n = tuple(int(rng.choice(SIZES)) for d in range(3))
    bc = []
    for d in range(3):
        if rng.integers(0, 3) == 0:
            bc += [PER, PER]
        else:
            bc += [int(rng.choice([V, SYM, O])), int(rng.choice([V, SYM, O]))]
    nonuni = bool(rng.integers(0, 2))
    singular = O not in bc
    tag = f"n={n} bc={bc} nonuni={nonuni}"
    try:
        P, g = make_pair(n, bc, kappa=1e-3, nonuniform=nonuni)
        S = g.assemble_S()
        p = rng.standard_normal(g.ncell)
        if singular:
            p -= p.mean()
        b = S.mult(p)
        mg = fo.MgOracle(g, nullspace=singular)
        mg = fo.MgOracle(g, nullspace=singular, bounds=bounds(mg), prolong="linear")  # the library's default since round 3 (knob mg_prolong = 1)
        xo, io = mg.pcg(b, rtol=1e-8, maxit=60)
        xg, ig = P.solve(dev(b), history=True, type=0, pc=2, remove_nullspace=int(singular), rtol=1e-8, maxit=60)
        xg = host(xg)
        res = np.linalg.norm(b - S.mult(xg)) / np.linalg.norm(b)
        m = min(len(ig["history"]), len(io["history"]), 3)
        hdev = np.abs(ig["history"][:m] / io["history"][:m] - 1).max()
        if ig["reason"] != io["reason"] or abs(ig["iters"] - io["iters"]) > 1 or hdev > 1e-5 or (ig["reason"] == 2 and res > 1e-6):
            print("MG    ", tag, "levels", len(mg.grids), "reason", ig["reason"], io["reason"], "iters", ig["iters"], io["iters"], "hist dev", hdev, "res", res); bad += 1
        P.close()
    except Exception:  # noqa: BLE001
        print("EXC   ", tag)
        traceback.print_exc()
        bad += 1
print("cases", ncase, "deviations", bad)
