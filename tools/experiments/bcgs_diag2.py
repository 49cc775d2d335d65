import sys, numpy as np, torch
sys.path.insert(0, '/root/repo')
from tests.gpu_common import make_pair, mean_free_rhs, dev, host
from oracle import fluca_oracle as fo
V, PER = fo.BC_VELOCITY, fo.BC_PERIODIC
for n, bc in (((2, 257, 3), [V, V, PER, PER, V, V]), ((129, 3, 2), [V, V, V, V, fo.BC_SYMMETRY, V])):
    P, g = make_pair(n, bc, kappa=1e-3)
    S = g.assemble_S()
    _, b = mean_free_rhs(S, g.ncell)
    xo, io = S.solve(b, ksp=1, rtol=1e-8, maxit=500)
    ro = np.linalg.norm(b - S.mult(xo)) / np.linalg.norm(b)
    print(n, "oracle iters", io["iters"], "reason", io["reason"], "true res", ro)
    for variant in (2, 0):
        xg, ig = P.solve(dev(b), type=1, rtol=1e-8, maxit=500, check_every=3, variant=variant, history=True)
        xg = host(xg)
        rg = np.linalg.norm(b - S.mult(xg)) / np.linalg.norm(b)
        d = np.linalg.norm((xg - xg.mean()) - (xo - xo.mean())) / np.linalg.norm(xo)
        h = ig["history"]; ho = io["history"]
        m = min(len(h), len(ho), 6)
        print("  variant", variant, "iters", ig["iters"], "reason", ig["reason"], "true res", rg, "diff vs oracle", d, "hist rel dev", np.abs(h[:m] / ho[:m] - 1).max())
    P.close()
