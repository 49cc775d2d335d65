import sys, numpy as np, torch
sys.path.insert(0, '/root/repo')
from tests.gpu_common import make_pair, mean_free_rhs, dev, host
from oracle import fluca_oracle as fo
V, PER = fo.BC_VELOCITY, fo.BC_PERIODIC
for n, bc in (((1, 4, 4), [PER, PER, V, V, V, V]), ((4, 1, 4), [V, V, PER, PER, V, V]), ((4, 4, 1), [V, V, V, V, PER, PER]), ((1, 4, 4), [V] * 6)):
    P, g = make_pair(n, bc, kappa=1e-3)
    S = g.assemble_S()
    p, b = mean_free_rhs(S, g.ncell)
    y = host(P.apply(dev(p)))
    print(n, bc[:2], "apply err", np.abs(y - b).max() / np.abs(b).max())
    xo, io = S.solve(b, rtol=1e-8, maxit=50)
    for variant in (1, 2, 0):
        xg, ig = P.solve(dev(b), rtol=1e-8, maxit=50, variant=variant, history=True)
        m = min(len(ig["history"]), len(io["history"]), 5)
        print("  variant", variant, "iters", ig["iters"], io["iters"], "reason", ig["reason"], io["reason"], "hist", ig["history"][:m], io["history"][:m])
    P.close()
