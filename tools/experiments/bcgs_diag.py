import sys, json, numpy as np, torch
sys.path.insert(0, '/root/repo')
from tests import test_solve_fixtures as T
from fluca_amd.poisson import Poisson
for f in T.FIX:
    if f["ksp"] != "bcgs": continue
    g, p = T.problem(f)
    S = g.assemble_S()
    b = S.mult(p)
    P = Poisson.uniform((f["n"],) * 3, T.BOX, f["bc"], f["kappa"])
    for variant in (2, 0):
        x, info = P.solve(torch.as_tensor(b, device="cuda"), history=True, type=1, variant=variant, **f["opts"])
        x = x.cpu().numpy(); x -= x.mean()
        r = b - S.mult(x)
        print(f["n"], "variant", variant, "iters", info["iters"], "fixture", f["iters"], "rnorm", info["rnorm"], "true |r|/|b|", np.linalg.norm(r)/np.linalg.norm(b), "err", abs(x-p).max(), "fixture err", f["err_inf"])
    P.close()
