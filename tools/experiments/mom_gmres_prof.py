"""KSPGMRES(30) + Jacobi on the momentum block at 512^3 (the reference's default type for kspA): one solve, for rocprofv3 --stats."""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import torch

from fluca_amd.poisson import Momentum, Poisson

n = int(sys.argv[1]) if len(sys.argv) > 1 else 512
P = Poisson.uniform((n,) * 3, [(0, 1)] * 3, [1, 1, 1, 1, 4, 1], 1e-3)
M = Momentum(P)
g = torch.Generator(device="cuda").manual_seed(1)
rnd = lambda m: torch.rand(m, dtype=torch.float64, device="cuda", generator=g) * 2 - 1
V0 = [rnd(P.nface[d]) for d in range(3)]
W = [rnd(P.nface[d]) for c in range(3) for d in range(3)]
h = 1.0 / n
M.set_state(0.5 * h, 1.0, 0.5 * h, V0, W)
del V0, W
v = rnd(3 * P.ncell)
for typ, name in ((3, "gmres"), (1, "bcgs")):
    x, info = M.solve(v, type=typ, rtol=1e-5, maxit=200)
    x, info = M.solve(v, type=typ, rtol=1e-5, maxit=200)
    print(name, "iters", info["iters"], "reason", info["reason"], "seconds", info["seconds"], "ms/iter", info["seconds"] * 1e3 / max(info["iters"], 1), flush=True)
