import sys, numpy as np
sys.path.insert(0, '.')
from oracle import fluca_oracle as fo
from tests.gpu_common import CAVITY, dev, host
from tests.test_gpu_momentum import _pair, _fields
from fluca_amd import capi
n, bc = (20, 9, 6), CAVITY
P, M, g = _pair(n, bc, True)
V0, W = _fields(g)
dt, rho, mu = 0.05, 1.3, 0.02
M.set_state(dt, rho, mu, [dev(a) for a in V0], [dev(a) for a in W])
A = g.assemble_momentum(1.0, dt, -0.5 * mu * dt / rho, V0, W)
b = np.random.default_rng(9).standard_normal(A.nrow)
xo, io = fo.gmres(A, b, pc=1, rtol=1e-8, restart=30, maxit=300)
xg, ig = M.solve(dev(b), history=True, type=capi.KSP_GMRES, pc=1, rtol=1e-8, maxit=40, gmres_restart=30)
print("oracle", io["iters"], io["reason"], io["history"][:8])
print("gpu   ", ig["iters"], ig["reason"], ig["history"][:8])
d = host(M.diagonal()); print("diag err", abs(d - A.diag()).max())
y = host(M.apply(dev(b))); print("apply err", abs(y - A.mult(b)).max())
