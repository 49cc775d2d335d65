"""Time fl_abf_schur_apply with schurainv = DIAG at N^3: the one-pass kernel (fl_schur_var.hip) against the composition of seven kernels, and the
two against each other.  usage: python tools/schur_var_bench.py [cells per axis, default 512]"""
import ctypes as C
import os
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
import torch

from fluca_amd import capi

from fluca_amd.poisson import Momentum, Poisson
from oracle import fluca_oracle as fo

n = int(sys.argv[1]) if len(sys.argv) > 1 else 512
V, O, PER = fo.BC_VELOCITY, fo.BC_PRESSURE_OUTLET, fo.BC_PERIODIC
bc = [V, O, V, V, PER, PER]
xf = [np.linspace(0.0, 1.0, n + 1) for _ in range(3)]
P = Poisson((n, n, n), xf, bc, 1e-3)
M = Momentum(P)
N = n ** 3
g = torch.Generator(device="cuda").manual_seed(3)
rnd = lambda m: torch.rand(m, dtype=torch.float64, device="cuda", generator=g) - 0.5
V0 = [8.0 * rnd(P.nface[d]) for d in range(3)]
v0 = 8.0 * rnd(3 * N)
M.set_state(1e-3, 1.0, 0.05, V0, M.interp_faces(v0), v0=v0)
M.set_ainv_types(schur=fo.AINV_DIAG)
p = rnd(N)
out = {}
for mode in (1, 0):
    capi.check(capi.lib.fl_tuning_set(b"schur_var_fused", mode))
    y = M.schur_apply(p)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(5):
        y = M.schur_apply(p)
    torch.cuda.synchronize()
    out[mode] = ((time.perf_counter() - t0) / 5, y.clone())
capi.check(capi.lib.fl_tuning_set(b"schur_var_fused", 1))
d = (out[1][1] - out[0][1]).abs().max().item() / out[0][1].abs().max().item()
print(f"cells {n}^3  one pass {out[1][0] * 1e3:.3f} ms  composition {out[0][0] * 1e3:.3f} ms  (each with diag(A) recomputed: fl_abf_schur_apply)  rel max diff {d:.2e}")
