#!/usr/bin/env python3
"""Chebyshev-Jacobi steps/s on BASELINE config 3 (512^3 channel: inlet / outlet in x, walls in y, periodic span), fixed-length
sweeps without a norm, with one step per launch ("cheb_fuse" 0) and two steps per sweep over memory (fl_cheb2.hip).  GPU only.
usage: cheb_bench.py [n] [steps]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from fluca_amd import capi
from fluca_amd.poisson import Poisson

n = int(sys.argv[1]) if len(sys.argv) > 1 else 512
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 100
bc = [1, 2, 1, 1, 3, 3]
P = Poisson.uniform((n, n, n), [(0, 1), (0, 1), (0, 0.5)], bc, 1e-3)
g = torch.Generator(device="cuda").manual_seed(1)
p = torch.rand(P.ncell, generator=g, dtype=torch.float64, device="cuda") * 2 - 1
b = P.apply(p)
x = P.empty()
ref = None
for mode in (0, 2, 0, 2):
    capi.check(capi.lib.fl_tuning_set(b"cheb_fuse", mode))
    kw = dict(type=2, norm_type=3, remove_nullspace=0, maxit=steps, check_every=steps, pc=int(os.environ.get("CHEB_PC", "1")))
    P.solve(b, x=x, **{**kw, "maxit": 10})
    torch.cuda.synchronize()
    _, info = P.solve(b, x=x, **kw)
    torch.cuda.synchronize()
    rate = info["iters"] / info["seconds"]
    if ref is None:
        ref = x.clone()
    err = float((x - ref).norm() / ref.norm())
    print(f"n={n} channel chebyshev+jacobi fuse={mode} {info['iters']:4d} steps {rate:9.1f} steps/s {1e3 / rate:8.4f} ms/step "
          f"algorithmic {40 * P.ncell * rate / 1e9:8.1f} GB/s  |x - x_fuse0| / |x| = {err:.2e}", flush=True)
P.close()
