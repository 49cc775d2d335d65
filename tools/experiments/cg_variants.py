"""A/B of the CG kernel pairs: variant 0 (k_cg_A<SQ=false> + k_cg_Bq, q never stored) against variant 2 (k_cg_A + k_cg_B, q stored).
usage: python tools/experiments/cg_variants.py [n ...]   (run under rocprofv3 --kernel-trace --stats for per-kernel times)"""
import os
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import torch

from fluca_amd import poisson as flp

V, SYM = 1, 4
sizes = [int(a) for a in sys.argv[1:]] or [256, 512]
for n in sizes:
    P = flp.Poisson.uniform((n, n, n), [(0, 1), (0, 1), (0, 0.5)], [V, V, V, V, SYM, V], 1e-3)
    p = torch.rand(P.ncell, dtype=torch.float64, device="cuda") * 2 - 1
    p -= p.mean()
    b = P.apply(p)
    x = P.empty()
    K = 400 if n <= 256 else 100
    for variant in (2, 0, 2, 0):
        kw = dict(rtol=0.0, atol=0.0, maxit=K, variant=variant, check_every=64)
        P.solve(b, x=x, **kw)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        _, info = P.solve(b, x=x, **kw)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        print(f"n={n} variant={variant}: {K / dt:8.1f} it/s  {dt / K * 1e3:.4f} ms/it  rnorm {info['rnorm']:.6e}", flush=True)
    P.close()
