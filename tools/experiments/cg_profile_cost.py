"""Does fl_ksp_opts.profile (four HIP events per iteration) cost iteration rate?  512^3, default variant, one box."""
import os
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import torch

from fluca_amd import poisson as flp

n = int(sys.argv[1]) if len(sys.argv) > 1 else 512
P = flp.Poisson.uniform((n, n, n), [(0, 1), (0, 1), (0, 0.5)], [1, 1, 1, 1, 4, 1], 1e-3)
p = torch.rand(P.ncell, dtype=torch.float64, device="cuda") * 2 - 1
p -= p.mean()
b = P.apply(p)
x = P.empty()
K = 100
for prof in (0, 1, 0, 1):
    kw = dict(rtol=0.0, atol=0.0, maxit=K, check_every=64, profile=prof)
    P.solve(b, x=x, **kw)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    _, info = P.solve(b, x=x, **kw)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print(f"n={n} profile={prof}: {K / dt:8.1f} it/s  {dt / K * 1e3:.4f} ms/it  device {info['seconds'] / K * 1e3:.4f} ms/it  A {info['kernel_ms']:.4f} Bq {info['kernel2_ms']:.4f}", flush=True)
print("placement probe (first, best):", P.tune_placement())
P.close()
