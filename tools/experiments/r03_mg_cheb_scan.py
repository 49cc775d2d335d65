"""Multigrid-PCG at 512^3: iterations and residual history against the smoother's Chebyshev interval (fractions of the Gershgorin bound;
the library's default is PETSc's 0.1 / 1.1).  One process per setting (the knob is read once).  usage: python tools/experiments/r03_mg_cheb_scan.py"""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
CHILD = r'''
import sys, json, torch
sys.path.insert(0, %r)
from fluca_amd.poisson import Poisson
out = {}
for name, bc, box in (("cavity", [1, 1, 1, 1, 4, 1], [(0, 1), (0, 1), (0, 0.5)]), ("channel", [1, 2, 1, 1, 3, 3], [(0, 1), (0, 1), (0, 0.5)])):
    P = Poisson.uniform((512,) * 3, box, bc, 1e-3)
    g = torch.Generator(device="cuda").manual_seed(1)
    p = torch.rand(P.ncell, dtype=torch.float64, device="cuda", generator=g) * 2 - 1
    p -= p.mean()
    b = P.apply(p)
    P.solve(b, type=0, pc=2, rtol=1e-8, maxit=60)
    x, info = P.solve(b, type=0, pc=2, rtol=1e-8, maxit=60, history=True)
    h = info["history"]
    out[name] = dict(iters=info["iters"], seconds=round(info["seconds"], 4), reason=info["reason"], last=[h[-2] / h[0], h[-1] / h[0]])
    P.close()
print(json.dumps(out))
''' % ROOT
for lo, hi in ((0.1, 1.1), (0.15, 1.1), (0.2, 1.1), (0.1, 1.0), (0.15, 1.0), (0.2, 1.0), (0.25, 1.0), (0.3, 1.05), (0.125, 1.05), (0.08, 1.1)):
    env = dict(os.environ, FLUCA_MG_CHEB_LO=str(lo), FLUCA_MG_CHEB_HI=str(hi))
    r = subprocess.run([sys.executable, "-c", CHILD], env=env, capture_output=True, text=True, timeout=300)
    print(lo, hi, r.stdout.strip() or r.stderr[-300:], flush=True)
