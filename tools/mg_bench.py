"""Time to solution of the pressure-Poisson solve: Jacobi-PCG vs multigrid-PCG (FL_PC_MG).

usage: python tools/mg_bench.py [--cells 512] [--rtol 1e-8]
"""
import argparse
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from fluca_amd.poisson import Poisson  # noqa: E402

V, SYM = 1, 4


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--cells", type=int, default=512)
    ap.add_argument("--rtol", type=float, default=1e-8)
    ap.add_argument("--smooth", type=int, default=0)
    ap.add_argument("--jacobi-rtol", type=float, default=None, help="tighter tolerance for the Jacobi run (to match the error of the MG run)")
    ap.add_argument("--skip-jacobi", action="store_true")
    ap.add_argument("--prolong", type=int, default=None, help="tuning knob mg_prolong: 0 piecewise constant, 1 tri-linear")
    a = ap.parse_args()
    n = (a.cells,) * 3
    if a.prolong is not None:
        from fluca_amd import capi
        capi.check(capi.lib.fl_tuning_set(b"mg_prolong", a.prolong))
    P = Poisson.uniform(n, [(0, 1), (0, 1), (0, 0.5)], [V, V, V, V, SYM, V], 1e-3)
    g = torch.Generator(device="cuda").manual_seed(1)
    p = torch.rand(P.ncell, dtype=torch.float64, device="cuda", generator=g) * 2 - 1
    p -= p.mean()
    b = P.apply(p)
    out = dict(cells=a.cells, rtol=a.rtol, prolong=a.prolong, smooth=a.smooth)
    for name, kw in (("mg", dict(pc=2, maxit=200, mg_smooth_its=a.smooth)), ("jacobi", dict(pc=1, maxit=20000))):
        if name == "jacobi" and a.skip_jacobi:
            continue
        rtol = a.jacobi_rtol if (name == "jacobi" and a.jacobi_rtol) else a.rtol
        x, info = P.solve(b, type=0, rtol=rtol, **kw)
        x, info = P.solve(b, type=0, rtol=rtol, **kw)
        err = float(torch.linalg.norm((x - x.mean()) - p) / torch.linalg.norm(p))
        out[name] = dict(rtol=rtol, iters=info["iters"], reason=info["reason"], seconds=info["seconds"], rel_error=err)
    if "jacobi" in out:
        out["speedup"] = out["jacobi"]["seconds"] / out["mg"]["seconds"]
    print(json.dumps(out))


if __name__ == "__main__":
    main()
