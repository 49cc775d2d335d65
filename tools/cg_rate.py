#!/usr/bin/env python3
"""Jacobi-PCG iteration rate on one grid size (BASELINE config 2 by default: 256^3 lid-driven cavity), a fixed number of iterations.

usage: python tools/cg_rate.py [--cells 256] [--iters 400] [--reps 3] [--single-reduction 0]
Prints one JSON line; FLUCA_* environment variables (plan overrides of the experiments) are echoed in it."""
import argparse
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from fluca_amd.poisson import Poisson  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--cells", type=int, default=256)
    ap.add_argument("--iters", type=int, default=400)
    ap.add_argument("--reps", type=int, default=3)
    ap.add_argument("--single-reduction", type=int, default=0)
    a = ap.parse_args()
    P = Poisson.uniform((a.cells,) * 3, [(0, 1), (0, 1), (0, 0.5)], [1, 1, 1, 1, 4, 1], 1e-3)
    g = torch.Generator(device="cuda").manual_seed(2)
    p = torch.rand(P.ncell, generator=g, dtype=torch.float64, device="cuda") * 2 - 1
    b, x = P.apply(p), P.empty()
    kw = dict(rtol=0.0, atol=0.0, check_every=64, cg_single_reduction=a.single_reduction)
    P.solve(b, x=x, maxit=50, **kw)
    ms = []
    for _ in range(a.reps):
        _, info = P.solve(b, x=x, maxit=a.iters, **kw)
        ms.append(info["seconds"] / a.iters * 1e3)
    P.close()
    best = min(ms)
    print(json.dumps(dict(cells=a.cells, iters=a.iters, ms_per_iter=best, its_per_s=1e3 / best, all_ms=ms,
                          env={k: v for k, v in os.environ.items() if k.startswith("FLUCA_")})), flush=True)


if __name__ == "__main__":
    main()
