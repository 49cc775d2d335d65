#!/usr/bin/env python3
"""Does the physical placement of the momentum block's vectors matter like it does for k_cg_A?  The same 512^3 momentum apply and
BiCGStab solve with (a) the library's back-to-back allocations and (b) spacer allocations of G GiB held between the allocation
steps (face fields | work vector of apply | the seven BiCGStab vectors).  One process, A/B/A.  GPU only."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from fluca_amd.poisson import Momentum, Poisson

V, SYM = 1, 4
n = (512,) * 3


def run(spacer_gib):
    keep = []

    def spacer():
        if spacer_gib > 0:
            keep.append(torch.empty(spacer_gib << 27, dtype=torch.float64, device="cuda"))
    P = Poisson.uniform(n, [(0, 1)] * 3, [V, V, V, V, SYM, V], 1e-3)
    M = Momentum(P)
    g = torch.Generator(device="cuda").manual_seed(1)
    rnd = lambda m: torch.rand(m, dtype=torch.float64, device="cuda", generator=g) * 2 - 1
    V0 = [rnd(P.nface[d]) for d in range(3)]
    W = [rnd(P.nface[d]) for c in range(3) for d in range(3)]
    h = 1.0 / n[0]
    spacer()
    M.set_state(0.5 * h, 1.0, 0.5 * h, V0, W)
    del V0, W
    spacer()
    v = rnd(3 * P.ncell)
    y = torch.empty_like(v)
    M.apply(v, y)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        M.apply(v, y)
    e1.record()
    torch.cuda.synchronize()
    out = dict(spacer_GiB=spacer_gib, apply_ms_incl_pad=e0.elapsed_time(e1) / 10)
    spacer()
    x, info = M.solve(v, rtol=1e-8, maxit=200)
    x, info = M.solve(v, rtol=1e-8, maxit=200)
    out.update(iters=info["iters"], ms_per_iter=info["seconds"] * 1e3 / max(info["iters"], 1))
    M.close()
    P.close()
    del keep, v, y, x
    torch.cuda.empty_cache()
    return out


for s in (0, 6, 0, 12, 20):
    print(json.dumps(run(s)), flush=True)
