"""Times the matrix-free momentum block (fl_momentum_apply / fl_momentum_solve) on one GPU.

usage: python tools/mom_bench.py [--cells 512] [--reps 10]
Algorithmic bytes of one application: 3 reads + 3 writes of the velocity + 12 face fields = 144 B per cell.
"""
import argparse
import json
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from fluca_amd.poisson import Momentum, Poisson  # noqa: E402

V, SYM = 1, 4


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--cells", type=int, default=512)
    ap.add_argument("--reps", type=int, default=10)
    ap.add_argument("--nosolve", action="store_true")
    ap.add_argument("--modes", type=int, default=4)
    ap.add_argument("--dif", type=float, default=2.56, help="nu dt / h^2 of the BiCGStab-against-Chebyshev comparison (512^3 at Re 100, CFL 0.5: 2.56); 0 skips it")
    ap.add_argument("--fly", type=int, default=1, help="1: v0interp = B v0, state handed over with v0 (k_mom3); 0: random v0interp fields, stored path (k_mom2)")
    a = ap.parse_args()
    n = (a.cells,) * 3
    P = Poisson.uniform(n, [(0, 1)] * 3, [V, V, V, V, SYM, V], 1e-3)
    M = Momentum(P)
    g = torch.Generator(device="cuda").manual_seed(1)
    rnd = lambda m: torch.rand(m, dtype=torch.float64, device="cuda", generator=g) * 2 - 1
    V0 = [rnd(P.nface[d]) for d in range(3)]
    h = 1.0 / a.cells
    if a.fly:
        v0 = rnd(3 * P.ncell)
        W = M.interp_faces(v0)
        M.set_state(0.5 * h, 1.0, 0.5 * h, V0, W, v0=v0)
        del v0
    else:
        W = [rnd(P.nface[d]) for c in range(3) for d in range(3)]
        M.set_state(0.5 * h, 1.0, 0.5 * h, V0, W)
    del V0, W
    v = rnd(3 * P.ncell)
    y = torch.empty_like(v)
    M.apply(v, y)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(a.reps):
        M.apply(v, y)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / a.reps
    out = dict(cells=a.cells, apply_ms_incl_pad=ms)
    if not a.nosolve:
      x, info = M.solve(v, rtol=1e-8, maxit=200)
      x, info = M.solve(v, rtol=1e-8, maxit=200)
      out.update(solve_iters=info["iters"], solve_reason=info["reason"], solve_ms=info["seconds"] * 1e3,
               ms_per_iter=info["seconds"] * 1e3 / max(info["iters"], 1))
      # KSPCHEBYSHEV on the same system with a viscous part of nu dt / h^2 = a.dif (the bench state above has 0.25: convection-dominated)
      if a.dif > 0:
        M.set_coefficients(1.0, 0.5 * h, -0.5 * a.dif * h * h)   # cI, cC = dt, cL = -mu dt / (2 rho)
        out["gershgorin_radius"] = M.gershgorin()
        for name, kw in (("bcgs", dict(type=1)), ("chebyshev", dict(type=2))):
            M.solve(v, rtol=1e-5, maxit=400, **kw)
            x, info = M.solve(v, rtol=1e-5, maxit=400, **kw)
            out[name + "_rtol1e-5"] = dict(iters=info["iters"], reason=info["reason"], ms=info["seconds"] * 1e3, ms_per_iter=info["seconds"] * 1e3 / max(info["iters"], 1))
        x, info = M.solve(v, type=2, norm_type=3, maxit=40)
        x, info = M.solve(v, type=2, norm_type=3, maxit=40)
        out["chebyshev_40_steps_no_norm_ms_per_step"] = info["seconds"] * 1e3 / 40
    out["env"] = {k: v for k, v in os.environ.items() if k.startswith("FLUCA_")}
    import ctypes as C
    from fluca_amd.capi import lib
    # the operator kernel alone on padded vectors: mode bit 0 Jacobi, bit 1 inner products
    lib.fldbg_mom_apply.argtypes = [C.c_void_p, C.c_int, C.c_int, C.POINTER(C.c_double)]
    for mode in range(a.modes):
        ms = C.c_double()
        P._pre()
        rc = lib.fldbg_mom_apply(M.h, mode, a.reps, C.byref(ms))
        P._post()
        assert rc == 0, rc
        out[f"kernel_ms_mode{mode}"] = ms.value
    out["kernel"] = os.environ.get("FLUCA_MOM_KERNEL", "3" if a.fly else "2")
    # streaming ceiling of the same access mix (15 reads + 3 writes, flat)
    if not hasattr(lib, "fldbg_mom_stream"):   # the probe lives in the kbench build of the library only (FLUCA_LIB_DIR=fluca_amd/lib_kbench)
        print(json.dumps(out))
        return
    lib.fldbg_mom_stream.argtypes = [C.c_void_p, C.c_int, C.c_int, C.POINTER(C.c_double)]
    for blocks in (2048, 8192):
        ms = C.c_double()
        P._pre()
        lib.fldbg_mom_stream(M.h, 5, blocks, C.byref(ms))
        P._post()
        out[f"stream15r3w_ms_{blocks}"] = ms.value
    print(json.dumps(out))


if __name__ == "__main__":
    main()
