#!/usr/bin/env python3
"""IBM kernels at BASELINE config 4 / config 5 scale on one GPU, and what marker REPLICATION costs a rank of config 5.

usage: python tools/ibm_bench.py [--loopback 1]
For each marker count L: re-binning (fl_ibm_update: count -> scan -> fill -> per-bin sort), interpolation, spreading of three components on a
512^3 block; with --loopback 1 the block has the multi-rank code path switched on with the rank as its own neighbour
(FLUCA_COMM_LOOPBACK=1), so interpolation ends with the all-reduce of U (3 L doubles) that replicated markers need.
L = 12 868: the sphere of config 4; 51 456: the cylinder markers inside ONE rank's 256-plane span of config 5 (what an owner-rank
scheme would hand each rank, to within the halo); 102 944: all of config 5's markers (what replication hands every rank); then a sweep.
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--loopback", type=int, default=0)
    ap.add_argument("--cells", type=int, default=512)
    ap.add_argument("--markers", type=str, default="12868,51456,102944,205888,411776,1647104")
    ap.add_argument("--sphere", type=int, default=0, help="1: the markers of config 4 (Fibonacci lattice on a sphere of diameter 64 h, spacing ~ h) instead of cylinders")
    a = ap.parse_args()
    os.environ["FLUCA_COMM_LOOPBACK"] = "1" if a.loopback else "0"
    from fluca_amd import capi
    from fluca_amd import poisson as flp
    n = a.cells
    bc = [1, 2, 1, 1, 3, 3] if a.loopback else [1, 2, 1, 1, 3, 3]
    P = flp.Poisson.uniform((n, n, n), [(0, 1)] * 3, bc, 1e-3)
    if a.loopback:
        P.comm_init_rccl(flp.rccl_unique_id(), 0, 1)
    h = 1.0 / n
    u = torch.rand(3 * P.ncell, dtype=torch.float64, device="cuda")
    f = torch.zeros(3 * P.ncell, dtype=torch.float64, device="cuda")
    ptr = lambda t: C.c_void_p(t.data_ptr())  # noqa: E731
    rows = []
    for L in [int(v) for v in a.markers.split(",")]:
        # markers on cylinders of radius 32 h around z (as many as L needs), spacing ~ h
        nth = int(round(2 * np.pi * 32))
        m = np.arange(L)
        th = (m % nth + 0.5) * 2 * np.pi / nth
        ring = m // nth
        rad = (32 + 3 * (ring // n)) * h
        X = [torch.as_tensor(v, device="cuda") for v in (0.5 + rad * np.cos(th), 0.5 + rad * np.sin(th), ((ring % n) + 0.5) * h)]
        if a.sphere:
            R = 32 * h
            L = int(round(4 * np.pi * R * R / (h * h)))
            i = np.arange(L) + 0.5
            phi, th = np.arccos(1 - 2 * i / L), np.pi * (1 + 5 ** 0.5) * i
            X = [torch.as_tensor(v, device="cuda") for v in (0.5 + R * np.cos(th) * np.sin(phi), 0.5 + R * np.sin(th) * np.sin(phi), 0.5 + R * np.cos(phi))]
        F = torch.rand(3 * L, dtype=torch.float64, device="cuda")
        dV = torch.full((L,), h ** 3, dtype=torch.float64, device="cuda")
        U = torch.empty(3 * L, dtype=torch.float64, device="cuda")
        hm = C.c_void_p()
        torch.cuda.synchronize()
        capi.check(capi.lib.fl_ibm_create(P.h, capi.DELTA_PESKIN4, L, ptr(X[0]), ptr(X[1]), ptr(X[2]), C.byref(hm)), "fl_ibm_create")

        def timed(fn, reps=20):
            fn()
            P.synchronize()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(reps):
                fn()
            P.synchronize()
            torch.cuda.synchronize()
            return (time.perf_counter() - t0) / reps * 1e3

        t_bin = timed(lambda: capi.check(capi.lib.fl_ibm_update(hm, ptr(X[0]), ptr(X[1]), ptr(X[2]))))
        t_int = timed(lambda: capi.check(capi.lib.fl_ibm_interp(hm, 3, ptr(u), ptr(U))))
        t_spr = timed(lambda: capi.check(capi.lib.fl_ibm_spread(hm, 3, ptr(F), ptr(dV), ptr(f))))
        st = [C.c_int(), C.c_int(), C.c_int()]
        if hasattr(capi.lib, "fldbg_ibm_stats"):   # bin statistics: kbench build of the library only (FLUCA_LIB_DIR=fluca_amd/lib_kbench)
            capi.lib.fldbg_ibm_stats.argtypes = [C.c_void_p] + [C.POINTER(C.c_int)] * 3
            capi.check(capi.lib.fldbg_ibm_stats(hm, *[C.byref(v) for v in st]))
        rows.append(dict(markers=L, rebin_ms=t_bin, interp_ms=t_int, spread_ms=t_spr, allreduce_bytes=24 * L if a.loopback else 0, tiles_with_markers=st[0].value,
                         bin_entries=st[1].value, largest_bin=st[2].value, env={k: v for k, v in os.environ.items() if k.startswith("FLUCA_IBM")}))
        capi.lib.fl_ibm_destroy(hm)
        print(json.dumps(rows[-1]), flush=True)
    P.close()


if __name__ == "__main__":
    main()
