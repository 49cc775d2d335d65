#!/usr/bin/env python3
"""Iterations/s of every Krylov type at the BASELINE.json grid sizes + IBM kernel timings (config 4 scale).  GPU only."""
import ctypes as C, sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from fluca_amd import capi
from fluca_amd.poisson import Poisson

ALGO = {0: 88, 1: 160, 2: 40}
NAME = {0: "cg+jacobi", 1: "bcgs+jacobi", 2: "chebyshev+jacobi"}
sizes = [int(a) for a in sys.argv[1:]] or [64, 256, 512]
for n in sizes:
    for bcname, bc in (("cavity", [1, 1, 1, 1, 4, 1]), ("channel", [1, 2, 1, 1, 3, 3])):
        P = Poisson.uniform((n, n, n), [(0, 1), (0, 1), (0, 0.5)], bc, 1e-3)
        g = torch.Generator(device="cuda").manual_seed(1)
        p = torch.rand(P.ncell, generator=g, dtype=torch.float64, device="cuda") * 2 - 1
        b = P.apply(p)
        x = P.empty()
        for ksp in (0, 1, 2):
            its = 200 if n >= 256 else 1000
            kw = dict(type=ksp, rtol=0.0, atol=0.0, maxit=its, remove_nullspace=int(2 not in bc), check_every=64)
            if ksp == 2:
                kw["norm_type"] = 3
            P.solve(b, x=x, **{**kw, "maxit": 20})
            torch.cuda.synchronize()
            _, info = P.solve(b, x=x, **kw)
            torch.cuda.synchronize()
            rate = info["iters"] / info["seconds"]
            print(f"n={n:4d} {bcname:8s} {NAME[ksp]:18s} {info['iters']:5d} its  {rate:9.1f} it/s  {1e3 / rate:8.4f} ms/it  "
                  f"algorithmic {ALGO[ksp] * P.ncell * rate / 1e9:8.1f} GB/s", flush=True)
        P.close()

# IBM: sphere D = 64 h at the centre of a 512^3 unit cube grid, Fibonacci lattice with spacing ~ h
n = 512 if 512 in sizes else max(sizes)
P = Poisson.uniform((n, n, n), [(0, 1), (0, 1), (0, 1)], [1] * 6, 1e-3)
h = 1.0 / n
R = 32 * h * (n / 512)
L = int(round(4 * np.pi * R * R / (h * h)))
i = np.arange(L) + 0.5
phi = np.arccos(1 - 2 * i / L); th = np.pi * (1 + 5 ** 0.5) * i
X = [0.5 + R * np.cos(th) * np.sin(phi), 0.5 + R * np.sin(th) * np.sin(phi), 0.5 + R * np.cos(phi)]
Xd = [torch.as_tensor(a, device="cuda") for a in X]
u = torch.rand(3 * P.ncell, dtype=torch.float64, device="cuda")
F = torch.rand(3 * L, dtype=torch.float64, device="cuda")
dV = torch.full((L,), h ** 3, dtype=torch.float64, device="cuda")
U = torch.empty(3 * L, dtype=torch.float64, device="cuda")
f = torch.zeros(3 * P.ncell, dtype=torch.float64, device="cuda")
m = C.c_void_p()
ptr = lambda t: C.c_void_p(t.data_ptr())
torch.cuda.synchronize()
for kind in (0, 1):
    t0 = time.perf_counter()
    capi.check(capi.lib.fl_ibm_create(P.h, kind, L, ptr(Xd[0]), ptr(Xd[1]), ptr(Xd[2]), C.byref(m)))
    P.synchronize(); t_create = time.perf_counter() - t0
    for name, call in (("interp", lambda: capi.lib.fl_ibm_interp(m, 3, ptr(u), ptr(U))), ("spread", lambda: capi.lib.fl_ibm_spread(m, 3, ptr(F), ptr(dV), ptr(f))),
                       ("rebin", lambda: capi.lib.fl_ibm_update(m, ptr(Xd[0]), ptr(Xd[1]), ptr(Xd[2])))):
        call(); P.synchronize()
        t0 = time.perf_counter()
        for _ in range(20):
            call()
        P.synchronize()
        dt = (time.perf_counter() - t0) / 20
        print(f"IBM kind={kind} L={L} grid {n}^3: {name:7s} {dt * 1e6:9.1f} us", flush=True)
    capi.lib.fl_ibm_destroy(m)
P.close()
