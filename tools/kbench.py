#!/usr/bin/env python3
"""A/B timing of the hot kernels in one process (rule: interleaved rounds, median and min).  GPU only.

usage: python tools/kbench.py [n] [rounds]
"""
import ctypes as C
import statistics
import sys
import os

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from fluca_amd import capi
from fluca_amd.poisson import Poisson

n = int(sys.argv[1]) if len(sys.argv) > 1 else 512
rounds = int(sys.argv[2]) if len(sys.argv) > 2 else 5
bc = [1, 1, 1, 1, 4, 1]
P = Poisson.uniform((n, n, n), [(0, 1), (0, 1), (0, 0.5)], bc, 1e-3)
src = torch.rand(P.ncell, dtype=torch.float64, device="cuda") - 0.5
torch.cuda.synchronize()
f = capi.lib.fldbg_bench
f.restype = C.c_int
f.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.POINTER(C.c_double), C.POINTER(C.c_int)]

cases = []
if os.environ.get("KB_TALL"):  # round 3: 128 x 32 tiles (four rows per wave) with the two register sets, against the shipped 128 x 16
    for rd_ in range(1):
        cases.append(("A", 0, 28, 112, 4, 24.0))
        cases.append(("A", 0, 48, 112, 4, 24.0))
        cases.append(("A", 0, 48, 112, 2, 24.0))
        cases.append(("A", 0, 48, 112, 8, 24.0))
        cases.append(("A", 0, 48, 102, 4, 24.0))
elif os.environ.get("KB_QUICK"):
    cases.append(("A", 0, 28, 112, 4, 48.0))
    cases.append(("A", 0, 24, 111, 2, 48.0))
    cases.append(("B", 1, 44, 1, 8, 24.0))
else:
    # A: (ry, nw, nchunk, remap, pf, nt)
    for ry, nw, ncs in ((4, 4, (4,)), (2, 4, (2, 8)), (4, 8, (4,)), (2, 8, (4, 8))):
        for nchunk in ncs:
            for remap in (0, 1):
                for pf in (0, 1):
                    for nt in (0, 1, 2):
                        if ry == 4 and pf == 1:
                            continue
                        cases.append(("A", 0, ry * 10 + nw, remap * 100 + pf * 10 + nt, nchunk, 48.0))
    for ry in (4, 2):
        for nchunk in (8,):
            for nt in (0, 1):
                cases.append(("B", 1, ry * 10 + 4, nt, nchunk, 24.0))
res = {c: [] for c in cases}
for rd in range(rounds):
    for c in cases:
        ms = C.c_double()
        nb = C.c_int()
        rc = f(P.h, c[1], c[2], c[3], c[4], 10, C.c_void_p(src.data_ptr()) if rd == 0 else None, C.byref(ms), C.byref(nb))
        assert rc == 0, rc
        res[c].append((ms.value, nb.value))
print(f"grid {n}^3, {rounds} rounds x 10 launches; GB/s = moved bytes (B/cell * cells) / median time")
for c in cases:
    t = [v[0] for v in res[c]]
    med, mn = statistics.median(t), min(t)
    cells = P.ncell if c[1] != 2 else (P.ncell * 1.07)
    print(f"{c[0]:>3s} ry,nw={c[2]} remap/pf/nt={c[3]:03d} nchunk={c[4]:2d} blocks={res[c][0][1]:5d}  median {med:7.4f} ms  min {mn:7.4f} ms  {c[5] * cells / med / 1e6:8.1f} GB/s")
P.close()
