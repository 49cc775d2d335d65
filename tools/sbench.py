#!/usr/bin/env python3
"""Streaming-bandwidth ceiling sweep (reads/writes mix, unroll, non-temporal, grid size).  GPU only."""
import ctypes as C, statistics, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from fluca_amd import capi
from fluca_amd.poisson import Poisson
n = 512
P = Poisson.uniform((n, n, n), [(0, 1), (0, 1), (0, 0.5)], [1, 1, 1, 1, 4, 1], 1e-3)
src = torch.rand(P.ncell, dtype=torch.float64, device="cuda") - 0.5
torch.cuda.synchronize()
f = capi.lib.fldbg_bench
f.restype = C.c_int
f.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.POINTER(C.c_double), C.POINTER(C.c_int)]
cases = []
for mix in (11, 21, 33):
    for u in (1, 2, 4, 8):
        for nt in (0, 1):
            for nb in (512, 1024, 2048, 8192, 65536):
                cases.append((mix, u, nt, nb))
res = {c: [] for c in cases}
for rd in range(3):
    for c in cases:
        ms = C.c_double(); nbk = C.c_int()
        rc = f(P.h, 3, c[0], c[1] * 10 + c[2], c[3], 5, C.c_void_p(src.data_ptr()) if rd == 0 and c == cases[0] else None, C.byref(ms), C.byref(nbk))
        assert rc == 0
        res[c].append(ms.value)
cells = (544 * 514 * 514)
for c in cases:
    med = statistics.median(res[c])
    nstream = c[0] // 10 + c[0] % 10
    print(f"mix {c[0]} U={c[1]} NT={c[2]} blocks={c[3]:6d}  {med:7.4f} ms  {8.0 * nstream * cells / med / 1e6:8.1f} GB/s")
P.close()
