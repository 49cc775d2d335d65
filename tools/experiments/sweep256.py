#!/usr/bin/env python3
"""k_cg_A / k_cg_B plan sweep at 256^3 (BASELINE config 2): tile shape and z-chunk count.  GPU only."""
import ctypes as C, os, statistics, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from fluca_amd import capi
from fluca_amd.poisson import Poisson

n = int(sys.argv[1]) if len(sys.argv) > 1 else 256
P = Poisson.uniform((n, n, n), [(0, 1), (0, 1), (0, 0.5)], [1, 1, 1, 1, 4, 1], 1e-3)
src = torch.rand(P.ncell, dtype=torch.float64, device="cuda") - 0.5
torch.cuda.synchronize()
f = capi.lib.fldbg_bench
f.restype = C.c_int
f.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.POINTER(C.c_double), C.POINTER(C.c_int)]
cases = []
for ry, nw in ((2, 8), (2, 4), (1, 4)):
    for nchunk in (1, 2, 4, 8, 16):
        cases.append(("A", 0, ry * 10 + nw, 112, nchunk, 48.0))
for ry in (4, 2, 1):
    for nchunk in (2, 4, 8, 16, 32):
        cases.append(("B", 1, ry * 10 + 4, 1, nchunk, 24.0))
res = {c: [] for c in cases}
for rd in range(5):
    for c in cases:
        ms, nb = C.c_double(), C.c_int()
        rc = f(P.h, c[1], c[2], c[3], c[4], 20, C.c_void_p(src.data_ptr()) if rd == 0 else None, C.byref(ms), C.byref(nb))
        assert rc == 0, rc
        res[c].append((ms.value, nb.value))
for c in cases:
    t = [v[0] for v in res[c]]
    med = statistics.median(t)
    print(f"{c[0]} ry,nw={c[2]:2d} nchunk={c[4]:2d} blocks={res[c][0][1]:5d}  median {med * 1e3:7.1f} us  min {min(t) * 1e3:7.1f} us  {c[5] * P.ncell / med / 1e6:8.1f} GB/s moved")
P.close()
