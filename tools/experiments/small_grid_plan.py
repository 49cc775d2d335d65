"""Small-grid tiling sweep of k_cg_A / k_cg_B (latency-bound regime: the reference's default 64 x 64 x 32 cavity)."""
import ctypes as C
import os
import statistics
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch

from fluca_amd import capi
from fluca_amd.poisson import Poisson

n = tuple(int(v) for v in (sys.argv[1:4] if len(sys.argv) > 3 else (64, 64, 32)))
P = Poisson.uniform(n, [(0, 1), (0, 1), (0, 0.5)], [1, 1, 1, 1, 4, 1], 1e-3)
src = torch.rand(P.ncell, dtype=torch.float64, device="cuda") - 0.5
torch.cuda.synchronize()
f = capi.lib.fldbg_bench
f.restype = C.c_int
f.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.POINTER(C.c_double), C.POINTER(C.c_int)]
cases = []
for ry, nw in ((2, 8), (2, 4), (1, 4)):
    for nchunk in (0, 4, 8, 16, 32):
        if nchunk > n[2]:
            continue
        cases.append(("A", 0, ry * 10 + nw, 112, nchunk))
for ry in (4, 2, 1):
    for nchunk in (0, 4, 8, 16, 32):
        if nchunk > n[2]:
            continue
        cases.append(("B", 1, ry * 10 + 4, 1, nchunk))
res = {c: [] for c in cases}
for rd in range(7):
    for c in cases:
        ms, nb = C.c_double(), C.c_int()
        rc = f(P.h, c[1], c[2], c[3], c[4], 50, C.c_void_p(src.data_ptr()) if rd == 0 else None, C.byref(ms), C.byref(nb))
        assert rc == 0, rc
        res[c].append((ms.value, nb.value))
print("grid", n)
for c in cases:
    t = [v[0] for v in res[c]]
    print(f"{c[0]} ry,nw={c[2]} nchunk={c[4]:2d} blocks={res[c][0][1]:4d}  median {statistics.median(t) * 1e3:7.2f} us  min {min(t) * 1e3:7.2f} us")
P.close()
