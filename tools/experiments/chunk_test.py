import ctypes as C, sys, os
sys.path.insert(0,'.')
import torch
from fluca_amd import capi
from fluca_amd.poisson import Poisson
P = Poisson.uniform((512,)*3, [(0,1),(0,1),(0,0.5)], [1,1,1,1,4,1], 1e-3)
src = torch.rand(P.ncell, dtype=torch.float64, device="cuda") - 0.5
torch.cuda.synchronize()
f = capi.lib.fldbg_bench
f.restype = C.c_int
f.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.POINTER(C.c_double), C.POINTER(C.c_int)]
def t(kernel, ry, pf, nchunk):
    ms = C.c_double(); nb = C.c_int()
    rc = f(P.h, kernel, ry, pf, nchunk, 10, C.c_void_p(src.data_ptr()), C.byref(ms), C.byref(nb)); assert rc == 0, rc
    return ms.value, nb.value
for cfg in ((28, 112), (24, 111), (28, 12), (24, 11)):
    for nchunk in (1, 2, 3, 4, 5, 6, 7, 8, 12, 16):
        r = [t(0, cfg[0], cfg[1], nchunk) for _ in range(3)]
        print(f"cfg {cfg} nchunk {nchunk:2d} blocks {r[0][1]:5d}  K_A {min(x[0] for x in r):.4f}", flush=True)
