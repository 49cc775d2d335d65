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
def t(kernel, ry, pf, nchunk, reps=10):
    ms = C.c_double(); nb = C.c_int()
    rc = f(P.h, kernel, ry, pf, nchunk, reps, C.c_void_p(src.data_ptr()), C.byref(ms), C.byref(nb)); assert rc == 0, rc
    return ms.value
for trial in range(20):
    a3 = t(3, 33, 41, 1024, 3)      # cheap probe: 3 launches of the plain 3r3w stream
    a = min(t(0, 28, 112, 4) for _ in range(2))
    b = t(1, 44, 1, 8)
    free = torch.cuda.mem_get_info()[0] / 2**30
    print(f"set {trial:2d}: probe(3r3w x3) {a3:.4f}  K_A {a:.4f}  K_B {b:.4f}   free {free:.0f} GiB", flush=True)
    t(8, 0, 0, 0)
