#!/usr/bin/env python3
"""Where are the boundaries?  Five vectors of k_cg_A (512^3) packed back to back (1100 MiB apart) inside one 96 GiB arena; the
packed window slides through the arena in 512 MiB steps.  GPU only."""
import ctypes as C, os, sys, subprocess
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from fluca_amd import capi
from fluca_amd.poisson import Poisson

MB = 1 << 20
GB = 1 << 30
print(subprocess.run("rocm-smi --showuniqueid 2>&1 | grep -i 'unique id:'", shell=True, capture_output=True, text=True).stdout, flush=True)
P = Poisson.uniform((512, 512, 512), [(0, 1), (0, 1), (0, 0.5)], [1, 1, 1, 1, 4, 1], 1e-3)
f = capi.lib.fldbg_kernel_ptrs
f.restype = C.c_int
f.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_void_p), C.c_int, C.c_int, C.POINTER(C.c_double)]
hip = C.CDLL("libamdhip64.so")
hip.hipMalloc.argtypes = [C.POINTER(C.c_void_p), C.c_size_t]
hip.hipMemset.argtypes = [C.c_void_p, C.c_int, C.c_size_t]
p = C.c_void_p()
AG = 96
assert hip.hipMalloc(C.byref(p), AG * GB) == 0
assert hip.hipMemset(p, 0, AG * GB) == 0
torch.cuda.synchronize()
a0 = (p.value + 2 * MB - 1) // (2 * MB) * (2 * MB)
print(f"arena {AG} GiB at {p.value:#x}", flush=True)


def t(offs_mib, kernel=0, nchunk=0, reps=2):
    ms = C.c_double()
    arr = (C.c_void_p * 5)(*[a0 + o * MB for o in offs_mib])
    assert f(P.h, kernel, arr, nchunk, reps, C.byref(ms)) == 0
    return ms.value


print("# packed window (spacing 1100 MiB) starting at b MiB: k_cg_A ms", flush=True)
line = []
for b in range(0, AG * 1024 - 5 * 1100 - 4, 512):
    line.append(f"{b}:{t([b + k * 1100 for k in range(5)]):.3f}")
    if len(line) == 16:
        print(" ".join(line), flush=True); line = []
print(" ".join(line), flush=True)
print("# the same window, fused Chebyshev kernel (per two steps), 4 GiB steps", flush=True)
print(" ".join(f"{b}:{t([b + k * 1100 for k in range(5)], 1):.3f}" for b in range(0, AG * 1024 - 5 * 1100 - 4, 4096)), flush=True)
print("# uniform spacing s MiB from offset 0", flush=True)
print(" ".join(f"{s}:{t([k * s for k in range(5)]):.3f}" for s in (1100, 2048, 4096, 6144, 7168, 8192, 9216, 10240, 12288, 16384, 20480, 23000)), flush=True)
P.close()
