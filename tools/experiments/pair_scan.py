#!/usr/bin/env python3
"""Pair conflict function of k_cg_A at 512^3: three of its five vectors parked 8 GiB apart (the fast configuration), the other
two at distance d MiB from each other; d scanned.  Roles: r, P0, P1, q, x.  GPU only."""
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
ARENA = 56 * GB
assert hip.hipMalloc(C.byref(p), ARENA) == 0
assert hip.hipMemset(p, 0, ARENA) == 0
torch.cuda.synchronize()
a0 = (p.value + 2 * MB - 1) // (2 * MB) * (2 * MB)
NAMES = ["r", "P0", "P1", "q", "x"]


def t(offs_mib, nchunk=0, reps=2):
    ms = C.c_double()
    arr = (C.c_void_p * 5)(*[a0 + o * MB for o in offs_mib])
    assert f(P.h, 0, arr, nchunk, reps, C.byref(ms)) == 0
    return ms.value


far = [24 * 1024 + k * 8 * 1024 for k in range(4)]     # 24, 32, 40, 48 GiB
print("all far apart:", " ".join(f"{t([0, far[0], far[1], far[2], far[3]]):.4f}" for _ in range(3)), flush=True)
for (ia, ib) in ((0, 1), (3, 4), (1, 2), (0, 4)):
    print(f"# pair {NAMES[ia]}-{NAMES[ib]}: distance MiB -> ms (the other three at 24.. GiB, 8 GiB apart)", flush=True)
    rest = [k for k in range(5) if k not in (ia, ib)]
    line = []
    for d in list(range(1100, 4200, 50)) + list(range(4200, 20000, 400)):
        o = [0] * 5
        o[ia] = 0
        o[ib] = d
        for n_, k in enumerate(rest):
            o[k] = far[n_ + 1]
        line.append(f"{d}:{t(o):.3f}")
        if len(line) == 16:
            print(" ".join(line), flush=True); line = []
    print(" ".join(line), flush=True)
print("# all five at uniform spacing s MiB", flush=True)
print(" ".join(f"{s}:{t([k * s for k in range(5)]):.3f}" for s in (1100, 1200, 1400, 1600, 1800, 2048, 2300, 2600, 3000, 3150, 3500, 4096, 5000, 6000, 8192, 10000, 12000)), flush=True)
P.close()
