#!/usr/bin/env python3
"""k_cg_A at 512^3 on (i) uniform layouts in the first 8 GiB of one 24 GiB arena, (ii) random 2 MiB-aligned layouts anywhere in
the arena, (iii) freshly and separately allocated vectors (what fl_ensure_vec does) -- all in ONE process on ONE box, with the
plain 3r+3w stream kernel on the same pointers beside it.  GPU only."""
import ctypes as C, os, sys, random, subprocess
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from fluca_amd import capi
from fluca_amd.poisson import Poisson

MB = 1 << 20
print(subprocess.run("rocm-smi --showuniqueid --showmemuse --showtemp 2>&1 | grep -i 'unique\\|memory\\|Temp' | head", shell=True, capture_output=True, text=True).stdout, flush=True)
P = Poisson.uniform((512, 512, 512), [(0, 1), (0, 1), (0, 0.5)], [1, 1, 1, 1, 4, 1], 1e-3)
f = capi.lib.fldbg_kernel_ptrs
f.restype = C.c_int
f.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_void_p), C.c_int, C.c_int, C.POINTER(C.c_double)]
g = capi.lib.fldbg_stream_ptrs
g.restype = C.c_int
g.argtypes = [C.c_void_p, C.POINTER(C.c_void_p), C.c_int64, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_double)]
sx = ((16 + 512 + 1 + 15) // 16) * 16
padbytes = (sx * 514 * 514 + 256) * 8
VMB = (padbytes + MB - 1) // MB + 1     # vector length in whole MiB


def both(ptrs, nchunk=0):
    ms = C.c_double()
    arr = (C.c_void_p * 5)(*ptrs)
    assert f(P.h, 0, arr, nchunk, 3, C.byref(ms)) == 0
    a = ms.value
    arr6 = (C.c_void_p * 6)(*(list(ptrs) + [ptrs[4]]))       # r, P0, P1 read; q, x, x written
    assert g(P.h, arr6, 512 ** 3, 3, 3, 3, C.byref(ms)) == 0
    return a, ms.value


arena = torch.zeros(24 * 1024 * MB // 8, dtype=torch.float64, device="cuda")
a0 = (arena.data_ptr() + 2 * MB - 1) // (2 * MB) * (2 * MB)
torch.cuda.synchronize()
print("# (i) uniform spacing in the arena's first 8 GiB: spacing MiB -> k_cg_A ms / stream ms", flush=True)
for sp in (VMB, 1100, 1152, 1280, 1280 + 51, 1536):
    a, s = both([a0 + k * sp * MB for k in range(5)])
    print(f"  spacing {sp:5d}: {a:.4f} / {s:.4f}", flush=True)
print("# (ii) random layouts anywhere in the arena (offsets MiB)", flush=True)
rnd = random.Random(3)
for trial in range(24):
    while True:
        o = sorted(rnd.randrange(0, (24 * 1024 - VMB - 4) // 2) * 2 for _ in range(5))
        if all(o[k + 1] - o[k] >= VMB for k in range(4)):
            break
    rnd.shuffle(o)
    a, s = both([a0 + x * MB for x in o])
    print(f"  {a:.4f} / {s:.4f}   " + " ".join(f"{x:6d}" for x in o), flush=True)
del arena
torch.cuda.empty_cache()
print("# (iii) separate allocations, re-allocated every round (bases hex)", flush=True)
keep = []
for rd in range(10):
    bufs = [torch.zeros(padbytes // 8 + 64, dtype=torch.float64, device="cuda") for _ in range(5)]
    torch.cuda.synchronize()
    a, s = both([b.data_ptr() for b in bufs])
    a2, _ = both([b.data_ptr() for b in bufs], 2)
    print(f"  {a:.4f} (nchunk 2: {a2:.4f}) / {s:.4f}   " + " ".join(f"{b.data_ptr():#x}" for b in bufs), flush=True)
    if rd % 2 == 0:
        keep.append(bufs)     # hold some, so that the next round lands elsewhere
    else:
        del bufs
        torch.cuda.empty_cache()
P.close()
