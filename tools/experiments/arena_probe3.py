#!/usr/bin/env python3
"""Third placement experiment: six SEPARATE allocations (as fl_ensure_vec makes them), each 256 MiB larger than the stream, and
stream k started k*D MiB into its allocation.  If separately allocated gigabyte buffers all start at the same physical phase
modulo the 256 MiB period found by arena_probe2.py, D = 0 is the slow mode and D = 256/6 MiB the fast one.  GPU only."""
import os, sys, subprocess, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch

MB = 1 << 20
N = 512 ** 3
print(subprocess.run("rocm-smi --showtemp --showclocks --showpower 2>&1 | grep -v '^=\\|^$' | head -30", shell=True, capture_output=True, text=True).stdout, flush=True)


def timeit(fn, reps=6):
    fn(); fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


import ctypes as C
from fluca_amd import capi
from fluca_amd.poisson import Poisson
P = Poisson.uniform((32, 32, 32), [(0, 1), (0, 1), (0, 0.5)], [1, 1, 1, 1, 4, 1], 1e-3)
f = capi.lib.fldbg_stream_ptrs
f.restype = C.c_int
f.argtypes = [C.c_void_p, C.POINTER(C.c_void_p), C.c_int64, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_double)]


def t(ptrs, nr=3, nw=3, reps=6):
    ms = C.c_double()
    arr = (C.c_void_p * 6)(*ptrs)
    rc = f(P.h, arr, N, nr, nw, reps, C.byref(ms))
    assert rc == 0, rc
    return ms.value


for rnd in range(4):
    bufs = [torch.empty(N + 256 * MB // 8 + rnd * 12345 * 16, dtype=torch.float64, device="cuda") for _ in range(6)]
    for b_ in bufs:
        b_.zero_()
    torch.cuda.synchronize()
    base = [b_.data_ptr() for b_ in bufs]
    print(f"round {rnd}: bases " + " ".join(f"{p:#x}" for p in base), flush=True)
    for D in (0, 0, 8, 16, 32, 42, 43, 48, 64, 96, 128, 0, 43):
        ptrs = [base[k] + ((k * D * MB) % (256 * MB)) for k in range(6)]
        print(f"  D={D:4d} MiB: 3r+3w {t(ptrs):.4f}   3r+2w {t(ptrs, 3, 2):.4f}", flush=True)
    # permuted phase assignment at D = 43 (which stream gets which phase should not matter)
    for perm in ((0, 3, 1, 4, 2, 5), (5, 4, 3, 2, 1, 0)):
        ptrs = [base[k] + perm[k] * 43 * MB for k in range(6)]
        print(f"  D=43 perm {perm}: {t(ptrs):.4f}", flush=True)
    del bufs
    torch.cuda.empty_cache()
print(subprocess.run("rocm-smi --showtemp --showclocks --showpower 2>&1 | grep -v '^=\\|^$' | head -30", shell=True, capture_output=True, text=True).stdout, flush=True)
P.close()
