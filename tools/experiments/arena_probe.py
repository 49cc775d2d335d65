#!/usr/bin/env python3
"""Where does the 1.11 / 1.27 ms bimodality of six-stream kernels come from?  One 24 GB arena, six 1.07 GB streams (3 read,
3 written by the plain streaming kernel) at chosen offsets.  Prints launch time against the offsets.  GPU only."""
import ctypes as C, os, sys, random
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from fluca_amd import capi
from fluca_amd.poisson import Poisson

P = Poisson.uniform((32, 32, 32), [(0, 1), (0, 1), (0, 0.5)], [1, 1, 1, 1, 4, 1], 1e-3)
f = capi.lib.fldbg_arena_probe
f.restype = C.c_int
f.argtypes = [C.c_void_p, C.c_int64, C.POINTER(C.c_int64), C.c_int64, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_double), C.POINTER(C.c_void_p)]
MB = 1 << 20
ARENA = 24 * 1024 * MB
N = 512 ** 3                       # doubles per stream (1 GiB)
VB = N * 8


def t(offs, nr=3, nw=3, reps=6):
    ms = C.c_double()
    a = C.c_void_p()
    arr = (C.c_int64 * 6)(*offs)
    rc = f(P.h, ARENA, arr, N, nr, nw, reps, C.byref(ms), C.byref(a))
    assert rc == 0, rc
    return ms.value, a.value


base = [k * (VB + 128 * MB) for k in range(6)]
ms, a = t(base)
print(f"arena at {a:#x}; streams of {VB / MB:.0f} MiB; baseline spacing 1152 MiB: {ms:.4f} ms", flush=True)
print("# A: same layout, repeated (noise)")
print(" ".join(f"{t(base)[0]:.4f}" for _ in range(6)), flush=True)
print("# B: move stream 5 (written) by d MiB")
for d in [0, 2, 4, 6, 8, 10, 12, 14, 16, 32, 64, 128, 256, 512, 1024, 2048, 4096]:
    o = list(base); o[5] += d * MB
    print(f"d={d:5d} MiB {t(o)[0]:.4f}", flush=True)
print("# C: move stream 5 by d KiB")
for d in [4, 8, 16, 32, 64, 128, 256, 512, 1024]:
    o = list(base); o[5] += d * 1024
    print(f"d={d:5d} KiB {t(o)[0]:.4f}", flush=True)
print("# D: uniform spacing S MiB between all six")
for S in [1024, 1026, 1028, 1032, 1040, 1056, 1088, 1152, 1280, 1536, 2048, 3072, 4096]:
    o = [k * S * MB for k in range(6)]
    print(f"S={S:5d} MiB {t(o)[0]:.4f}", flush=True)
print("# E: 40 random 2 MiB-aligned layouts (no overlap)")
rnd = random.Random(7)
res = []
for trial in range(40):
    while True:
        o = sorted(rnd.randrange(0, (ARENA - VB) // (2 * MB)) * 2 * MB for _ in range(6))
        if all(o[k + 1] - o[k] >= VB for k in range(5)):
            break
    rnd.shuffle(o)
    ms = t(o, reps=4)[0]
    res.append((ms, o))
    print(f"{ms:.4f} " + " ".join(f"{x // MB:6d}" for x in o), flush=True)
print("# F: fewer streams on the baseline layout: 3r+2w, 2r+2w, 2r+1w, 1r+1w")
for nr, nw in ((3, 2), (2, 2), (2, 1), (1, 1)):
    print(f"{nr}r+{nw}w {t(base, nr, nw)[0]:.4f}", flush=True)
P.close()
