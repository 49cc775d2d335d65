#!/usr/bin/env python3
"""Second placement experiment: from a fast six-stream layout inside one arena, move ONE stream across the whole arena and
record the launch time of the plain 3r+3w streaming kernel -- a 1-D map of where that stream collides with the other five.
Then 300 random layouts for model fitting.  GPU only."""
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
N = 512 ** 3
VB = N * 8


def t(offs_mib, nr=3, nw=3, reps=3):
    ms = C.c_double()
    arr = (C.c_int64 * 6)(*[o * MB for o in offs_mib])
    rc = f(P.h, ARENA, arr, N, nr, nw, reps, C.byref(ms), None)
    assert rc == 0, rc
    return ms.value


fast = [9856, 13310, 18262, 7820, 15832, 21494]
print("fast layout", fast, " ".join(f"{t(fast):.4f}" for _ in range(4)), flush=True)
for who in (0, 5):
    print(f"# scan stream {who} ({'read' if who < 3 else 'write'}) in 16 MiB steps", flush=True)
    others = [fast[k] for k in range(6) if k != who]
    line = []
    for pos in range(0, 24576 - 1024 + 1, 16):
        if any(abs(pos - o) < 1024 for o in others):
            continue
        o = list(fast); o[who] = pos
        line.append(f"{pos}:{t(o, reps=2):.3f}")
        if len(line) == 16:
            print(" ".join(line), flush=True); line = []
    print(" ".join(line), flush=True)
print("# fine scan of stream 0 in 2 MiB steps over [0, 768)", flush=True)
line = []
for pos in range(0, 768, 2):
    o = list(fast); o[0] = pos
    line.append(f"{pos}:{t(o, reps=2):.3f}")
    if len(line) == 16:
        print(" ".join(line), flush=True); line = []
print(" ".join(line), flush=True)
print("# 300 random 2 MiB-aligned layouts", flush=True)
rnd = random.Random(11)
for trial in range(300):
    while True:
        o = sorted(rnd.randrange(0, (24576 - 1024) // 2) * 2 for _ in range(6))
        if all(o[k + 1] - o[k] >= 1024 for k in range(5)):
            break
    rnd.shuffle(o)
    print(f"R {t(o, reps=2):.4f} " + " ".join(str(x) for x in o), flush=True)
P.close()
