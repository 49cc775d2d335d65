#!/usr/bin/env python3
"""Launch time of k_cg_A and of the fused Chebyshev kernel at 512^3 against the PHASE SPACING of their five vectors inside
one arena: vector k starts at k * (S + s), S = vector size rounded up to 256 MiB (so that s = 0 puts all five at the same
address modulo 256 MiB), s scanned.  GPU only."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from fluca_amd import capi
from fluca_amd.poisson import Poisson

MB = 1 << 20
P = Poisson.uniform((512, 512, 512), [(0, 1), (0, 1), (0, 0.5)], [1, 1, 1, 1, 4, 1], 1e-3)
f = capi.lib.fldbg_kernel_ptrs
f.restype = C.c_int
f.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_void_p), C.c_int, C.c_int, C.POINTER(C.c_double)]
sx = ((16 + 512 + 1 + 15) // 16) * 16
padbytes = (sx * 514 * 514 + 256) * 8
S = ((padbytes + 256 * MB - 1) // (256 * MB)) * 256 * MB
arena = torch.zeros((5 * (S + 256 * MB) + 256 * MB) // 8, dtype=torch.float64, device="cuda")
a0 = (arena.data_ptr() + 256 * MB - 1) // (256 * MB) * (256 * MB)     # VA aligned to 256 MiB (the physical phase is unknown)
print(f"vector {padbytes / MB:.1f} MiB, S = {S // MB} MiB, arena {arena.numel() * 8 / MB:.0f} MiB at {arena.data_ptr():#x}", flush=True)
torch.cuda.synchronize()


def t(kernel, s_mib, nchunk, reps=3, perm=(0, 1, 2, 3, 4)):
    ms = C.c_double()
    ptrs = (C.c_void_p * 5)(*[a0 + perm[k] * S + int(perm[k] * s_mib * MB) // 128 * 128 for k in range(5)])
    rc = f(P.h, kernel, ptrs, nchunk, reps, C.byref(ms))
    assert rc == 0, rc
    return ms.value


for kernel, name, chunks in ((0, "k_cg_A", (4, 2, 8)), (1, "k_cheb2", (2, 4, 1))):
    for nchunk in chunks:
        print(f"# {name} nchunk={nchunk}: spacing s MiB -> ms", flush=True)
        line = []
        for s in [0, 1, 2, 3, 4, 6, 8, 10, 12, 14, 16, 20, 24, 28, 32, 36, 40, 44, 48, 51.2, 56, 64, 72, 80, 96, 112, 128, 160, 192, 224]:
            line.append(f"{s}:{t(kernel, s, nchunk):.4f}")
        print(" ".join(line), flush=True)
print("# repeat k_cg_A nchunk=4 (noise check)", flush=True)
print(" ".join(f"{s}:{t(0, s, 4):.4f}" for s in [0, 8, 16, 32, 51.2, 64, 128]), flush=True)
P.close()
