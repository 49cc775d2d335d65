#!/usr/bin/env python3
"""k_cg_A / fused Chebyshev / plain 3r+3w stream at 512^3 on five vectors allocated one after the other with a SPACER allocation
of D GiB between consecutive vectors (spacers freed before timing): does the launch time depend on how far apart in physical
memory the vectors live?  GPU only."""
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
g = capi.lib.fldbg_stream_ptrs
g.restype = C.c_int
g.argtypes = [C.c_void_p, C.POINTER(C.c_void_p), C.c_int64, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_double)]
hip = C.CDLL("libamdhip64.so")
hip.hipMalloc.argtypes = [C.POINTER(C.c_void_p), C.c_size_t]
hip.hipFree.argtypes = [C.c_void_p]
hip.hipMemset.argtypes = [C.c_void_p, C.c_int, C.c_size_t]
sx = ((16 + 512 + 1 + 15) // 16) * 16
padbytes = (sx * 514 * 514 + 256) * 8


def malloc(n):
    p = C.c_void_p()
    rc = hip.hipMalloc(C.byref(p), n)
    assert rc == 0, rc
    return p.value


def measure(ptrs):
    ms = C.c_double()
    arr = (C.c_void_p * 5)(*ptrs)
    out = []
    for kernel, nchunk in ((0, 0), (0, 2), (1, 0)):
        assert f(P.h, kernel, arr, nchunk, 3, C.byref(ms)) == 0
        out.append(ms.value)
    arr6 = (C.c_void_p * 6)(*(list(ptrs) + [ptrs[4]]))
    assert g(P.h, arr6, 512 ** 3, 3, 3, 3, C.byref(ms)) == 0
    out.append(ms.value)
    return out


print("# D GiB spacer between consecutive vectors -> k_cg_A, k_cg_A nchunk 2, k_cheb2 (per two steps), 3r+3w stream [ms]", flush=True)
for D in (0, 2, 4, 8, 12, 16, 18, 20, 24, 32, 40, 0, 16, 24):
    vecs, spacers = [], []
    for k in range(5):
        v = malloc(padbytes)
        hip.hipMemset(v, 0, padbytes)
        vecs.append(v)
        if D > 0 and k < 4:
            spacers.append(malloc(D * GB))
    for s in spacers:
        hip.hipFree(s)
    torch.cuda.synchronize()
    r = measure(vecs)
    print(f"D={D:3d}: " + " ".join(f"{x:.4f}" for x in r) + "   bases " + " ".join(f"{v:#x}" for v in vecs), flush=True)
    for v in vecs:
        hip.hipFree(v)
P.close()
