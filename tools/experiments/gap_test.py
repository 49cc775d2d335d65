import ctypes as C, sys, os, subprocess
if len(sys.argv) > 1:
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
        return ms.value
    out = []
    for trial in range(4):
        a = min(t(0, 28, 112, 4) for _ in range(3))
        b = min(t(1, 44, 1, 8) for _ in range(2))
        out.append(f"{a:.4f}/{b:.4f}")
        t(9, trial % 3, 37 + 64 * trial, 0)
    print("gap", os.environ.get("FLUCA_GAP"), " K_A/K_B per re-allocation:", " ".join(out), flush=True)
else:
    for gap in (0, 128, 4096, 4224, 65536, 69760, 1048576, 2097152, 2101376, 16777216 + 4224):
        subprocess.run([sys.executable, __file__, "x"], env=dict(os.environ, FLUCA_GAP=str(gap)))
