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
    res = []
    for cfg in ((28, 112, 4), (28, 110, 4), (28, 12, 4), (24, 111, 2), (28, 112, 8), (44, 101, 4)):
        res.append(f"A{cfg}={min(t(0, *cfg) for _ in range(3)):.4f}")
    for cfg in ((44, 1, 8), (44, 0, 8), (24, 1, 8)):
        res.append(f"B{cfg}={min(t(1, *cfg) for _ in range(2)):.4f}")
    print("IL", os.environ.get("FLUCA_INTERLEAVE"), " ".join(res), flush=True)
    b = P.apply(src); x = P.empty()
    for _ in range(2):
        _, info = P.solve(b, x=x, rtol=0.0, atol=0.0, maxit=100, check_every=64)
    print("IL", os.environ.get("FLUCA_INTERLEAVE"), "solve ms/it", info["seconds"] / info["iters"] * 1e3, flush=True)
else:
    for il in (0, 6, 8, 5, 0, 6):
        subprocess.run([sys.executable, __file__, "x"], env=dict(os.environ, FLUCA_INTERLEAVE=str(il)))
