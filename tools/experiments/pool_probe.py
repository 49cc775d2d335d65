"""How the k_cg_A probe time depends on WHICH of K pre-allocated vectors play the five roles (placement experiment)."""
import ctypes as C
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from fluca_amd import capi  # noqa: E402
from fluca_amd.poisson import Poisson  # noqa: E402

K, M = int(sys.argv[1]) if len(sys.argv) > 1 else 16, int(sys.argv[2]) if len(sys.argv) > 2 else 80
P = Poisson.uniform((512,) * 3, [(0, 1), (0, 1), (0, 0.5)], [1, 1, 1, 1, 4, 1], 1e-3)
f = capi.lib.fldbg_pool_probe
f.restype = C.c_int
f.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_uint, C.POINTER(C.c_double), C.POINTER(C.c_int)]
ms = (C.c_double * M)()
sel = (C.c_int * (5 * M))()
assert f(P.h, K, M, 7, ms, sel) == 0
t = np.array(ms)
s = np.array(sel).reshape(M, 5)
print("K", K, "M", M, "min %.4f median %.4f max %.4f" % (t.min(), np.median(t), t.max()))
print("sorted:", " ".join("%.3f" % v for v in np.sort(t)))
order = np.argsort(t)
for i in order[:6]:
    print("fast", "%.4f" % t[i], s[i])
for i in order[-3:]:
    print("slow", "%.4f" % t[i], s[i])
# does any single vector explain it?  mean time of the combinations that contain vector k in role a
for a, name in enumerate(("r", "p0", "p1", "q", "x")):
    means = [t[s[:, a] == k].mean() if (s[:, a] == k).any() else np.nan for k in range(K)]
    print(name, " ".join("%.3f" % v for v in means))
P.close()
