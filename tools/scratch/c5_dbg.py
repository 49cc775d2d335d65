import faulthandler, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
log = open("gpurun_out/c5_dbg.log", "a", buffering=1)
faulthandler.enable(file=log)
faulthandler.dump_traceback_later(float(sys.argv[2]) if len(sys.argv) > 2 else 90, exit=True, file=log)
import numpy as np
import torch
from tests import inproc, test_gpu_config5 as T
size = int(sys.argv[1]) if len(sys.argv) > 1 else 2
ranks = {2: (1, 1, 2), 4: (1, 2, 2), 8: (2, 2, 2)}[size]
case = T.Case(n=(64, 48, 32), ranks=ranks, bc=T.C5_BC, box=[(0.0, 2.0), (0.0, 1.5), (0.0, 1.0)])
ref = T._reference(case)
print("reference done", file=log)
t0 = time.time()

def say(R, *a):
    print(f"[{time.time() - t0:7.3f} r{R.rank}]", *a, file=log)

def worker(R):
    say(R, "start")
    P, d, s = T._handle(R, case)
    say(R, "handle")
    with torch.cuda.stream(s):
        dev = lambda a: torch.as_tensor(np.ascontiguousarray(a, dtype=np.float64).ravel(), device="cuda")
        x = dev(T._blk(case, d, ref["p"]))
        say(R, "uploaded")
        y = P.apply(x)
        say(R, "apply enqueued")
        s.synchronize()
        say(R, "apply done", float(abs(y.cpu().numpy() - T._blk(case, d, ref["b"])).max()))
        bd = dev(T._blk(case, d, ref["b"]))
        xg, ig = P.solve(bd, history=True, remove_nullspace=0, rtol=1e-10, maxit=4000, check_every=8)
        s.synchronize()
        say(R, "cg", ig["iters"], ig["reason"], ref["cg"][1]["iters"])
    P.close()
    say(R, "closed")
    return True

print(inproc.run_threads(size, worker, timeout=80, wire_timeout=30), file=log)
print("ALL OK", size, file=log)
