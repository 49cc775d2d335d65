"""Is the 2-rank CG (host transport) bit-reproducible run to run, and does the overlapped exchange change bits?"""
import os, sys, tempfile
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from tests import mp_common as mpc
from tests.test_gpu_multirank import _overlap_worker

if __name__ == "__main__":
    n, ranks, bc = (24, 20, 16), (1, 1, 2), [1, 1, 1, 1, 4, 1]
    res = {}
    for tag, ov in (("a1", 1), ("b1", 1), ("a0", 0), ("b0", 0)):
        d = tempfile.mkdtemp()
        mpc.run_ranks(2, _overlap_worker, n, ranks, bc, ov, d)
        res[tag] = [np.load(os.path.join(d, f"ov{ov}_r{r}.npz")) for r in range(2)]
    def cmp(x, y):
        return [(float(np.abs(x[r]["hist"] - y[r]["hist"]).max()), float(np.abs(x[r]["x"] - y[r]["x"]).max()), int(np.argmax(x[r]["hist"] != y[r]["hist"])) ) for r in range(2)]
    print("overlap=1 run vs run:", cmp(res["a1"], res["b1"]))
    print("overlap=0 run vs run:", cmp(res["a0"], res["b0"]))
    print("overlap=1 vs overlap=0:", cmp(res["a1"], res["a0"]))
