"""Cost of the multi-rank code path on one GPU: the same 512^3 Jacobi-PCG iteration with the ghost layers of the periodic axes
(a) copied locally and (b) sent to the rank itself through RCCL (FLUCA_COMM_LOOPBACK=1: pack, grouped ncclSend/ncclRecv,
unpack, partial sums -> ncclAllReduce -> scalar kernel).  (b) - (a) is what every rank of an N-GPU run pays per iteration
before any xGMI transfer time.   usage: python tools/experiments/loopback_bench.py [--cells 512] [--axes 3]"""
import argparse
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))

V, PER = 1, 3


def run(cells, axes, loopback, iters=200, sr=0):
    os.environ["FLUCA_COMM_LOOPBACK"] = "1" if loopback else "0"
    from fluca_amd import poisson as flp
    bc = []
    for d in range(3):
        bc += [PER, PER] if d >= 3 - axes else [V, V]
    P = flp.Poisson.uniform((cells,) * 3, [(0, 1)] * 3, bc, 1e-3)
    if loopback:
        P.comm_init_rccl(flp.rccl_unique_id(), 0, 1)
    P.tune_placement(8)
    g = torch.Generator(device="cuda").manual_seed(1)
    p = torch.rand(P.ncell, dtype=torch.float64, device="cuda", generator=g) * 2 - 1
    b = P.apply(p)
    x = P.empty()
    P.solve(b, x=x, rtol=0.0, atol=0.0, maxit=20, check_every=64, cg_single_reduction=sr)
    _, info = P.solve(b, x=x, rtol=0.0, atol=0.0, maxit=iters, check_every=64, cg_single_reduction=sr)
    P.close()
    return info["seconds"] / iters * 1e3


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--cells", type=int, default=512)
    ap.add_argument("--axes", type=int, default=3, help="number of periodic axes (= neighbours pairs exchanged)")
    ap.add_argument("--single-reduction", type=int, default=0, help="1: KSPCG with -ksp_cg_single_reduction (fl_ksp_opts.cg_single_reduction)")
    a = ap.parse_args()
    local = run(a.cells, a.axes, False, sr=a.single_reduction)
    loop = run(a.cells, a.axes, True, sr=a.single_reduction)
    print(json.dumps(dict(cells=a.cells, periodic_axes=a.axes, single_reduction=a.single_reduction, ms_per_iter_local_wrap=local, ms_per_iter_rccl_loopback=loop,
                          overhead_ms=loop - local, efficiency_bound=local / loop)))


if __name__ == "__main__":
    main()
