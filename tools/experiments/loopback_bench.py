"""Cost of the multi-rank code path on one GPU: the same 512^3 Jacobi-PCG iteration with the ghost layers of the periodic axes
(a) copied locally and (b) sent to the rank itself through RCCL (tuning knob "comm_loopback" = 1: pack, grouped ncclSend/ncclRecv,
unpack, partial sums -> ncclAllReduce -> scalar kernel).  (b) - (a) is what every rank of an N-GPU run pays per iteration
before any xGMI transfer time.   usage: python tools/experiments/loopback_bench.py [--cells 512] [--axes 3]"""
import argparse
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))

V, PER = 1, 3


def run(cells, axes, loopback, iters=200, sr=0, oneshot=0):
    import ctypes as C
    from fluca_amd import capi
    from fluca_amd import poisson as flp
    capi.check(capi.lib.fl_tuning_set(b"comm_loopback", 1 if loopback else 0))   # looked at by fl_poisson_create
    capi.check(capi.lib.fl_tuning_set(b"allreduce", 1 if oneshot else 0))
    bc = []
    for d in range(3):
        bc += [PER, PER] if d >= 3 - axes else [V, V]
    P = flp.Poisson.uniform((cells,) * 3, [(0, 1)] * 3, bc, 1e-3)
    if loopback:
        P.comm_init_rccl(flp.rccl_unique_id(), 0, 1)
        if oneshot:   # the scalar reductions through the rank's own mailbox (round 5: fl_poisson_comm_oneshot_*), the halos still through RCCL
            addr = C.c_void_p()
            capi.check(capi.lib.fl_poisson_comm_oneshot_handle(P.h, None, C.byref(addr)))
            capi.check(capi.lib.fl_poisson_comm_oneshot_attach(P.h, None, (C.c_void_p * 1)(addr.value)))
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
    one = run(a.cells, a.axes, True, sr=a.single_reduction, oneshot=1)
    print(json.dumps(dict(cells=a.cells, periodic_axes=a.axes, single_reduction=a.single_reduction, ms_per_iter_local_wrap=local, ms_per_iter_rccl_loopback=loop,
                          overhead_ms=loop - local, efficiency_bound=local / loop, ms_per_iter_oneshot_allreduce=one, overhead_ms_oneshot=one - local,
                          efficiency_bound_oneshot=local / one)))


if __name__ == "__main__":
    main()
