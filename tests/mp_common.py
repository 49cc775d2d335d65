"""Helpers shared by the multi-process tests (CPU gloo and GPU): process spawning, the gloo halo transport, block slicing."""
import ctypes as C
import os
import socket
import traceback

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _entry(rank, world, port, fn, args, errq):
    try:
        os.environ["MASTER_ADDR"] = "127.0.0.1"
        os.environ["MASTER_PORT"] = str(port)
        # the oracle is OpenMP-parallel: `world` ranks each spinning up one thread per core turn a two-second CPU solve into minutes
        # (round 4: a four-rank case sat in MgOracle.pcg for 140 s and looked like a hang).  Share the cores out before the oracle loads.
        try:
            cores = len(os.sched_getaffinity(0))
        except AttributeError:  # pragma: no cover
            cores = os.cpu_count() or 1
        os.environ["OMP_NUM_THREADS"] = str(max(1, cores // world))
        torch.set_num_threads(max(1, cores // world))
        if os.environ.get("FLUCA_MP_DUMP"):   # debugging aid: the Python stack of every rank after that many seconds (where a hang sits)
            import faulthandler
            import sys
            faulthandler.dump_traceback_later(float(os.environ["FLUCA_MP_DUMP"]), exit=False, file=sys.stderr)
        dist.init_process_group("gloo", rank=rank, world_size=world)
        fn(rank, world, *args)
        dist.barrier()
        dist.destroy_process_group()
    except Exception:  # pragma: no cover
        errq.put((rank, traceback.format_exc()))
        raise


def run_ranks(world, fn, *args, timeout=600):
    """Run fn(rank, world, *args) in `world` processes joined by a gloo group on 127.0.0.1; re-raise the first failure."""
    ctx = mp.get_context("spawn")
    errq = ctx.SimpleQueue()
    port = free_port()
    procs = [ctx.Process(target=_entry, args=(r, world, port, fn, args, errq)) for r in range(world)]
    import time
    for p in procs:
        p.start()
    # a rank that dies (assertion, GPU fault) leaves the others blocked in gloo: stop waiting a little after the first failure instead of
    # sitting out the whole timeout in silence
    t0, first_fail = time.time(), None
    while any(p.is_alive() for p in procs):
        if first_fail is None and any(p.exitcode not in (None, 0) for p in procs):
            first_fail = time.time()
        if (first_fail is not None and time.time() - first_fail > 10.0) or time.time() - t0 > timeout:
            break
        time.sleep(0.2)
    for p in procs:
        p.join(0.1)
    failed = [p for p in procs if p.exitcode != 0]
    for p in procs:
        if p.is_alive():
            p.kill()
    if failed or not errq.empty():
        msgs = []
        while not errq.empty():
            r, tb = errq.get()
            msgs.append(f"--- rank {r} ---\n{tb}")
        raise AssertionError("multi-process test failed:\n" + "\n".join(msgs) + f"\nexit codes {[p.exitcode for p in procs]}")


from fluca_amd.hostcomm import gloo_exchange  # noqa: E402,F401  (the transport lives in the package)
from fluca_amd import hostcomm as _hostcomm  # noqa: E402

_ALLREDUCE_CALLS = [0]


def gloo_allreduce(vals):
    """the package's all-reduce callback, counted: tests of the single-reduction CG assert one call per iteration"""
    _ALLREDUCE_CALLS[0] += 1
    return _hostcomm.gloo_allreduce(vals)


def allreduce_calls():
    return _ALLREDUCE_CALLS[0]


def decomp_of(capi, n, ranks, rank):
    d = capi.fl_decomp()
    rc = capi.lib.fl_decomp_default((C.c_int64 * 3)(*n), (C.c_int * 3)(*ranks), rank, C.byref(d))
    assert rc == 0
    return d


def halo_plan(capi, d, periodic):
    out = (capi.fl_halo_msg * 12)()
    n = capi.lib.fl_halo_plan(C.byref(d), (C.c_int * 3)(*[int(p) for p in periodic]), out)
    assert 0 <= n <= 12
    return [(out[i].peer, out[i].send_boundary, out[i].recv_boundary, out[i].sendtag, out[i].recvtag) for i in range(n)]


def block(d):
    """slices (z, y, x) of this rank's owned cells in a global (nz, ny, nx) array"""
    return tuple(slice(d.lo[a], d.lo[a] + d.len[a]) for a in (2, 1, 0))


def face_block(d, axis, periodic):
    """slices of the owned faces of `axis` in the global face array (nz, ny, nx) with the face axis extended"""
    sl = []
    for a in (2, 1, 0):
        n = d.len[a]
        if a == axis and d.coord[a] == d.ranks[a] - 1 and not periodic[a]:
            n += 1
        sl.append(slice(d.lo[a], d.lo[a] + n))
    return tuple(sl)
