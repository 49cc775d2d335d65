"""Several ranks in ONE process: one host thread and one fl_poisson handle per rank, all on the same device, joined by the in-memory
transport of tests/plugins/inproc_comm.c (fl_poisson_comm_init_host).  What include/fluca_hip.h promises -- "a handle is driven by one
host thread; different handles are independent" -- and the only way to run config 5's 2 x 2 x 2 rank grid on a one-GPU box: eight
processes would exceed the box's limit of six processes on the card, eight threads do not.

The threads are Python threads: ctypes releases the GIL around every call into libflucahip.so, the transport's callbacks are C and never
take it, so the ranks really run concurrently inside the library (and block on each other there, as processes would)."""
import ctypes as C
import os
import subprocess
import threading
import traceback

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "tests", "plugins", "inproc_comm.c")
OUT = os.path.join(ROOT, "tests", "plugins", "libinproc_comm.so")

_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(OUT) or os.path.getmtime(OUT) < os.path.getmtime(SRC):
            subprocess.check_call(["gcc", "-std=gnu99", "-O2", "-Wall", "-Werror", "-fPIC", "-shared", "-pthread", "-o", OUT, SRC])
        L = C.CDLL(OUT)
        L.inproc_world_create.restype = C.c_void_p
        L.inproc_world_create.argtypes = [C.c_int, C.c_double]
        L.inproc_world_destroy.argtypes = [C.c_void_p]
        L.inproc_ctx_create.restype = C.c_void_p
        L.inproc_ctx_create.argtypes = [C.c_void_p, C.c_int]
        L.inproc_ctx_destroy.argtypes = [C.c_void_p]
        L.inproc_abort.argtypes = [C.c_void_p]
        L.inproc_aborted.argtypes = [C.c_void_p]
        L.inproc_stats.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_longlong)]
        L.inproc_barrier.argtypes = [C.c_void_p]
        L.inproc_allreduce.argtypes = [C.c_void_p, C.POINTER(C.c_double), C.c_int]
        _lib = L
    return _lib


class Rank:
    """What a rank's thread gets: its number, the world size and the transport to hand to fl_poisson_comm_init_host."""

    def __init__(self, world, rank):
        self.world, self.rank, self.size = world, rank, world.size
        self.ctx = C.c_void_p(lib().inproc_ctx_create(world.w, rank))

    def attach(self, handle):
        """fl_poisson_comm_init_host(handle, inproc_exchange, inproc_allreduce, ctx, rank, size); handle = fl_poisson* as c_void_p"""
        from fluca_amd import capi
        L = lib()
        xf = C.cast(L.inproc_exchange, capi.EXCHANGE_FN)
        af = C.cast(L.inproc_allreduce, capi.ALLREDUCE_FN)
        capi.check(capi.lib.fl_poisson_comm_init_host(handle, xf, af, self.ctx, self.rank, self.size), "fl_poisson_comm_init_host")

    def barrier(self):
        assert lib().inproc_barrier(self.ctx) == 0, "in-process barrier failed (another rank aborted or timed out)"

    def allreduce(self, arr):
        """in-place sum of a float64 numpy array over the ranks"""
        assert arr.dtype.name == "float64" and arr.flags.c_contiguous
        assert lib().inproc_allreduce(self.ctx, arr.ctypes.data_as(C.POINTER(C.c_double)), arr.size) == 0

    def stats(self):
        out = (C.c_longlong * 4)()
        lib().inproc_stats(self.world.w, self.rank, out)
        return dict(allreduces=out[0], exchanges=out[1], messages=out[2], bytes=out[3])


class World:
    def __init__(self, size, timeout_s=120.0):
        self.size = size
        self.w = C.c_void_p(lib().inproc_world_create(size, float(timeout_s)))
        assert self.w
        self.ranks = [Rank(self, r) for r in range(size)]

    def close(self):
        for r in self.ranks:
            lib().inproc_ctx_destroy(r.ctx)
        lib().inproc_world_destroy(self.w)
        self.w = None


def run_threads(size, fn, *args, timeout=600.0, wire_timeout=60.0):
    """fn(rank: Rank, *args) on `size` threads; returns the list of results in rank order; the first failure aborts the wire (so nobody
    hangs in an exchange) and is re-raised with every rank's traceback."""
    world = World(size, wire_timeout)
    results, errors = [None] * size, []

    def entry(r):
        try:
            import torch
            torch.cuda.set_device(0)
            results[r] = fn(world.ranks[r], *args)
        except BaseException:  # noqa: BLE001  (an assertion in one rank must free the others)
            errors.append((r, traceback.format_exc()))
            lib().inproc_abort(world.w)

    threads = [threading.Thread(target=entry, args=(r,), name=f"rank{r}", daemon=True) for r in range(size)]
    for t in threads:
        t.start()
    for t in threads:
        t.join(timeout)
    hung = [t.name for t in threads if t.is_alive()]
    if hung:
        lib().inproc_abort(world.w)
        for t in threads:
            t.join(30.0)
    if errors or hung:
        errors.sort()
        raise AssertionError("in-process multi-rank run failed" + (f" (hung: {hung})" if hung else "") + ":\n" +
                             "\n".join(f"--- rank {r} ---\n{tb}" for r, tb in errors))
    world.close()
    return results
