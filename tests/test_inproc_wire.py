"""CPU: the in-memory test transport itself (tests/plugins/inproc_comm.c) -- the wire under tests/test_gpu_config5.py must be beyond suspicion
before a GPU result is blamed on the library: FIFO matching by (source, tag), rank-ordered sums with the same bits everywhere, abort instead
of a hang, a size mismatch reported to both sides."""
import ctypes as C
import threading

import numpy as np

from tests import inproc


def _run(size, fn):
    world = inproc.World(size, 10.0)
    out, err = [None] * size, []

    def entry(r):
        try:
            out[r] = fn(world.ranks[r])
        except BaseException as e:  # noqa: BLE001
            err.append((r, e))
            inproc.lib().inproc_abort(world.w)

    ts = [threading.Thread(target=entry, args=(r,)) for r in range(size)]
    [t.start() for t in ts]
    [t.join(30) for t in ts]
    assert not any(t.is_alive() for t in ts)
    return out, err, world


def _exchange(R, msgs):
    """msgs: (peer, sendtag, recvtag, send array | None, recv array | None)"""
    n = len(msgs)
    peer = (C.c_int * n)(*[m[0] for m in msgs])
    st = (C.c_int * n)(*[m[1] for m in msgs])
    rt = (C.c_int * n)(*[m[2] for m in msgs])
    sp = (C.c_void_p * n)(*[None if m[3] is None else m[3].ctypes.data for m in msgs])
    rp = (C.c_void_p * n)(*[None if m[4] is None else m[4].ctypes.data for m in msgs])
    nb = (C.c_int64 * n)(*[(m[3] if m[3] is not None else m[4]).nbytes for m in msgs])
    L = inproc.lib()
    L.inproc_exchange.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    return L.inproc_exchange(R.ctx, n, peer, st, rt, sp, rp, nb)


def test_allreduce_is_rank_ordered_and_identical_on_every_rank():
    vals = [np.array([1e16, 1.0, -1e16, 3.0 + r, 0.1 * r]) for r in range(8)]

    def fn(R):
        res = []
        for rep in range(50):                      # back to back: the buffers are reused, nobody may overtake
            a = vals[R.rank] * (rep + 1)
            R.allreduce(a)
            res.append(a.copy())
        return res

    out, err, world = _run(8, fn)
    assert not err
    for rep in range(50):
        want = np.zeros(5)
        for r in range(8):                          # rank order, one addition at a time: the sum the wire promises
            want = want + vals[r] * (rep + 1)
        for r in range(8):
            assert np.array_equal(out[r][rep], want)
    assert world.ranks[3].stats()["allreduces"] == 50
    world.close()


def test_messages_match_by_source_and_tag_in_fifo_order():
    def fn(R):
        nxt, prv = (R.rank + 1) % R.size, (R.rank - 1) % R.size
        got = []
        for rep in range(20):
            a, b = np.full(7, 100.0 * R.rank + rep), np.full(3, -100.0 * R.rank - rep)
            ra, rb = np.empty(7), np.empty(3)
            # two messages to the same peer with different tags, posted in the opposite order to the receives
            assert _exchange(R, [(nxt, 5, 9, a, None), (nxt, 9, 5, b, None), (prv, 0, 9, None, rb), (prv, 0, 5, None, ra)]) == 0
            got.append((ra.copy(), rb.copy()))
        return got

    out, err, world = _run(4, fn)
    assert not err
    for r in range(4):
        p = (r - 1) % 4
        for rep, (ra, rb) in enumerate(out[r]):
            assert np.all(ra == 100.0 * p + rep) and np.all(rb == -100.0 * p - rep)
    world.close()


def test_a_failing_rank_frees_the_others():
    def fn(R):
        if R.rank == 1:
            raise RuntimeError("rank 1 gives up")
        a = np.ones(1)
        R.allreduce(a)           # would wait for rank 1 for ever

    out, err, world = _run(3, fn)
    assert len(err) == 3 and inproc.lib().inproc_aborted(world.w)
    world.close()


def test_a_size_mismatch_is_an_error_not_a_truncated_copy():
    def fn(R):
        a, r = np.ones(4 + R.rank), np.empty(4 + R.rank)
        return _exchange(R, [(1 - R.rank, 1, 1, a, r)])

    out, err, world = _run(2, fn)
    assert not err and sorted(out) != [0, 0]
    world.close()
