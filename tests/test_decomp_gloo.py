"""CPU, world_size 2, 4 and 8 over gloo: the product's decomposition + ghost-exchange plan (host-only C-ABI functions
fl_decomp_default / fl_decomp_neighbor / fl_halo_plan -- the same plan fl_poisson_* executes on the GPU) moves exactly the
cells DMGlobalToLocal would: every ghost layer ends up holding the neighbouring block's boundary cells, periodic wrap
included; scalar all-reduce sums over ranks."""
import numpy as np
import pytest

from tests import mp_common as mpc


def _worker(rank, world, n, ranks, periodic):
    from fluca_amd import capi
    d = mpc.decomp_of(capi, n, ranks, rank)
    plan = mpc.halo_plan(capi, d, periodic)
    rng = np.random.default_rng(42)
    F = rng.standard_normal((n[2], n[1], n[0]))                      # the global field, identical on every rank
    ln = (d.len[2], d.len[1], d.len[0])
    L = np.full((ln[0] + 2, ln[1] + 2, ln[2] + 2), np.nan)
    L[1:-1, 1:-1, 1:-1] = F[mpc.block(d)]
    ax_of = {0: 2, 1: 1, 2: 0}                                       # grid axis -> numpy axis of the (z,y,x) array

    def layer(arr, boundary, ghost):
        axis, side = boundary // 2, boundary % 2
        na = ax_of[axis]
        idx = [slice(1, -1)] * 3
        nloc = arr.shape[na] - 2
        idx[na] = (nloc + 1 if side else 0) if ghost else (nloc if side else 1)
        return tuple(idx)

    # periodic axis held by one rank: local wrap (k_wrap_ghosts)
    for axis in range(3):
        if periodic[axis] and ranks[axis] == 1:
            L[layer(L, 2 * axis, True)] = L[layer(L, 2 * axis + 1, False)]
            L[layer(L, 2 * axis + 1, True)] = L[layer(L, 2 * axis, False)]
    msgs, targets = [], []
    for peer, sb, rb, stag, rtag in plan:
        send = np.ascontiguousarray(L[layer(L, sb, False)]).ravel()
        recv = np.empty_like(send) if rb >= 0 else None
        msgs.append((peer, stag, rtag, send, recv))
        targets.append((rb, recv))
    mpc.gloo_exchange(msgs)
    for rb, recv in targets:
        if rb >= 0:
            sl = layer(L, rb, True)
            L[sl] = recv.reshape(L[sl].shape)
    # expectation from the global field
    for axis in range(3):
        na = ax_of[axis]
        for side in (0, 1):
            g_idx = d.lo[axis] - 1 if side == 0 else d.lo[axis] + d.len[axis]
            inside = 0 <= g_idx < n[axis]
            if not inside and not periodic[axis]:
                continue                                              # physical wall: ghost unused
            g_idx %= n[axis]
            sl = list(mpc.block(d))
            sl[na] = g_idx
            want = F[tuple(sl)]
            got = L[layer(L, 2 * axis + side, True)]
            assert np.array_equal(got, want), (rank, axis, side)
    # scalar all-reduce (the dot products of the Krylov solvers)
    v = np.array([float(rank + 1), 2.0 * rank])
    mpc.gloo_allreduce(v)
    assert v[0] == world * (world + 1) / 2 and v[1] == world * (world - 1)


@pytest.mark.parametrize("world,n,ranks,periodic", [
    (2, (8, 6, 5), (1, 1, 2), (False, False, False)),
    (2, (8, 6, 5), (2, 1, 1), (True, False, True)),       # two ranks on a periodic axis: both faces go to the same peer
    (2, (7, 9, 4), (1, 2, 1), (True, True, False)),
    (4, (9, 8, 6), (1, 2, 2), (False, True, True)),
    (4, (12, 5, 4), (4, 1, 1), (True, False, False)),     # >2 ranks on a periodic axis
    (8, (16, 16, 8), (2, 2, 2), (False, False, True)),    # BASELINE config 5's decomposition (1024 x 1024 x 512 on 2 x 2 x 2 GPUs,
                                                          # periodic span) at toy size: every rank has three face neighbours
])
def test_halo_plan_moves_the_right_cells(world, n, ranks, periodic):
    mpc.run_ranks(world, _worker, n, ranks, periodic)
