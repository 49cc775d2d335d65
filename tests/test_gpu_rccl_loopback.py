"""-m gpu: the RCCL transport on a one-GPU box.  With FLUCA_COMM_LOOPBACK=1 a single rank does not copy the ghost layers of
its periodic axes locally but sends them to ITSELF through the communicator -- dlopen of librccl, ncclCommInitRank,
grouped ncclSend/ncclRecv of the packed faces on the handle's stream, ncclAllReduce of the partial sums: the calls the
N-GPU bench makes, minus a second device.  (RCCL refuses two ranks on one device, so the genuine multi-rank tests use the
host-staged transport: tests/test_gpu_multirank.py.)"""
import os

import numpy as np
import pytest

from oracle import fluca_oracle as fo
from tests.gpu_common import PER, V, dev, host

pytestmark = pytest.mark.gpu


@pytest.fixture
def loopback(monkeypatch):
    monkeypatch.setenv("FLUCA_COMM_LOOPBACK", "1")
    yield


@pytest.mark.parametrize("bc", [[PER] * 6, [PER, PER, V, V, PER, PER]])
def test_rccl_self_exchange_matches_oracle(loopback, bc):
    from fluca_amd import poisson as flp
    n = (40, 24, 16)
    box = [(0, 1), (0, 1), (0, 0.5)]
    P = flp.Poisson.uniform(n, box, bc, 1e-3)
    P.comm_init_rccl(flp.rccl_unique_id(), 0, 1)
    g = fo.Grid.uniform(n, box, bc, 1e-3)
    S = g.assemble_S()
    rng = np.random.default_rng(5)
    p = rng.standard_normal(g.ncell)
    p -= p.mean()
    b = S.mult(p)
    y = host(P.apply(dev(p)))
    assert np.abs(y - b).max() <= 1e-12 * np.abs(b).max()
    # CG (two all-reduces per iteration) and BiCGStab through the communicator
    for ksp in (fo.KSP_CG, fo.KSP_BCGS):
        xo, io = S.solve(b, ksp=ksp, rtol=1e-8, maxit=2000)
        xg, ig = P.solve(dev(b), history=True, type=ksp, rtol=1e-8, maxit=2000, check_every=8)
        assert ig["reason"] == io["reason"] == 2 and abs(ig["iters"] - io["iters"]) <= max(2, io["iters"] // 8)
        m = min(len(ig["history"]), len(io["history"]), 6)
        assert np.allclose(ig["history"][:m], io["history"][:m], rtol=1e-6)
        xg = host(xg)
        assert np.linalg.norm((xg - xg.mean()) - (xo - xo.mean())) <= 1e-5 * np.linalg.norm(xo)
    # face exchange of the divergence and the projection
    Vg = [rng.standard_normal(nf) for nf in g.nface]
    assert np.abs(host(P.rhs(*[dev(a) for a in Vg])) - g.rhs(*Vg)).max() <= 1e-11 * max(1.0, np.abs(g.rhs(*Vg)).max())
    # the momentum block exchanges three components and twelve face fields the same way
    M = flp.Momentum(P)
    V0 = [rng.standard_normal(g.nface[d]) for d in range(3)]
    W = [rng.standard_normal(g.nface[d]) for c in range(3) for d in range(3)]
    M.set_state(0.01, 1.0, 0.02, [dev(a) for a in V0], [dev(a) for a in W])
    A = g.assemble_momentum(1.0, 0.01, -0.5 * 0.02 * 0.01, V0, W)
    v = rng.standard_normal(3 * g.ncell)
    want = A.mult(v)
    assert np.abs(host(M.apply(dev(v))) - want).max() <= 2e-13 * np.abs(want).max()
    M.close()
    P.close()


def test_multigrid_levels_share_the_rccl_communicator(loopback):
    """FL_PC_MG with every level's halo exchange and all-reduce going through the borrowed RCCL communicator: the same
    iteration history as the same solve with local ghost copies."""
    from fluca_amd import poisson as flp
    n = (64, 32, 32)
    box = [(0, 1), (0, 1), (0, 0.5)]
    bc = [PER, PER, V, V, PER, PER]
    rng = np.random.default_rng(9)
    p = rng.standard_normal(n[0] * n[1] * n[2])
    p -= p.mean()
    res = []
    for loop in (True, False):
        if not loop:
            os.environ.pop("FLUCA_COMM_LOOPBACK", None)
        P = flp.Poisson.uniform(n, box, bc, 1e-3)
        if loop:
            P.comm_init_rccl(flp.rccl_unique_id(), 0, 1)
        b = P.apply(dev(p))
        x, info = P.solve(b, history=True, type=fo.KSP_CG, pc=2, rtol=1e-10, maxit=60)
        res.append((host(x), info))
        P.close()
    (xa, ia), (xb, ib) = res
    assert ia["reason"] == ib["reason"] == 2 and ia["iters"] == ib["iters"] and ia["iters"] <= 25
    assert np.allclose(ia["history"], ib["history"], rtol=1e-9)
    assert np.linalg.norm(xa - xb) <= 1e-10 * np.linalg.norm(xb)
    assert np.linalg.norm((xa - xa.mean()) - p) <= 1e-7 * np.linalg.norm(p)

