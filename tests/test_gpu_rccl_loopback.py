"""-m gpu: the RCCL transport on a one-GPU box.  With the tuning knob "comm_loopback" = 1 (fl_tuning_set; FLUCA_COMM_LOOPBACK=1 gives it its
initial value) a single rank does not copy the ghost layers of
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


def _loop(on):
    from fluca_amd import capi
    capi.check(capi.lib.fl_tuning_set(b"comm_loopback", int(on)))


@pytest.fixture
def loopback():
    # the knob is looked at when a handle is created (the environment only once per process, for its initial value)
    _loop(1)
    yield
    _loop(0)


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
        _loop(loop)
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


def test_whole_time_steps_through_rccl(loopback):
    """Two CNLinear steps of the C host mirror on a periodic box (Taylor-Green) with every ghost exchange of the step -- pressure,
    three velocity components, twelve face fields, the multigrid levels -- and every reduction going through RCCL."""
    import ctypes as C
    from fluca_amd import capi, hostapi as H, poisson as flp
    P = C.c_void_p
    L, n = 2 * np.pi, 16

    def run(loop):
        _loop(loop)
        mesh = P()
        assert H.lib.MeshCartCreate3d(1, 1, 1, n, n, 8, 1, 1, 1, None, None, None, C.byref(mesh)) == 0 and H.lib.MeshSetUp(mesh) == 0
        assert H.lib.MeshCartSetUniformCoordinates(mesh, 0., L, 0., L, 0., L * 8 / n) == 0
        ns = P()
        assert H.lib.NSCreate(C.byref(ns)) == 0 and H.lib.NSSetType(ns, b"cnlinear") == 0 and H.lib.NSSetMesh(ns, mesh) == 0
        assert H.lib.NSSetDensity(ns, 1.0) == 0 and H.lib.NSSetViscosity(ns, 0.1) == 0
        for b in range(6):
            assert H.lib.NSSetBoundaryCondition(ns, b, H.NSBoundaryCondition(type=H.NS_BC_PERIODIC)) == 0
        argc, av = H.argv("-ns_time_step_size", 0.05, "-ns_max_steps", 2, "-ns_ksp_rtol", 1e-9, "-ns_abf_schur_ksp_rtol", 1e-11,
                          "-ns_abf_momentum_ksp_rtol", 1e-11, "-ns_abf_schur_pc_type", "mg")
        assert H.lib.NSSetFromOptions(ns, argc, av) == 0 and H.lib.NSSetUp(ns) == 0
        if loop:
            hp = P()
            assert H.lib.NSGetPoisson(ns, C.byref(hp)) == 0
            idb = (C.c_char * capi.UNIQUE_ID_BYTES).from_buffer_copy(flp.rccl_unique_id())
            capi.check(capi.lib.fl_poisson_comm_init_rccl(hp, idb, 0, 1))
        v, p, V = P(), P(), (C.c_void_p * 3)()
        assert H.lib.NSGetSolutionArrays(ns, C.byref(v), V, C.byref(p)) == 0
        h = L / n
        xc, xf = (np.arange(n) + 0.5) * h, np.arange(n) * h
        Z = np.ones((8, 1, 1))
        put = lambda ptr, a: capi.check(capi.lib.fl_memcpy_h2d(0, ptr, np.ascontiguousarray(a, dtype=np.float64).ctypes.data_as(C.c_void_p), a.size * 8))
        u0 = Z * np.sin(xc)[None, None, :] * np.cos(xc)[None, :, None]
        w0 = Z * (-np.cos(xc)[None, None, :] * np.sin(xc)[None, :, None])
        put(v, np.stack([u0, w0, np.zeros_like(u0)]))
        put(C.c_void_p(V[0]), Z * np.sin(xf)[None, None, :] * np.cos(xc)[None, :, None])
        put(C.c_void_p(V[1]), Z * (-np.cos(xc)[None, None, :] * np.sin(xf)[None, :, None]))
        X, Y = np.meshgrid(xc, xc, indexing="xy")
        put(p, Z * (0.25 * (np.cos(2 * X) + np.cos(2 * Y)))[None, :, :])
        assert H.lib.NSSolve(ns) == 0
        out = np.empty(3 * 8 * n * n)
        capi.check(capi.lib.fl_memcpy_d2h(0, out.ctypes.data_as(C.c_void_p), v, out.size * 8))
        H.lib.NSDestroy(C.byref(ns)), H.lib.MeshDestroy(C.byref(mesh))
        return out

    a, b = run(True), run(False)
    assert np.abs(b).max() > 0.5 and np.abs(a - b).max() <= 1e-9 * np.abs(b).max()


def test_ibm_interpolation_reduces_through_rccl(loopback):
    """fl_ibm_interp of a multi-rank handle ends with an all-reduce of 3 L doubles (markers are replicated on the ranks): here
    through ncclAllReduce, against the oracle; spreading needs no communication."""
    import ctypes as C
    import torch
    from fluca_amd import capi, poisson as flp
    n, box, bc = (24, 20, 16), [(0.0, 1.0)] * 3, [PER, PER, V, V, PER, PER]
    P = flp.Poisson.uniform(n, box, bc, 1e-3)
    P.comm_init_rccl(flp.rccl_unique_id(), 0, 1)
    g = fo.Grid.uniform(n, box, bc, 1e-3)
    rng = np.random.default_rng(17)
    L = 1001
    X = [rng.uniform(0.0, 1.0, L) for _ in range(3)]
    X[1] = 0.1 + 0.8 * X[1]                      # keep the supports off the walls in y
    X[0][:2], X[2][:2] = [0.003, 0.998], [0.999, 0.001]   # across the periodic seams
    Xd = [dev(a) for a in X]
    m = C.c_void_p()
    P._pre()
    capi.check(capi.lib.fl_ibm_create(P.h, 0, L, *[C.c_void_p(t.data_ptr()) for t in Xd], C.byref(m)))
    u = rng.standard_normal((3, g.ncell))
    ud = dev(u)
    U = torch.empty(3 * L, dtype=torch.float64, device="cuda")
    capi.check(capi.lib.fl_ibm_interp(m, 3, C.c_void_p(ud.data_ptr()), C.c_void_p(U.data_ptr())))
    P.synchronize()
    assert np.allclose(host(U).reshape(3, L), g.ibm_interp(0, X, u), rtol=1e-12, atol=1e-13)
    F, dV, f0 = rng.standard_normal((3, L)), rng.uniform(0.5, 1.5, L) * 1e-3, rng.standard_normal((3, g.ncell))
    fd, Fd, dVd = dev(f0), dev(F), dev(dV)
    capi.check(capi.lib.fl_ibm_spread(m, 3, C.c_void_p(Fd.data_ptr()), C.c_void_p(dVd.data_ptr()), C.c_void_p(fd.data_ptr())))
    P.synchronize()
    want = g.ibm_spread(0, X, dV, F, f0.copy())
    assert np.allclose(host(fd).reshape(3, -1), want, rtol=1e-12, atol=1e-12 * np.abs(want).max())
    capi.lib.fl_ibm_destroy(m)
    P.close()

