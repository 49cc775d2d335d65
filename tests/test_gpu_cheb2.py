"""-m gpu: the fused two-step Chebyshev kernel (fl_cheb2.hip) against the CPU oracle and against the one-step kernel.

The fused kernel is only legal without a convergence test between the steps (KSP_NORM_NONE).  "cheb_fuse" = 2 forces it on
every grid where it is legal, 0 switches it off; the answer must be the oracle's KSPCHEBYSHEV restatement either way
(oracle/fluca_oracle.c) and the two GPU paths must agree to round-off.
"""
import ctypes as C

import numpy as np
import pytest

from oracle import fluca_oracle as fo
from tests.gpu_common import CAVITY, O, PER, SYM, V, dev, host, make_pair, mean_free_rhs

pytestmark = pytest.mark.gpu


def _fuse(mode):
    from fluca_amd import capi
    capi.check(capi.lib.fl_tuning_set(b"cheb_fuse", mode), "fl_tuning_set")


@pytest.fixture(autouse=True)
def _restore():
    yield
    _fuse(1)


GRIDS = [
    ((17, 9, 11), CAVITY, False, True),                       # smaller than a tile in every direction, all walls
    ((12, 10, 9), [PER] * 6, False, True),                    # every ring cell comes through a periodic seam
    ((9, 12, 7), [V, O, V, V, PER, PER], False, False),       # BASELINE config 3's boundary types
    ((130, 37, 20), CAVITY, True, True),                      # two tiles in x (the second 2 cells wide), three in y, stretched
    ((136, 70, 12), [PER, PER, V, V, PER, PER], False, True),  # periodic seam between different tiles
    ((131, 33, 5), [PER, PER, PER, PER, V, V], False, True),  # odd nx: the last pair straddles the seam
    ((256, 16, 3), [V, V, PER, PER, PER, PER], True, True),   # exact tile multiples, three planes
    ((2, 2, 2), [PER] * 6, False, True),                      # the smallest legal periodic box
    ((64, 48, 40), [V, O, V, V, PER, PER], True, False),      # several z chunks
]


@pytest.mark.parametrize("n,bc,nonuni,nullspace", GRIDS)
@pytest.mark.parametrize("maxit", [2, 7, 20])
def test_fused_steps_match_oracle_and_single_steps(n, bc, nonuni, nullspace, maxit):
    P, g = make_pair(n, bc, kappa=1e-3, nonuniform=nonuni)
    S = g.assemble_S()
    if nullspace:
        _, b = mean_free_rhs(S, g.ncell)
    else:
        b = np.random.default_rng(5).standard_normal(g.ncell)
    lam = S.gershgorin(fo.PC_JACOBI)
    emin, emax = 0.1 * lam, 1.1 * lam
    xo, io = S.solve(b, ksp=fo.KSP_CHEBYSHEV, pc=fo.PC_JACOBI, norm=fo.NORM_NONE, nullspace=nullspace, maxit=maxit, emin=emin, emax=emax)
    out = {}
    for mode in (0, 2):
        _fuse(mode)
        xg, ig = P.solve(dev(b), type=2, pc=fo.PC_JACOBI, norm_type=fo.NORM_NONE, remove_nullspace=int(nullspace), maxit=maxit,
                         emin=emin, emax=emax, check_every=5)
        assert ig["reason"] == io["reason"] == 4 and ig["iters"] == io["iters"] == maxit
        out[mode] = host(xg)
    scale = max(np.linalg.norm(xo), 1e-300)
    assert np.linalg.norm(out[2] - xo) <= 1e-9 * scale
    assert np.linalg.norm(out[0] - xo) <= 1e-9 * scale
    # same arithmetic per cell in both kernels (only FMA contraction may differ)
    assert np.linalg.norm(out[2] - out[0]) <= 1e-13 * scale
    P.close()


@pytest.mark.parametrize("pc", [fo.PC_JACOBI, fo.PC_NONE])
def test_fused_steps_without_poll_and_without_jacobi(pc):
    """check_every < 0: the host never waits (the smoother's way of calling); also the un-preconditioned variant."""
    n, bc = (70, 40, 24), [V, V, PER, PER, SYM, V]
    P, g = make_pair(n, bc, kappa=1e-3, nonuniform=True)
    S = g.assemble_S()
    _, b = mean_free_rhs(S, g.ncell)
    lam = S.gershgorin(pc)
    xo, _ = S.solve(b, ksp=fo.KSP_CHEBYSHEV, pc=pc, norm=fo.NORM_NONE, nullspace=True, maxit=9, emin=0.1 * lam, emax=1.1 * lam)
    _fuse(2)
    xg, ig = P.solve(dev(b), type=2, pc=pc, norm_type=fo.NORM_NONE, remove_nullspace=1, maxit=9, emin=0.1 * lam, emax=1.1 * lam, check_every=-1)
    assert ig["iters"] == 9 and ig["reason"] == 4
    assert np.linalg.norm(host(xg) - xo) <= 1e-9 * np.linalg.norm(xo)
    P.close()


def test_fused_steps_then_a_tested_solve_on_the_same_handle():
    """The fused sweep flips the d buffer; a solve WITH a norm (one-step kernel, convergence test) must start clean after it."""
    P, g = make_pair((40, 36, 34), CAVITY, kappa=1e-3)
    S = g.assemble_S()
    _, b = mean_free_rhs(S, g.ncell)
    lam = S.gershgorin(fo.PC_JACOBI)
    kw = dict(type=2, pc=fo.PC_JACOBI, remove_nullspace=1, emin=0.1 * lam, emax=1.1 * lam)
    _fuse(2)
    P.solve(dev(b), norm_type=fo.NORM_NONE, maxit=6, **kw)     # leaves dcur = 1 behind on the device only if not reset
    xo, io = S.solve(b, ksp=fo.KSP_CHEBYSHEV, pc=fo.PC_JACOBI, norm=fo.NORM_PRECONDITIONED, nullspace=True, rtol=1e-2, maxit=80, emin=0.1 * lam, emax=1.1 * lam)
    xg, ig = P.solve(dev(b), history=True, norm_type=fo.NORM_PRECONDITIONED, rtol=1e-2, maxit=80, check_every=7, **kw)
    assert ig["reason"] == io["reason"] and ig["iters"] == io["iters"]
    assert np.linalg.norm(host(xg) - xo) <= 1e-9 * np.linalg.norm(xo)
    P.close()


def test_tuning_knob_rejects_unknown_names():
    from fluca_amd import capi
    assert capi.lib.fl_tuning_set(b"no_such_knob", 1) == -62
    v = C.c_int(-1)
    _fuse(2)
    capi.check(capi.lib.fl_tuning_get(b"cheb_fuse", C.byref(v)))
    assert v.value == 2
