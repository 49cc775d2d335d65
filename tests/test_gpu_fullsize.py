"""-m gpu: the BASELINE configurations bench.py times, checked at their OWN size (512^3, 134 M cells).

C3  512^3 channel [VELOCITY inlet, PRESSURE_OUTLET, walls in y, PERIODIC span], Chebyshev-Jacobi sweeps: the fused two-step kernel
    (k_cheb2: a periodic seam and an outlet across 4 x 64 tiles, several z chunks) against the one-step kernel on the LDS-staged walk
    (k_cheb_st) on the same handle -- the same arithmetic per cell, so 1e-13 -- and the damping a sweep must deliver.
C4  512^3 with the immersed sphere of diameter 64 h (12 868 markers on a Fibonacci lattice): fl_ibm_interp / fl_ibm_spread against the
    oracle's fo_ibm_interp / fo_ibm_spread.  The host cost is O(markers), not O(cells): the oracle loops over the markers' supports.
momentum  the same 512^3 channel: the momentum block with v0interp formed inside the kernel (k_mom3) against the stored-path kernel.
Against the ORACLE at full size (round 4; the host has the memory for the assembled matrices: 12 GB for a 512^3 S, 8 GB for a 256^3 A) --
the same functions bench.py prints as parity_on_full_grid / configs.C3.parity_on_full_grid / configs.momentum.parity:
headline  512^3 cavity Jacobi-PCG, the driver's 20 iterations, against the oracle's assembled CSR + KSPCG restatement;
C3        512^3 channel, 20 Chebyshev-Jacobi steps of the fused kernel against the oracle's KSPCHEBYSHEV on the assembled channel S;
momentum  256^3 cavity: MatMult(A), diag(A) and five Jacobi-BiCGStab iterations against the oracle's assembled A (50 M rows).
"""
import os
import sys

import numpy as np
import pytest
import torch

from oracle import fluca_oracle as fo
from tests.gpu_common import CAVITY_BOX, O, PER, V
from tests.test_gpu_ibm import Ibm, sphere_markers

pytestmark = pytest.mark.gpu
N512 = (512, 512, 512)


def _fuse(mode):
    from fluca_amd import capi
    capi.check(capi.lib.fl_tuning_set(b"cheb_fuse", mode), "fl_tuning_set")


def test_c3_channel_fused_chebyshev_equals_single_steps_at_512():
    from fluca_amd.poisson import Poisson
    bc = [V, O, V, V, PER, PER]
    P = Poisson.uniform(N512, CAVITY_BOX, bc, 1e-3)
    gen = torch.Generator(device="cuda").manual_seed(3)
    p = torch.rand(P.ncell, generator=gen, dtype=torch.float64, device="cuda") * 2 - 1
    b = P.apply(p)
    kw = dict(type=2, norm_type=3, remove_nullspace=0, check_every=100)
    out = {}
    try:
        for mode in (0, 2):
            _fuse(mode)
            x, info = P.solve(b, maxit=20, profile=1, **kw)
            assert info["iters"] == 20 and info["reason"] == 4
            # which kernel ran: the fused one needs half as many launches as steps
            assert info["kernel_launches"] == (10 if mode == 2 else 20), info
            out[mode] = x
    finally:
        _fuse(1)
    scale = float(out[0].norm())
    assert scale > 0 and float((out[2] - out[0]).norm()) <= 1e-13 * scale
    # residual damping: 40 steps leave less than 20, both less than the right-hand side (zero initial guess)
    r20 = float((b - P.apply(out[2])).norm())
    x40, _ = P.solve(b, maxit=40, **kw)
    r40 = float((b - P.apply(x40)).norm())
    assert r40 < r20 < float(b.norm())
    P.close()


@pytest.mark.parametrize("kind", [fo.DELTA_PESKIN4])
def test_c4_sphere_ibm_matches_oracle_at_512(kind):
    from fluca_amd.poisson import Poisson
    box = [(0.0, 1.0)] * 3
    bc = [V] * 6
    P = Poisson.uniform(N512, box, bc, 1e-3)
    g = fo.Grid.uniform(N512, box, bc, 1e-3)
    h = 1.0 / 512
    R = 32 * h
    L = int(round(4 * np.pi * R * R / (h * h)))
    assert L == 12868
    X = sphere_markers(L, (0.5, 0.5, 0.5), R)
    m = Ibm(P, kind, X)
    gen = torch.Generator(device="cuda").manual_seed(4)
    u = torch.rand(3 * P.ncell, generator=gen, dtype=torch.float64, device="cuda") * 2 - 1
    U = m.interp(u, 3).cpu().numpy().reshape(3, L)
    uh = u.cpu().numpy()
    Uo = g.ibm_interp(kind, X, uh)
    assert np.abs(U - Uo).max() <= 1e-13 * np.abs(Uo).max()
    rng = np.random.default_rng(5)
    F = rng.standard_normal((3, L))
    dV = np.full(L, h ** 3)
    f = torch.zeros(3 * P.ncell, dtype=torch.float64, device="cuda")
    m.spread(torch.as_tensor(F.reshape(-1), device="cuda"), torch.as_tensor(dV, device="cuda"), f, 3)
    fo_ = g.ibm_spread(kind, X, dV, F)
    fh = f.cpu().numpy().reshape(3, -1)
    assert np.abs(fh - fo_).max() <= 1e-13 * np.abs(fo_).max()
    # the invariants of SURVEY 8(c) at this size: spreading conserves the total force; interpolation and spreading are adjoint
    assert np.allclose(fh.sum(axis=1) * h ** 3, (F * dV).sum(axis=1), rtol=1e-11)
    lhs, rhs = float((U * F * dV).sum()), float((uh.reshape(3, -1) * fh).sum() * h ** 3)
    assert abs(lhs - rhs) <= 1e-11 * max(abs(lhs), abs(rhs))
    m.close()
    P.close()


def test_momentum_state_with_v0_equals_the_stored_fields_at_512():
    """The momentum block of bench.py's `configs.momentum` at its own size (512^3 channel with an outlet and a periodic span, 403 M velocity
    unknowns): k_mom3 (v0interp formed from v0 inside the kernel, fl_momentum_set_state_v0) against k_mom2 reading the nine stored fields
    of the same state -- the same products in the same order per face value, so the two operators agree to the rounding of the row sums --
    and the Jacobi-BiCGStab histories of the two paths."""
    from fluca_amd.poisson import Momentum, Poisson
    P = Poisson.uniform(N512, CAVITY_BOX, [V, O, V, V, PER, PER], 1e-3)
    M = Momentum(P)
    gen = torch.Generator(device="cuda").manual_seed(17)
    rnd = lambda m: torch.rand(m, dtype=torch.float64, device="cuda", generator=gen) * 2 - 1  # noqa: E731
    V0 = [rnd(P.nface[d]) for d in range(3)]
    v0 = rnd(3 * P.ncell)
    W = M.interp_faces(v0)
    h = 1.0 / 512
    x = rnd(3 * P.ncell)
    M.set_state(0.5 * h, 1.0, 0.5 * h, V0, W)
    y_stored, d_stored = M.apply(x), M.diagonal()
    s_stored, i_stored = M.solve(x, history=True, rtol=1e-9, maxit=100)
    M.set_state(0.5 * h, 1.0, 0.5 * h, V0, W, v0=v0)
    del V0, W, v0
    y_fly, d_fly = M.apply(x), M.diagonal()
    assert float((y_fly - y_stored).abs().max()) <= 4e-15 * float(y_stored.abs().max())
    assert float((d_fly - d_stored).abs().max()) <= 4e-15 * float(d_stored.abs().max())
    del y_stored, d_stored, d_fly
    s_fly, i_fly = M.solve(x, history=True, rtol=1e-9, maxit=100)
    assert i_fly["reason"] == i_stored["reason"] == 2 and i_fly["iters"] == i_stored["iters"]
    assert np.allclose(i_fly["history"], i_stored["history"], rtol=1e-6)
    assert float((s_fly - s_stored).norm()) <= 1e-9 * float(s_stored.norm())
    # and the answer solves the system
    assert float((M.apply(s_fly) - x).norm()) <= 1e-7 * float(x.norm())
    M.close()
    P.close()


def _bench():
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    if root not in sys.path:
        sys.path.insert(0, root)
    import bench
    return bench


def _needs_host_memory(gb):
    avail = _bench()._host_mem_available_gb()
    if not (avail and avail >= gb):
        pytest.skip(f"host memory {avail} GB < {gb} GB for the assembled oracle matrix")


def test_headline_512_cavity_pcg_matches_oracle():
    """BASELINE's metric configuration at its own size: 20 Jacobi-PCG iterations (the driver's K) on the 512^3 cavity Schur complement,
    HIP path against the oracle's assembled S + KSPCG: same iteration count, residual norms to 1e-10, x to 1e-12."""
    _needs_host_memory(32.0)
    bench = _bench()
    bench._oracle_threads()
    leg = bench._cpu_leg(512, 20)
    par = leg["parity"]
    assert "error" not in par, par
    assert par["iters_gpu"] == par["iters_cpu"] == 20
    assert par["rel_max_diff_x"] <= 1e-12, par
    assert abs(par["rnorm_gpu"] - par["rnorm_cpu"]) <= 1e-10 * par["rnorm_cpu"], par


def test_c3_channel_chebyshev_matches_oracle_at_512():
    from fluca_amd.poisson import Poisson
    _needs_host_memory(32.0)
    bench = _bench()
    bc = [V, O, V, V, PER, PER]
    P = Poisson.uniform(N512, CAVITY_BOX, bc, 1e-3)
    gen = torch.Generator(device="cuda").manual_seed(3)
    p = torch.rand(P.ncell, generator=gen, dtype=torch.float64, device="cuda") * 2 - 1
    b = P.apply(p)
    del p
    par = bench.c3_parity(P, b, CAVITY_BOX, bc, steps=20)
    P.close()
    assert par.get("fused_kernel") is True, par             # the kernel the C3 bench line times
    assert par["steps_gpu"] == par["steps_cpu"] == 20 and par["reason_gpu"] == par["reason_cpu"] == 4
    assert par["rel_max_diff_x"] <= 1e-12 and par["rel_l2_diff_x"] <= 1e-12, par


def test_momentum_block_matches_the_assembled_oracle_at_256():
    _needs_host_memory(40.0)
    bench = _bench()
    par = bench.momentum_parity(n1=256, its=5)
    assert par["rows"] == 3 * 256 ** 3
    assert par["rel_max_diff_apply"] <= 1e-12 and par["rel_max_diff_diag"] <= 1e-12, par
    assert par["bcgs_iters_gpu"] == par["bcgs_iters_cpu"] == 5
    assert par["rel_max_diff_history"] <= 1e-9 and par["rel_max_diff_x"] <= 1e-9, par
