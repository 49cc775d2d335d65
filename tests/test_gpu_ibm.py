"""-m gpu: IBM interpolation / spreading kernels against the CPU oracle and the analytic invariants of the spec
(no reference implementation exists: SURVEY section 0 / 8a row I1)."""
import ctypes as C

import numpy as np
import pytest
import torch

from oracle import fluca_oracle as fo
from tests.gpu_common import PER, V, dev, host, make_pair

pytestmark = pytest.mark.gpu


def sphere_markers(L, c, R, seed=0):
    i = np.arange(L) + 0.5
    phi = np.arccos(1 - 2 * i / L)
    th = np.pi * (1 + 5 ** 0.5) * i
    return [c[0] + R * np.cos(th) * np.sin(phi), c[1] + R * np.sin(th) * np.sin(phi), c[2] + R * np.cos(phi)]


class Ibm:
    def __init__(self, P, kind, X):
        from fluca_amd.capi import check, lib
        self.lib, self.check, self.P = lib, check, P
        self.L = X[0].size
        self.Xd = [dev(a) for a in X]
        h = C.c_void_p()
        P._pre()
        check(lib.fl_ibm_create(P.h, kind, self.L, *[C.c_void_p(t.data_ptr()) for t in self.Xd], C.byref(h)), "fl_ibm_create")
        self.h = h

    def interp(self, u, ncomp):
        U = torch.empty(ncomp * self.L, dtype=torch.float64, device="cuda")
        self.P._pre()
        self.check(self.lib.fl_ibm_interp(self.h, ncomp, C.c_void_p(u.data_ptr()), C.c_void_p(U.data_ptr())))
        self.P._post()
        return U

    def spread(self, F, dV, f, ncomp):
        self.P._pre()
        self.check(self.lib.fl_ibm_spread(self.h, ncomp, C.c_void_p(F.data_ptr()), C.c_void_p(dV.data_ptr()), C.c_void_p(f.data_ptr())))
        self.P._post()
        return f

    def close(self):
        self.lib.fl_ibm_destroy(self.h)


@pytest.mark.parametrize("kind", [fo.DELTA_PESKIN4, fo.DELTA_ROMA3])
@pytest.mark.parametrize("bc,n", [([V] * 6, (24, 20, 18)), ([PER] * 6, (16, 16, 16)), ([PER, PER, V, V, V, V], (33, 17, 9))])
def test_interp_spread_match_oracle_and_invariants(kind, bc, n):
    box = [(0.0, 1.0), (0.0, 1.0), (0.0, 1.0)]
    P, g = make_pair(n, bc, box=box)
    rng = np.random.default_rng(4)
    L = 300
    X = sphere_markers(L, (0.5, 0.5, 0.5), 0.3)
    # a few markers near / across the boundary: clipped at walls, wrapped on periodic axes
    X[0][:5] = [0.01, 0.99, 0.5, 0.02, 0.97]
    X[1][:5] = [0.5, 0.5, 0.01, 0.98, 0.03]
    X[2][:5] = [0.03, 0.5, 0.99, 0.5, 0.5]
    m = Ibm(P, kind, X)
    u = rng.standard_normal((3, g.ncell))
    U = host(m.interp(dev(u), 3)).reshape(3, L)
    assert np.allclose(U, g.ibm_interp(kind, X, u), rtol=1e-12, atol=1e-13)
    F = rng.standard_normal((3, L))
    dV = rng.uniform(0.5, 1.5, L) * 1e-3
    f0 = rng.standard_normal((3, g.ncell))
    f = host(m.spread(dev(F), dev(dV), dev(f0), 3)).reshape(3, -1)
    ref = g.ibm_spread(kind, X, dV, F, f0.copy())
    assert np.allclose(f, ref, rtol=1e-12, atol=1e-12 * abs(ref).max())
    # bitwise reproducible (gather form, id-sorted bins)
    f2 = host(m.spread(dev(F), dev(dV), dev(f0), 3)).reshape(3, -1)
    assert np.array_equal(f, f2)
    # adjointness <interp(u), F dV> = <u, spread(F)> h^3 for markers whose support is inside / wrapped
    hvol = np.prod([(b[1] - b[0]) / n[d] for d, b in enumerate(box)])
    fz = host(m.spread(dev(F), dev(dV), torch.zeros(3 * g.ncell, dtype=torch.float64, device="cuda"), 3)).reshape(3, -1)
    lhs = (U * F * dV).sum()
    rhs = (u * fz).sum() * hvol
    assert abs(lhs - rhs) <= 1e-12 * max(abs(lhs), abs(rhs), 1.0)
    m.close()
    P.close()


@pytest.mark.parametrize("kind", [fo.DELTA_PESKIN4, fo.DELTA_ROMA3])
def test_moment_conditions_on_device(kind):
    """sum delta h^3 = 1 and exact interpolation of linear fields for markers away from walls; spreading conserves sum F"""
    n = (20, 20, 20)
    box = [(0.0, 2.0), (0.0, 2.0), (0.0, 2.0)]
    P, g = make_pair(n, [V] * 6, box=box)
    rng = np.random.default_rng(9)
    L = 200
    X = [rng.uniform(0.4, 1.6, L) for _ in range(3)]
    m = Ibm(P, kind, X)
    xc = [0.5 * (a[1:] + a[:-1]) for a in g.xf]
    Z, Y, Xg = np.meshgrid(xc[2], xc[1], xc[0], indexing="ij")
    u = np.stack([np.ones(g.ncell), (0.3 + 1.1 * Xg - 0.7 * Y + 2.0 * Z).ravel()])
    U = host(m.interp(dev(u), 2)).reshape(2, L)
    assert np.allclose(U[0], 1.0, rtol=0, atol=1e-13)
    assert np.allclose(U[1], 0.3 + 1.1 * X[0] - 0.7 * X[1] + 2.0 * X[2], rtol=0, atol=1e-12)
    F = rng.standard_normal((1, L))
    dV = np.full(L, 0.01)
    f = host(m.spread(dev(F), dev(dV), torch.zeros(g.ncell, dtype=torch.float64, device="cuda"), 1))
    hvol = (2.0 / 20) ** 3
    assert abs(f.sum() * hvol - (F[0] * dV).sum()) <= 1e-13 * abs(F[0] * dV).sum() + 1e-15
    m.close()
    P.close()


@pytest.mark.parametrize("kind", [fo.DELTA_PESKIN4, fo.DELTA_ROMA3])
@pytest.mark.parametrize("bc,n", [([V] * 6, (24, 20, 18)), ([PER, PER, V, V, PER, PER], (33, 17, 16)), ([V, V, V, V, PER, PER], (140, 12, 9))])
def test_stretched_grids_match_oracle_and_invariants(kind, bc, n):
    """Stretched axes (wall-refined channels, BASELINE configs 3-5): the delta function in index space, spreading per target-cell
    volume.  Against the oracle, and the three invariants that hold on ANY grid: sum of the weights = 1 (a constant is interpolated
    exactly), <interp(u), F dV> = sum_i u_i f_i V_i (adjointness with the cell volumes), sum_i f_i V_i = sum F dV (force conserved)."""
    from fluca_amd.poisson import Poisson
    from tests.gpu_common import stretched
    box = [(0.0, 1.0), (0.0, 2.0), (0.0, 1.5)]
    xf = [stretched(n[d], box[d][0], box[d][1], 1.4 + 0.3 * d) for d in range(3)]
    if bc[4] == PER:
        xf[2] = np.linspace(box[2][0], box[2][1], n[2] + 1)              # one uniform axis next to two stretched ones
    P, g = Poisson(n, xf, bc, 1e-3), fo.Grid(n, xf, bc, 1e-3)
    rng = np.random.default_rng(4)
    L = 400
    X = sphere_markers(L, (0.5, 1.0, 0.75), 0.35)
    X[0][:6] = [0.003, 0.997, 0.5, 0.02, 0.97, 0.5]                      # next to the walls (mirror-image ghost centres) / the seam
    X[1][:6] = [1.0, 1.0, 0.004, 1.99, 0.03, 1.0]
    X[2][:6] = [0.03, 0.7, 1.49, 0.7, 0.7, 0.001]
    m = Ibm(P, kind, X)
    u = rng.standard_normal((3, g.ncell))
    U = host(m.interp(dev(u), 3)).reshape(3, L)
    assert np.allclose(U, g.ibm_interp(kind, X, u), rtol=1e-12, atol=1e-13)
    F = rng.standard_normal((3, L))
    dV = rng.uniform(0.5, 1.5, L) * 1e-3
    f0 = rng.standard_normal((3, g.ncell))
    f = host(m.spread(dev(F), dev(dV), dev(f0), 3)).reshape(3, -1)
    ref = g.ibm_spread(kind, X, dV, F, f0.copy())
    assert np.allclose(f, ref, rtol=1e-12, atol=1e-12 * abs(ref).max())
    # invariants, for the markers whose whole support lies inside the domain or wraps (drop the six wall-hugging ones)
    inner = slice(6, None)
    one = host(m.interp(dev(np.ones(g.ncell)), 1))
    assert np.allclose(one[inner], 1.0, rtol=0, atol=1e-13)
    vol = np.einsum("k,j,i->kji", np.diff(xf[2]), np.diff(xf[1]), np.diff(xf[0])).ravel()
    Fi, dVi = F.copy(), dV.copy()
    Fi[:, :6] = 0.0
    fz = host(m.spread(dev(Fi), dev(dVi), torch.zeros(3 * g.ncell, dtype=torch.float64, device="cuda"), 3)).reshape(3, -1)
    lhs = (U * Fi * dVi).sum()
    rhs = (u * fz * vol).sum()
    assert abs(lhs - rhs) <= 1e-12 * max(abs(lhs), abs(rhs), 1.0)
    assert np.allclose((fz * vol).sum(axis=1), (Fi * dVi).sum(axis=1), rtol=1e-12, atol=1e-15)
    m.close()
    P.close()
