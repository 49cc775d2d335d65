"""CPU: the oracle's restatement of the momentum block A = I + dt C - (mu dt / 2 rho) L.

Pins: the second-derivative rows against the reference's FlucaFD golden files (the NS assembly and FlucaFD produce the
same one-sided Dirichlet and central rows); the convection rows have no golden in the reference -- they are checked by
the properties the formula (C v)_c = 1/2 d/dx_d (v_c V0_d + v0interp_c v_d) implies.
"""
import os

import numpy as np
import pytest

from oracle import fluca_oracle as fo
from tests.flucafd_golden import fmt_g, parse

V, O, PER, SYM = fo.BC_VELOCITY, fo.BC_PRESSURE_OUTLET, fo.BC_PERIODIC, fo.BC_SYMMETRY
UNIT = [(0.0, 1.0)] * 3


def _golden_rows(golden_dir, name):
    return parse(os.path.join(golden_dir, "flucafd", name + ".out"))[1]


def test_lap_central_row_matches_flucafd(golden_dir):
    rows = _golden_rows(golden_dir, "ex1_second_deriv")      # stencil at i=4 of an 8-cell unit grid
    g = fo.Grid.uniform((8, 3, 3), UNIT, [V] * 6)
    got = {4 + off: fmt_g(v) for off, v in g.lap_row(0, 4, 0)}
    assert got == {r["i"]: r["v_text"] for r in rows}


def test_lap_dirichlet_row_matches_flucafd(golden_dir):
    rows = _golden_rows(golden_dir, "ex1_second_deriv_left_bc_dirichlet")   # i=0, left Dirichlet
    g = fo.Grid.uniform((8, 3, 3), UNIT, [V] * 6)
    elem = {r["i"]: r["v_text"] for r in rows if r["loc"] == "ELEMENT"}
    for c in range(3):          # VELOCITY: all three components take the Dirichlet row
        assert {off: fmt_g(v) for off, v in g.lap_row(0, 0, c)} == elem
    gs = fo.Grid.uniform((8, 3, 3), UNIT, [SYM, V, V, V, V, V])
    assert {off: fmt_g(v) for off, v in gs.lap_row(0, 0, 0)} == elem        # SYMMETRY: normal component is Dirichlet
    assert len(gs.lap_row(0, 0, 1)) == 2                                    # tangential: two-point Neumann row
    # mirror image on the right wall
    r = g.lap_row(0, 7, 0)
    assert [fmt_g(v) for _, v in r] == [elem[2], elem[1], elem[0]] and [o for o, _ in r] == [-2, -1, 0]


def _coords(n, stretch):
    if not stretch:
        return [np.linspace(0.0, 1.0, m + 1) for m in n]
    return [np.linspace(0.0, 1.0, m + 1) ** (1.0 + 0.3 * (d + 1)) for d, m in enumerate(n)]


@pytest.mark.parametrize("stretch", [False, True])
def test_lap_rows_exact_for_quadratics(stretch):
    n = (7, 6, 5)
    xf = _coords(n, stretch)
    g = fo.Grid(n, xf, [V, O, SYM, V, PER, PER])
    for d in range(3):
        xc = 0.5 * (xf[d][1:] + xf[d][:-1])
        for c in range(3):
            for i in range(n[d]):
                row = g.lap_row(d, i, c)
                if g.periodic[d] and (i == 0 or i == n[d] - 1):
                    continue
                lo_wall, hi_wall = i == 0, i == n[d] - 1
                bc = g.bc[2 * d] if lo_wall else (g.bc[2 * d + 1] if hi_wall else None)
                dirichlet = bc == V or (bc == SYM and c == d)
                xw = xf[d][0] if lo_wall else xf[d][-1]
                if bc is None:
                    # central row on a non-uniform grid: exact for linears, second-order for quadratics
                    assert abs(sum(v for _, v in row)) < 1e-9 * max(abs(v) for _, v in row)
                    assert abs(sum(v * xc[i + o] for o, v in row)) < 1e-8 * max(abs(v) for _, v in row)
                elif dirichlet:
                    # one-sided row through the wall value: exact for quadratics that vanish at the wall
                    f = lambda x: (x - xw) * (1.0 + 2.0 * (x - xw))
                    assert sum(v * f(xc[i + o]) for o, v in row) == pytest.approx(4.0, rel=1e-9)
                else:
                    # two-point zero-gradient row: annihilates constants
                    assert sum(v for _, v in row) == pytest.approx(0.0, abs=1e-9 * abs(row[0][1]))


def test_conv_rows_interior_are_half_face_flux_interpolation():
    n = (6, 5, 4)
    xf = _coords(n, True)
    g = fo.Grid(n, xf, [V] * 6)
    d, i, vf = 0, 3, 0.7
    h = xf[d][i + 1] - xf[d][i]
    xc = 0.5 * (xf[d][1:] + xf[d][:-1])
    lo, hi = g.conv_row(d, i, 0, False, vf), g.conv_row(d, i, 1, False, vf)
    assert [o for o, _ in lo] == [-1, 0] and [o for o, _ in hi] == [0, 1]
    assert sum(v for _, v in lo) == pytest.approx(-0.5 * vf / h)       # weights of a linear interpolation sum to one
    assert sum(v for _, v in hi) == pytest.approx(+0.5 * vf / h)
    # interpolation reproduces linear fields at the face
    assert sum(v * xc[i + o] for o, v in lo) == pytest.approx(-0.5 * vf / h * xf[d][i])
    assert sum(v * xc[i + o] for o, v in hi) == pytest.approx(+0.5 * vf / h * xf[d][i + 1])
    # VELOCITY wall faces carry no matrix entries (the boundary value goes to the right-hand side)
    assert g.conv_row(d, 0, 0, False, vf) == [] and g.conv_row(d, n[d] - 1, 1, True, vf) == []


def test_conv_outlet_rows_as_in_reference():
    """cartdiscret.c:335-371: zero-gradient extrapolation through the two cells next to the outlet.  The high-side row is
    +0.5 vf/h times extrapolation weights that sum to one; the low-side row, as written in the reference, sums to
    +0.5 vf/h as well (not -0.5 vf/h as a low face would).  The oracle restates it as it stands."""
    n = (6, 5, 4)
    xf = _coords(n, True)
    g = fo.Grid(n, xf, [O, O, SYM, SYM, V, V])
    vf = 1.3
    h0, h1 = xf[0][1] - xf[0][0], xf[0][-1] - xf[0][-2]
    lo, hi = g.conv_row(0, 0, 0, True, vf), g.conv_row(0, n[0] - 1, 1, True, vf)
    assert [o for o, _ in lo] == [0, 1] and [o for o, _ in hi] == [-1, 0]
    assert sum(v for _, v in hi) == pytest.approx(0.5 * vf / h1)
    assert sum(v for _, v in lo) == pytest.approx(0.5 * vf / h0)
    # SYMMETRY: no entries for the normal component, extrapolation for a tangential one
    assert g.conv_row(1, 0, 0, True, vf) == [] and len(g.conv_row(1, 0, 0, False, vf)) == 2
    assert g.conv_row(1, n[1] - 1, 1, True, vf) == [] and len(g.conv_row(1, n[1] - 1, 1, False, vf)) == 2


def _fields(g, seed=3):
    rng = np.random.default_rng(seed)
    V0 = [rng.standard_normal(g.nface[d]) for d in range(3)]
    W = [rng.standard_normal(g.nface[d]) for c in range(3) for d in range(3)]
    return V0, W


def test_periodic_uniform_convection_is_the_skew_form():
    """Uniform periodic grid, constant advecting fields: (C v)_c = 1/2 (U . grad_h v_c + U_c div_h v) with central
    differences -- the textbook discretisation of the formula quoted at cnlinearcart3d.c:920."""
    n = (8, 6, 5)
    g = fo.Grid.uniform(n, UNIT, [PER] * 6)
    U = np.array([0.7, -0.4, 1.1])
    V0 = [np.full(g.nface[d], U[d]) for d in range(3)]
    W = [np.full(g.nface[d], U[c]) for c in range(3) for d in range(3)]
    Cm = g.assemble_momentum(0.0, 1.0, 0.0, V0, W)
    rng = np.random.default_rng(1)
    v = rng.standard_normal((3, n[2], n[1], n[0]))
    h = [1.0 / n[0], 1.0 / n[1], 1.0 / n[2]]
    ax = [3, 2, 1]   # array axis of x, y, z in (c, k, j, i)
    cd = lambda a, d: (np.roll(a, -1, axis=ax[d] - 1) - np.roll(a, 1, axis=ax[d] - 1)) / (2 * h[d])
    div = sum(cd(v[d], d) for d in range(3))
    want = np.stack([0.5 * (sum(U[d] * cd(v[c], d) for d in range(3)) + U[c] * div) for c in range(3)])
    got = Cm.mult(v.ravel()).reshape(v.shape)
    assert np.allclose(got, want, rtol=1e-12, atol=1e-12)


@pytest.mark.parametrize("bc", [[V] * 6, [V, V, V, V, SYM, V], [PER, PER, V, O, SYM, SYM], [O, V, PER, PER, V, V]])
def test_assembled_A_is_identity_plus_scaled_parts(bc):
    n = (6, 5, 4)
    g = fo.Grid(n, _coords(n, True), bc)
    V0, W = _fields(g)
    dt, rho, mu = 0.01, 1.3, 0.02
    A = g.assemble_momentum(1.0, dt, -0.5 * mu * dt / rho, V0, W)
    Cm = g.assemble_momentum(0.0, 1.0, 0.0, V0, W)
    L = g.assemble_momentum(0.0, 0.0, 1.0)
    v = np.random.default_rng(2).standard_normal(3 * g.ncell)
    assert np.allclose(A.mult(v), v + dt * Cm.mult(v) - 0.5 * mu * dt / rho * L.mult(v), rtol=1e-12, atol=1e-12)
    assert A.nrow == 3 * g.ncell
    # L never couples components; C couples component c only to itself and to the face-normal ones
    rp, col, val = L.arrays()
    rows = np.repeat(np.arange(L.nrow), np.diff(rp))
    assert np.all((rows // g.ncell == col // g.ncell) | (val == 0.0))


def test_momentum_bcgs_solve_cpu():
    n = (8, 7, 6)
    g = fo.Grid(n, _coords(n, True), [V, V, V, V, SYM, V])
    V0, W = _fields(g)
    A = g.assemble_momentum(1.0, 0.02, -0.5 * 0.01 * 0.02, V0, W)
    b = np.random.default_rng(4).standard_normal(3 * g.ncell)
    x, info = A.solve(b, ksp=fo.KSP_BCGS, pc=fo.PC_JACOBI, nullspace=False, rtol=1e-10, maxit=200)
    assert info["reason"] > 0
    assert np.linalg.norm(b - A.mult(x)) <= 1e-8 * np.linalg.norm(b)


def test_T_rows():
    """Face-normal interpolation (cnlinearcart3d.c:1934-2140): linear interpolation inside, nothing on VELOCITY /
    SYMMETRY walls, zero-gradient extrapolation at outlets -- the high-side row as the reference writes it."""
    n = (8, 6, 5)
    g = fo.Grid.uniform(n, UNIT, [O, O, V, SYM, PER, PER])
    assert g.T_row(0, 3) == [(2, 0.5), (3, 0.5)]
    assert g.T_row(1, 0) == [] and g.T_row(1, n[1]) == []
    lo, hi = g.T_row(0, 0), g.T_row(0, n[0])
    assert [c for c, _ in lo] == [0, 1] and [c for c, _ in hi] == [n[0] - 2, n[0] - 1]
    assert lo[0][1] == pytest.approx(9 / 8) and lo[1][1] == pytest.approx(-1 / 8)      # exact extrapolation weights
    assert hi[0][1] == pytest.approx(-1 / 3) and hi[1][1] == pytest.approx(4 / 3)      # the reference's shifted arguments (:1993)
    assert sum(w for _, w in lo) == pytest.approx(1.0) and sum(w for _, w in hi) == pytest.approx(1.0)
    per = g.T_row(2, 0)
    assert [c for c, _ in per] == [-1, 0] and per[0][1] == pytest.approx(0.5)
    # stretched grid: interpolation reproduces linear fields at the face
    xf = _coords(n, True)
    gs = fo.Grid(n, xf, [V] * 6)
    xc = 0.5 * (xf[0][1:] + xf[0][:-1])
    for f in range(1, n[0]):
        assert sum(w * xc[c] for c, w in gs.T_row(0, f)) == pytest.approx(xf[0][f])
    v = np.concatenate([np.full(gs.ncell, 2.0), np.full(gs.ncell, -1.0), np.full(gs.ncell, 0.5)])
    Vf = gs.apply_T(v)
    assert np.allclose(np.unique(Vf[0]), [0.0, 2.0]) and np.allclose(np.unique(Vf[2]), [0.0, 0.5])


def test_B_rows():
    """Face interpolation of every component (cnlinearcart3d.c:1513-1747): like T, except that a SYMMETRY wall
    extrapolates the tangential components and the high-side outlet row has the intended arguments."""
    n = (8, 6, 5)
    g = fo.Grid.uniform(n, UNIT, [O, O, SYM, V, PER, PER])
    for c in range(3):
        assert g.B_row(0, 3, c) == [(2, 0.5), (3, 0.5)]
        lo, hi = g.B_row(0, 0, c), g.B_row(0, n[0], c)
        assert [w for _, w in lo] == pytest.approx([9 / 8, -1 / 8]) and [w for _, w in hi] == pytest.approx([-1 / 8, 9 / 8])
        assert [q for q, _ in hi] == [n[0] - 2, n[0] - 1]
        assert g.B_row(1, n[1], c) == []                                   # VELOCITY wall: boundary-condition vector
    assert g.B_row(1, 0, 1) == [] and len(g.B_row(1, 0, 0)) == 2 and len(g.B_row(1, 0, 2)) == 2    # SYMMETRY
    # B and T agree wherever T has a row, except at the high-side outlet
    for d in range(3):
        for f in range(g.nf[d]):
            if not (d == 0 and f == n[0]):
                assert g.B_row(d, f, d) == g.T_row(d, f)


def test_step_oracle_advances_taylor_green():
    """The CPU composition of one CNLinear step (StepOracle) on the reference's own accuracy test
    (fluca/tests/taylor_green_vortex/taylor_green_vortex.c), extended along a periodic z: with VELOCITY walls taken from
    the exact solution, two steps stay within the discretisation error of the exact solution."""
    n, L, nu, dt = 12, 2 * np.pi, 0.1, 0.05
    box = [(0, L), (0, L), (0, L * 4 / n)]
    g = fo.Grid.uniform((n, n, 4), box, [V, V, V, V, PER, PER], dt / 1.0)
    d = lambda t: np.exp(-2 * nu * t)
    ex = lambda x, y, t: (np.sin(x) * np.cos(y) * d(t), -np.cos(x) * np.sin(y) * d(t))

    def velocity(b, t, X):
        u, w = ex(X[:, 0], X[:, 1], t)
        return np.stack([u, w, np.zeros_like(u)])

    so = fo.StepOracle(g, dt, 1.0, nu, velocity, krylov_rtol=1e-11, outer_rtol=1e-8)
    h = L / n
    xc, xf = (np.arange(n) + 0.5) * h, np.arange(n + 1) * h
    Z = np.ones((4, 1, 1))
    fld = lambda xs, ys, t: [Z * a for a in ex(xs[None, None, :], ys[None, :, None], t)]
    u0, w0 = fld(xc, xc, 0.0)
    v = np.concatenate([u0.ravel(), w0.ravel(), np.zeros(g.ncell)])
    Vf = [fld(xf, xc, 0.0)[0].ravel(), fld(xc, xf, 0.0)[1].ravel(), np.zeros(g.nface[2])]
    X, Y = np.meshgrid(xc, xc, indexing="xy")
    p = (Z * (0.25 * (np.cos(2 * X) + np.cos(2 * Y)))[None]).ravel()
    for _ in range(2):
        v, Vf, p, info = so.step_once(v, Vf, p)
        assert info["outer_its"] < 40
    ue, we = fld(xc, xc, 2 * dt)
    vh = v.reshape(3, 4, n, n)
    err = np.sqrt(((vh[0] - ue) ** 2 + (vh[1] - we) ** 2).mean())
    assert err < 5e-3 and np.abs(vh[2]).max() < 1e-12
    assert np.abs(g.rhs(*Vf)).max() < 1e-6          # discretely divergence-free face velocity
