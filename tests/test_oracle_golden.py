"""Pins the CPU oracle's operator coefficients to the reference's own golden files.

The files under tests/golden/flucafd/ are verbatim copies of DATA files of the
reference's test-suite (fluca/tests/fd/output/<name>.out): stdout of FlucaFD
stencil queries on an 8-cell (16 when "refined") unit-interval grid.  FlucaFD is
an independent code path of the reference that yields the same D, Gst and
D o Gst numbers as the NS assembly restated by the oracle.
"""
import os

import numpy as np
import pytest

from oracle import fluca_oracle as fo
from tests.flucafd_golden import fmt_g, parse

V, O, PER, SYM = fo.BC_VELOCITY, fo.BC_PRESSURE_OUTLET, fo.BC_PERIODIC, fo.BC_SYMMETRY
UNIT = [(0.0, 1.0)] * 3


def G(golden_dir, name):
    return parse(os.path.join(golden_dir, "flucafd", name + ".out"))[1]


def grid(n, bc, kappa=1.0, box=UNIT):
    return fo.Grid.uniform(n, box, bc, kappa)


def test_gst_interior_row(golden_dir):
    rows = G(golden_dir, "ex1_first_deriv_input_loc_elem_output_loc_left")
    got = grid((8, 1, 1), [V] * 6).gst_row(0, 4)
    assert [(r["i"], r["v_text"]) for r in rows] == [(c, fmt_g(v)) for c, v in got]


def test_gst_periodic_row(golden_dir):
    rows = G(golden_dir, "ex1_first_deriv_input_loc_elem_output_loc_left_left_bc_periodic")
    got = grid((8, 1, 1), [PER, PER, V, V, V, V]).gst_row(0, 0)
    assert [(r["i"], r["v_text"]) for r in rows] == [(c, fmt_g(v)) for c, v in got]


def test_gst_wall_row_is_pure_boundary_term(golden_dir):
    rows = G(golden_dir, "ex1_first_deriv_input_loc_elem_output_loc_left_left_bc_neumann")
    # FlucaFD: the only column is the prescribed boundary gradient itself -> no interior columns,
    # which is the empty Gst row of cnlinearcart3d.c:2449-2452
    assert [r for r in rows if not r["c"].endswith("_boundary")] == []
    for bc in (V, SYM):
        assert grid((8, 1, 1), [bc, V, V, V, V, V]).gst_row(0, 0) == []
    assert grid((8, 1, 1), [V] * 6).gst_row(0, 8) == []


def test_divergence_row(golden_dir):
    rows = G(golden_dir, "ex2_all_first_deriv_input_loc_face_output_loc_elem")
    g = grid((8, 8, 8), [V] * 6)
    want = {}
    for r in rows:
        d = {"LEFT": 0, "DOWN": 1, "BACK": 2}[r["loc"]]
        want[(d, (r["i"], r["j"], r["k"])[d])] = r["v_text"]
    got = {}
    for d in range(3):
        for f, v in g.div_row(d, 4):
            got[(d, f)] = fmt_g(v)
    assert got == want


def _row(S, g, ijk):
    rp, col, val = S.arrays()
    M, N, P = g.n
    r = (ijk[2] * N + ijk[1]) * M + ijk[0]
    out = {}
    for p in range(rp[r], rp[r + 1]):
        c = int(col[p])
        out[(c % M, (c // M) % N, c // (M * N))] = val[p]
    return out


@pytest.mark.parametrize("name,bc,cell", [
    ("ex2_all_second_deriv", [V] * 6, (4, 4, 4)),
    ("ex2_all_second_deriv_up_bc_neumann", [V] * 6, (4, 7, 4)),
    ("ex2_all_second_deriv_back_bc_periodic", [V, V, V, V, PER, PER], (4, 4, 0)),
])
def test_schur_row_equals_flucafd_laplacian(golden_dir, name, bc, cell):
    """S = -kappa * D*Gst (abfpc.c:150-171): with kappa=1, -S must print exactly like FlucaFD's Laplacian row."""
    rows = [r for r in G(golden_dir, name) if not r["c"].endswith("_boundary")]
    g = grid((8, 8, 8), bc)
    got = _row(g.assemble_S(), g, cell)
    want = {(r["i"] % 8, r["j"] % 8, r["k"] % 8): r["v_text"] for r in rows}
    assert {k: fmt_g(-v) for k, v in got.items()} == want


@pytest.mark.parametrize("name,n,i", [("ex1_second_deriv", 8, 4), ("ex1_second_deriv_refined", 16, 8),
                                      ("ex4_second_deriv_compact", 8, 4)])
def test_1d_second_derivative(golden_dir, name, n, i):
    rows = G(golden_dir, name)
    g = grid((n, 1, 1), [V] * 6)
    got = _row(g.assemble_S(), g, (i, 0, 0))
    assert {k[0]: fmt_g(-v) for k, v in got.items()} == {r["i"]: r["v_text"] for r in rows}


# ---- properties the goldens do not cover -------------------------------------------

def stretched(n, lo, hi, beta=1.7):
    s = np.linspace(0.0, 1.0, n + 1)
    return lo + (hi - lo) * (np.tanh(beta * (2 * s - 1)) / np.tanh(beta) + 1) / 2


@pytest.mark.parametrize("bc", [[V] * 6, [V, V, V, V, SYM, V], [PER, PER, V, V, V, V], [PER] * 6])
def test_S_neumann_properties(bc):
    g = grid((6, 5, 4), bc, kappa=0.37, box=[(0, 1), (0, 1), (0, 0.5)])
    A = g.assemble_S().to_scipy()
    assert abs(A @ np.ones(g.ncell)).max() < 1e-10                     # S 1 = 0 (constant null space)
    assert abs(A - A.T).max() < 1e-10                                  # symmetric on uniform grids
    w = np.linalg.eigvalsh(A.toarray())
    assert w.min() > -1e-9 and (w < 1e-9).sum() == 1                   # PSD, one-dimensional kernel
    diag = A.diagonal()
    assert (diag > 0).all()


def test_S_nonuniform_rowsum_and_outlet():
    xf = [stretched(7, 0, 2), stretched(5, 0, 1), stretched(6, -1, 1)]
    g = fo.Grid((7, 5, 6), xf, [V, O, V, V, PER, PER], kappa=0.5)
    A = g.assemble_S().to_scipy()
    rs = np.asarray(A.sum(axis=1)).ravel().reshape(6, 5, 7)
    assert abs(rs[:, :, :-1]).max() < 1e-9      # all rows not touching the outlet sum to zero
    assert (rs[:, :, -1] > 0).all()             # outlet rows: Dirichlet -> strictly positive row sum -> S nonsingular
    assert np.linalg.matrix_rank(A.toarray()) == g.ncell


def test_outlet_face_gradient_exact_for_quadratics():
    """The 2-pt one-sided rows + boundary term (cartdiscret.c:425-476, cnlinearcart3d.c:2643-2674) differentiate
    any quadratic exactly at the boundary face."""
    xf = stretched(9, -0.3, 1.1)
    g = fo.Grid((9, 1, 1), [xf, np.array([0.0, 1.0]), np.array([0.0, 1.0])], [O, O, V, V, V, V])
    xc = 0.5 * (xf[1:] + xf[:-1])
    q = lambda x: 0.7 - 1.3 * x + 2.1 * x * x
    dq = lambda x: -1.3 + 4.2 * x
    for side, f, xb in ((0, 0, xf[0]), (1, 9, xf[-1])):
        val = sum(v * q(xc[c]) for c, v in g.gst_row(0, f)) + g.gst_bc_coeff(0, side) * q(xb)
        assert abs(val - dq(xb)) < 1e-11


def test_assembled_S_matches_composed_operators():
    """D * (kappa Gst p) computed matrix-free equals -S p (two independent routes through the oracle)."""
    rng = np.random.default_rng(1)
    xf = [stretched(6, 0, 1), stretched(7, 0, 1), stretched(5, 0, 1)]
    for bc in ([V] * 6, [V, O, V, V, PER, PER], [O, V, SYM, V, V, O]):
        g = fo.Grid((6, 7, 5), xf, bc, kappa=0.8)
        p = rng.standard_normal(g.ncell)
        Gp = g.apply_gst(p)
        lhs = g.rhs(*Gp)            # = 0 - D (kappa Gst p)
        rhs = g.assemble_S().mult(p)
        assert np.allclose(lhs, rhs, rtol=0, atol=1e-11 * abs(rhs).max())
