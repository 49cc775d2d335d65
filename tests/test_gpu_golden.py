"""-m gpu: the HIP kernels pinned DIRECTLY to the reference's FlucaFD golden files (tests/golden/flucafd/, verbatim copies of
fluca/tests/fd/output/*.out), without the oracle in between.

Rows are extracted from the device operators by applying them to unit vectors through the C-ABI on the goldens' own grids
(8 cells per axis on the unit interval, 16 for "refined", kappa = 1) and are compared TEXTUALLY with what the reference
printed ("%g" with PETSc's trailing point), exactly as tests/test_oracle_golden.py does for the CPU oracle.

golden file                                                   device operator (C-ABI entry)            reference rows
ex1_first_deriv_input_loc_elem_output_loc_left[...]           Gst   fl_poisson_project (faces)         cnlinearcart3d.c:2410-2600
ex2_all_first_deriv_input_loc_face_output_loc_elem            D     fl_poisson_rhs                     cnlinearcart3d.c:2314-2408
ex2_all_second_deriv[_up_bc_neumann|_back_bc_periodic],
ex1_second_deriv[_refined], ex4_second_deriv_compact          S     fl_poisson_apply                   abfpc.c:150-171
ex1_first_deriv                                               G     fl_poisson_project (cells)         cnlinearcart3d.c:4-217
ex1_second_deriv_left_bc_dirichlet                            L     fl_momentum_apply, coefficients 0,0,1  cnlinearcart3d.c:425-632
ex4_second_deriv                                              D T G fl_poisson_project + fl_momentum_face_interp + fl_poisson_rhs
                                                                    (the wide Laplacian inside S for PCABFAinvType DIAG / ROWSUM, abfpc.c:150-171)

Four more goldens of the same directory describe rows the NS assembly never forms (no boundary information: FlucaFD's "bc none"
one-sided rows, and its three-point Neumann row); test_rows_the_ns_assembly_does_not_form says which reference lines show that.
"""
import os

import numpy as np
import pytest
import torch

from oracle import fluca_oracle as fo
from tests.flucafd_golden import fmt_g, parse

pytestmark = pytest.mark.gpu
V, O, PER, SYM = fo.BC_VELOCITY, fo.BC_PRESSURE_OUTLET, fo.BC_PERIODIC, fo.BC_SYMMETRY
UNIT = [(0.0, 1.0)] * 3


def G(golden_dir, name):
    return parse(os.path.join(golden_dir, "flucafd", name + ".out"))[1]


def poisson(n, bc):
    from fluca_amd.poisson import Poisson
    return Poisson.uniform(n, UNIT, bc, 1.0)


def unit(n, c):
    e = torch.zeros(n, dtype=torch.float64, device="cuda")
    e[c] = 1.0
    return e


def nz_text(vals):
    """{index: printed value} of the non-zero entries"""
    return {int(i): fmt_g(float(vals[i])) for i in np.flatnonzero(vals)}


def gst_row(P, n, face):
    """row `face` of Gst along x on an (n,1,1) grid: V = 0 - kappa Gst e_c, kappa = 1"""
    row = np.zeros(n)
    for c in range(n):
        Vx = torch.zeros(P.nface[0], dtype=torch.float64, device="cuda")
        P.project(unit(n, c), V=(Vx, None, None))
        row[c] = -float(Vx[face])
    return row


def test_gst_rows(golden_dir):
    P = poisson((8, 1, 1), [V] * 6)
    rows = G(golden_dir, "ex1_first_deriv_input_loc_elem_output_loc_left")            # interior face 4
    assert nz_text(gst_row(P, 8, 4)) == {r["i"]: r["v_text"] for r in rows}
    rows = G(golden_dir, "ex1_first_deriv_input_loc_elem_output_loc_left_left_bc_neumann")
    assert [r for r in rows if not r["c"].endswith("_boundary")] == []                  # wall face: no interior column,
    assert nz_text(gst_row(P, 8, 0)) == {} and nz_text(gst_row(P, 8, 8)) == {}          # the empty rows of cnlinearcart3d.c:2449-2452
    P.close()
    P = poisson((8, 1, 1), [PER, PER, V, V, V, V])
    rows = G(golden_dir, "ex1_first_deriv_input_loc_elem_output_loc_left_left_bc_periodic")   # face 0 through the seam
    assert nz_text(gst_row(P, 8, 0)) == {r["i"] % 8: r["v_text"] for r in rows}
    P.close()


def test_divergence_rows(golden_dir):
    rows = G(golden_dir, "ex2_all_first_deriv_input_loc_face_output_loc_elem")          # cell (4,4,4) of 8^3
    want = {}
    for r in rows:
        d = {"LEFT": 0, "DOWN": 1, "BACK": 2}[r["loc"]]
        want[(d, (r["i"], r["j"], r["k"])[d])] = r["v_text"]
    P = poisson((8, 8, 8), [V] * 6)
    cell = (4 * 8 + 4) * 8 + 4
    got = {}
    for d in range(3):
        fdim = [8, 8, 8]
        fdim[d] = 9
        for f in (3, 4, 5, 6):
            idx = [4, 4, 4]
            idx[d] = f
            flat = (idx[2] * fdim[1] + idx[1]) * fdim[0] + idx[0]
            Vs = [torch.zeros(P.nface[a], dtype=torch.float64, device="cuda") for a in range(3)]
            Vs[d][flat] = 1.0
            b = P.rhs(*Vs)                        # b = 0 - D V
            if float(b[cell]) != 0.0:
                got[(d, f)] = fmt_g(-float(b[cell]))
    assert got == want
    P.close()


def dense_S(P, ncell):
    cols = [P.apply(unit(ncell, c)).cpu().numpy() for c in range(ncell)]
    return np.stack(cols, axis=1)


@pytest.mark.parametrize("name,bc,cell", [
    ("ex2_all_second_deriv", [V] * 6, (4, 4, 4)),
    ("ex2_all_second_deriv_up_bc_neumann", [V] * 6, (4, 7, 4)),
    ("ex2_all_second_deriv_back_bc_periodic", [V, V, V, V, PER, PER], (4, 4, 0)),
])
def test_schur_rows_3d(golden_dir, name, bc, cell):
    """S = -kappa D Gst (abfpc.c:150-171): with kappa = 1, -S prints like FlucaFD's Laplacian row"""
    rows = [r for r in G(golden_dir, name) if not r["c"].endswith("_boundary")]
    P = poisson((8, 8, 8), bc)
    S = dense_S(P, 512)
    r = (cell[2] * 8 + cell[1]) * 8 + cell[0]
    got = {}
    for c in np.flatnonzero(S[r]):
        got[(int(c) % 8, (int(c) // 8) % 8, int(c) // 64)] = fmt_g(-S[r, c])
    assert got == {(q["i"] % 8, q["j"] % 8, q["k"] % 8): q["v_text"] for q in rows}
    assert np.abs(S.sum(axis=1)).max() < 1e-9          # S 1 = 0: the null space the reference attaches (abfpc.c:173-177)
    P.close()


@pytest.mark.parametrize("name,n,i", [("ex1_second_deriv", 8, 4), ("ex1_second_deriv_refined", 16, 8), ("ex4_second_deriv_compact", 8, 4)])
def test_schur_rows_1d(golden_dir, name, n, i):
    P = poisson((n, 1, 1), [V] * 6)
    S = dense_S(P, n)
    assert nz_text(-S[i]) == {r["i"]: r["v_text"] for r in G(golden_dir, name)}
    P.close()


def test_cell_gradient_row(golden_dir):
    rows = G(golden_dir, "ex1_first_deriv")               # central row at cell 4
    P = poisson((8, 1, 1), [V] * 6)
    row = np.zeros(8)
    for c in range(8):
        vx = torch.zeros(8, dtype=torch.float64, device="cuda")
        P.project(unit(8, c), v=(vx, None, None))          # v = 0 - kappa G e_c
        row[c] = -float(vx[4])
    assert nz_text(row) == {r["i"]: r["v_text"] for r in rows}
    P.close()


def test_velocity_laplacian_dirichlet_row(golden_dir):
    from fluca_amd.poisson import Momentum
    rows = G(golden_dir, "ex1_second_deriv_left_bc_dirichlet")     # i = 0 next to a VELOCITY wall
    elem = {r["i"]: r["v_text"] for r in rows if r["loc"] == "ELEMENT"}
    n = (8, 3, 3)
    P = poisson(n, [V, V, PER, PER, PER, PER])
    M = Momentum(P)
    M.set_coefficients(0.0, 0.0, 1.0)                             # A = L
    ncell = 72
    cell = (1 * 3 + 1) * 8 + 0                                     # (0, 1, 1)
    for comp in range(3):                                          # VELOCITY: every component takes the Dirichlet row
        got = {}
        for i in range(8):
            e = torch.zeros(3 * ncell, dtype=torch.float64, device="cuda")
            e[comp * ncell + (1 * 3 + 1) * 8 + i] = 1.0
            y = M.apply(e)
            # the y and z parts of L act on columns (0, j +- 1, 1), (0, 1, k +- 1) and on the centre: take the x part alone
            if i != 0:
                if float(y[comp * ncell + cell]) != 0.0:
                    got[i] = float(y[comp * ncell + cell])
            else:
                yz = 2 * (-2.0 * 9.0)                              # periodic central rows in y and z: -2 / h^2, h = 1/3
                got[0] = float(y[comp * ncell + cell]) - yz
        assert {k: fmt_g(round(v, 9)) for k, v in got.items()} == elem
    M.close()
    P.close()


def test_wide_laplacian_D_T_G(golden_dir):
    """ex4_second_deriv: the composition of two central first derivatives -- D T G, the 13-point part of the Schur complement
    for PCABFAinvType DIAG / ROWSUM (abfpc.c:150-171), here composed from the device kernels that form it."""
    from fluca_amd.poisson import Momentum
    rows = G(golden_dir, "ex4_second_deriv")                       # (i, j) = (4, 4): columns i = 2, 4, 6
    n = (8, 8, 2)
    P = poisson(n, [V, V, V, V, PER, PER])
    M = Momentum(P)
    ncell = 128
    row = np.zeros(8)
    for i in range(8):
        v = torch.zeros(3 * ncell, dtype=torch.float64, device="cuda")
        vx = v[:ncell]
        P.project(unit(ncell, 4 * 8 + i), v=(vx, None, None))       # v_x = -G_x e
        Vf = M.face_interp(v)                                      # V = T v
        Vf[1].zero_()
        Vf[2].zero_()
        b = P.rhs(*Vf)                                             # b = -D V = D T G e
        row[i] = float(b[4 * 8 + 4])
    assert nz_text(row) == {r["i"]: r["v_text"] for r in rows}
    M.close()
    P.close()


def test_rows_the_ns_assembly_does_not_form(golden_dir):
    """FlucaFD's rows WITHOUT boundary information (bc "none": one-sided through cell centres only) and its three-point Neumann
    row are not rows of any operator on the path: the NS assembly takes the Dirichlet / Neumann helpers at every physical
    boundary (cnlinearcart3d.c:471-512 for L; :2449-2600 for Gst, whose outlet rows use the face value, cartdiscret.c:425-476)
    and never calls NSComputeSecondDeriv*NoCond_Cart (cartdiscret.c:139-166, 234-261).  The device rows next to a wall must
    therefore DIFFER from these goldens -- kept as data so that a change of that fact in the reference would be noticed."""
    P = poisson((8, 1, 1), [V] * 6)
    S = dense_S(P, 8)
    for name, i in (("ex1_second_deriv_left_bc_none", 0), ("ex1_second_deriv_right_bc_none", 7), ("ex1_second_deriv_right_bc_neumann", 7)):
        rows = [r for r in G(golden_dir, name) if r["loc"] == "ELEMENT"]
        assert nz_text(-S[i]) != {r["i"]: r["v_text"] for r in rows}
        assert nz_text(-S[i]) == {i: "-64.", (1 if i == 0 else 6): "64."}      # the two-point Neumann row of D Gst next to a wall
    rows = G(golden_dir, "ex1_first_deriv_input_loc_elem_output_loc_left_left_bc_none")
    assert len(rows) == 3 and nz_text(gst_row(P, 8, 0)) == {}                   # wall face: empty, not the three-point extrapolation
    P.close()
    P = poisson((8, 8, 8), [V] * 6)
    rows = G(golden_dir, "ex2_all_second_deriv_left_bc_none")                    # cell (0,4,4)
    y = P.apply(unit(512, (4 * 8 + 4) * 8 + 0)).cpu().numpy()
    assert fmt_g(-y[(4 * 8 + 4) * 8 + 0]) == "-320." != [r["v_text"] for r in rows if (r["i"], r["j"], r["k"]) == (0, 4, 4)][0]
    P.close()
