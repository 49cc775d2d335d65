// fl_coeff.cpp -- host-side construction of the 1-D operator tables (no GPU code).
//
// The reference assembles D, Gst and S = D((-T)G - (-R)) as PETSc AIJ matrices
// (fluca/src/ns/impl/linearcn/cnlinearcart3d.c:2314-2600, fluca/src/ns/utils/abfpc/abfpc.c:150-171).
// On a Cartesian product grid every one of those rows is a sum of three independent 1-D rows, so the
// whole operator is described by O(M+N+P) numbers per axis.  This file computes them once; the HIP
// kernels never see a matrix.
#include <algorithm>
#include <cmath>
#include <limits>

#include "fl_internal.h"

namespace fl {

// Face-normal first derivative at face f of a grid line: the three 1-D rows of
// fluca/src/ns/utils/cartdiscret.c:425-476, selected as in cnlinearcart3d.c:2446-2480.
static void gst_row(const Axis &a, int64_t f, double &v0, double &v1, int64_t &c0)
{
  const int64_t n = a.n;
  v0 = v1 = 0.;
  c0      = f - 1;
  if (f == 0 && !a.periodic) {
    if (a.bc_lo == FL_BC_PRESSURE_OUTLET) {
      // one-sided through (face value, cell 0, cell 1)
      const double h1 = a.xcc(0) - a.xf[0], h2 = a.xcc(1) - a.xf[0];
      v0 = -h2 / (h1 * (h1 - h2));
      v1 = h1 / (h2 * (h1 - h2));
      c0 = 0;
    } else {
      c0 = 0;  // VELOCITY / SYMMETRY wall: zero pressure gradient, empty row
    }
  } else if (f == n && !a.periodic) {
    if (a.bc_hi == FL_BC_PRESSURE_OUTLET) {
      const double h1 = a.xf[n] - a.xcc(n - 1), h2 = a.xf[n] - a.xcc(n - 2);
      v0 = -h1 / (h2 * (h1 - h2));
      v1 = h2 / (h1 * (h1 - h2));
      c0 = n - 2;
    } else {
      c0 = n - 2 < 0 ? 0 : n - 2;
    }
  } else {
    // interior face, or the periodic face 0 (== face n): central difference between the two adjacent centres
    const int64_t ff = (f == n) ? 0 : f;
    const double  dc = a.xcc(ff) - a.xcc(ff - 1);
    v0 = -1. / dc;
    v1 = 1. / dc;
    c0 = f - 1;  // unwrapped: -1 on the periodic face 0, n-1 on its alias f == n
  }
}

int build_axis(Axis &a, int64_t n, const double *xf, const double *xc, int bc_lo, int bc_hi, double kappa)
{
  if (n < 1 || !xf) return FL_ERR_ARG_WRONG;
  if ((bc_lo == FL_BC_PERIODIC) != (bc_hi == FL_BC_PERIODIC)) return FL_ERR_ARG_WRONG;
  for (int b : {bc_lo, bc_hi})
    if (b != FL_BC_VELOCITY && b != FL_BC_PRESSURE_OUTLET && b != FL_BC_PERIODIC && b != FL_BC_SYMMETRY) return FL_ERR_SUP;  // "Unsupported boundary condition type"
  a.n        = n;
  a.periodic = (bc_lo == FL_BC_PERIODIC);
  a.bc_lo    = bc_lo;
  a.bc_hi    = bc_hi;
  if ((bc_lo == FL_BC_PRESSURE_OUTLET || bc_hi == FL_BC_PRESSURE_OUTLET || bc_lo == FL_BC_VELOCITY || bc_hi == FL_BC_VELOCITY) && !a.periodic && n < 2) {
    /* one-sided rows need two cells; a 1-cell axis is only meaningful with walls, where G is not formed */
    if (bc_lo == FL_BC_PRESSURE_OUTLET || bc_hi == FL_BC_PRESSURE_OUTLET) return FL_ERR_ARG_OUTOFRANGE;
  }
  a.xf.assign(xf, xf + n + 1);
  for (int64_t i = 0; i < n; ++i)
    if (!(a.xf[i + 1] > a.xf[i])) return FL_ERR_ARG_WRONG;
  a.xc.resize(n + 2);
  for (int64_t i = 0; i < n; ++i) a.xc[i + 1] = xc ? xc[i] : (a.xf[i] + a.xf[i + 1]) / 2.;
  const double L = a.xf[n] - a.xf[0];
  a.xc[0]        = a.xc[n] - L;  // periodic images of the last / first centre
  a.xc[n + 1]    = a.xc[1] + L;

  a.ga0.resize(n + 1);
  a.ga1.resize(n + 1);
  a.gc0.resize(n + 1);
  for (int64_t f = 0; f <= n; ++f) gst_row(a, f, a.ga0[f], a.ga1[f], a.gc0[f]);

  a.idx.resize(n);
  a.sl.assign(n, 0.);
  a.sc.assign(n, 0.);
  a.sh.assign(n, 0.);
  for (int64_t i = 0; i < n; ++i) {
    const double dx = a.xf[i + 1] - a.xf[i];
    a.idx[i]        = 1. / dx;
    // (S p)_i = -kappa (1/dx_i) (g_{i+1} - g_i): scatter both face rows into the slots {i-1, i, i+1}
    double slot[3] = {0., 0., 0.};
    for (int s = 0; s < 2; ++s) {
      const int64_t f    = i + s;
      const double  sign = s ? -1. : 1.;
      const int64_t rel  = a.gc0[f] - (i - 1);  // slot of the row's first column
      if (a.ga0[f] != 0. || a.ga1[f] != 0.) {
        if (rel < 0 || rel > 1) return FL_ERR_LIB;
        slot[rel] += sign * kappa * a.idx[i] * a.ga0[f];
        slot[rel + 1] += sign * kappa * a.idx[i] * a.ga1[f];
      }
    }
    a.sl[i] = slot[0];
    a.sc[i] = slot[1];
    a.sh[i] = slot[2];
  }
  if (a.periodic && n == 1) {
    // a single cell across a periodic axis is its own neighbour on both sides: an assembled matrix (MatSetValues with ADD_VALUES,
    // the oracle's CSR) carries the three entries summed on the diagonal, and PCJACOBI divides by that sum -- so do the tables
    a.sc[0] += a.sl[0] + a.sh[0];
    a.sl[0] = a.sh[0] = 0.;
  }

  // outlet pressure coefficient of the Gst boundary vector, cnlinearcart3d.c:2643-2646 / :2671-2674
  a.bcc_lo = a.bcc_hi = 0.;
  if (bc_lo == FL_BC_PRESSURE_OUTLET) {
    const double h1 = a.xcc(0) - a.xf[0], h2 = a.xcc(1) - a.xf[0];
    a.bcc_lo        = -(h1 + h2) / (h1 * h2);
  }
  if (bc_hi == FL_BC_PRESSURE_OUTLET) {
    const double h1 = a.xf[n] - a.xcc(n - 1), h2 = a.xf[n] - a.xcc(n - 2);
    a.bcc_hi        = (h1 + h2) / (h1 * h2);
  }

  // cell-centred gradient G (cnlinearcart3d.c:40-84 for x; same per axis), rows of cartdiscret.c:3-137
  a.Gs.assign(n, 0);
  a.Gv0.assign(n, 0.);
  a.Gv1.assign(n, 0.);
  a.Gv2.assign(n, 0.);
  for (int64_t i = 0; i < n; ++i) {
    double h1, h2;
    if (i == 0 && !a.periodic) {
      if (bc_lo == FL_BC_VELOCITY) {
        if (n < 3) continue;  // no 3-point one-sided row on such a line; G is left zero (the reference would index out of range)
        h1       = a.xcc(1) - a.xcc(0);
        h2       = a.xcc(2) - a.xcc(0);
        a.Gs[i]  = 0;
        a.Gv0[i] = -(h1 + h2) / (h1 * h2);
        a.Gv1[i] = -h2 / (h1 * (h1 - h2));
        a.Gv2[i] = h1 / (h2 * (h1 - h2));
      } else if (bc_lo == FL_BC_PRESSURE_OUTLET) {
        h1       = a.xcc(0) - a.xf[0];
        h2       = a.xcc(1) - a.xcc(0);
        a.Gs[i]  = 0;
        a.Gv0[i] = (h2 - h1) / (h1 * h2);
        a.Gv1[i] = h1 / (h2 * (h1 + h2));
      } else {  // SYMMETRY
        if (n < 2) continue;
        h1       = a.xcc(0) - a.xf[0];
        h2       = a.xcc(1) - a.xcc(0);
        a.Gs[i]  = 0;
        a.Gv0[i] = -2. * h1 / (h2 * (2. * h1 + h2));
        a.Gv1[i] = 2. * h1 / (h2 * (2. * h1 + h2));
      }
    } else if (i == n - 1 && !a.periodic) {
      if (bc_hi == FL_BC_VELOCITY) {
        if (n < 3) continue;
        h1       = a.xcc(i) - a.xcc(i - 1);
        h2       = a.xcc(i) - a.xcc(i - 2);
        a.Gs[i]  = i - 2;
        a.Gv0[i] = -h1 / (h2 * (h1 - h2));
        a.Gv1[i] = h2 / (h1 * (h1 - h2));
        a.Gv2[i] = (h1 + h2) / (h1 * h2);
      } else if (bc_hi == FL_BC_PRESSURE_OUTLET) {
        h1       = a.xf[i + 1] - a.xcc(i);
        h2       = a.xcc(i) - a.xcc(i - 1);
        a.Gs[i]  = i - 1;
        a.Gv0[i] = -h1 / (h2 * (h1 + h2));
        a.Gv1[i] = (h1 - h2) / (h1 * h2);
      } else {
        if (n < 2) continue;
        h1       = a.xf[i + 1] - a.xcc(i);
        h2       = a.xcc(i) - a.xcc(i - 1);
        a.Gs[i]  = i - 1;
        a.Gv0[i] = -2. * h1 / (h2 * (2. * h1 + h2));
        a.Gv1[i] = 2. * h1 / (h2 * (2. * h1 + h2));
      }
    } else {
      const double d = a.xcc(i + 1) - a.xcc(i - 1);
      a.Gs[i]        = i - 1;
      a.Gv0[i]       = -1. / d;
      a.Gv2[i]       = 1. / d;
    }
  }
  return FL_SUCCESS;
}

}  // namespace fl

// ---- host-only decomposition helpers (C-ABI) ------------------------------------------------------------------------

extern "C" int fl_decomp_default(const int64_t n[3], const int ranks[3], int rank, fl_decomp *out)
{
  if (!n || !ranks || !out) return FL_ERR_ARG_NULL;
  const int nr = ranks[0] * ranks[1] * ranks[2];
  if (ranks[0] < 1 || ranks[1] < 1 || ranks[2] < 1 || rank < 0 || rank >= nr) return FL_ERR_ARG_OUTOFRANGE;
  // DMStag numbers ranks x-fastest: rank = (rz*ny + ry)*nx + rx
  int c[3] = {rank % ranks[0], (rank / ranks[0]) % ranks[1], rank / (ranks[0] * ranks[1])};
  for (int d = 0; d < 3; ++d) {
    if (n[d] < ranks[d]) return FL_ERR_ARG_OUTOFRANGE;
    const int64_t q = n[d] / ranks[d], r = n[d] % ranks[d];
    out->ranks[d] = ranks[d];
    out->coord[d] = c[d];
    out->len[d]   = q + (c[d] < r ? 1 : 0);  // the first N%m ranks get one more cell
    out->lo[d]    = q * c[d] + (c[d] < r ? c[d] : r);
  }
  return FL_SUCCESS;
}

extern "C" int fl_decomp_neighbor(const fl_decomp *d, const int periodic[3], int boundary)
{
  if (!d || !periodic || boundary < 0 || boundary > 5) return -1;
  const int ax = boundary / 2, side = boundary % 2;
  int       c[3] = {d->coord[0], d->coord[1], d->coord[2]};
  c[ax] += side ? 1 : -1;
  if (c[ax] < 0 || c[ax] >= d->ranks[ax]) {
    if (!periodic[ax]) return -1;
    c[ax] = (c[ax] + d->ranks[ax]) % d->ranks[ax];
  }
  return (c[2] * d->ranks[1] + c[1]) * d->ranks[0] + c[0];
}

extern "C" int fl_halo_plan(const fl_decomp *d, const int periodic[3], fl_halo_msg out[12])
{
  if (!d || !periodic || !out) return FL_ERR_ARG_NULL;
  int n = 0;
  for (int ax = 0; ax < 3; ++ax) {
    if (d->ranks[ax] == 1) continue;  // a periodic axis held by one rank is wrapped locally, a wall has no neighbour
    const int lo = fl_decomp_neighbor(d, periodic, 2 * ax), hi = fl_decomp_neighbor(d, periodic, 2 * ax + 1);
    if (hi >= 0 && hi == lo) {
      // two ranks on a periodic axis: both faces go to the same peer.  Order: (my high face -> its low ghost) first,
      // then (my low face -> its high ghost); the peer posts the same order, so in-order matching pairs them correctly.
      out[n++] = {hi, 2 * ax + 1, 2 * ax, 2 * ax + 1, 2 * ax + 1};
      out[n++] = {lo, 2 * ax, 2 * ax + 1, 2 * ax, 2 * ax};
    } else {
      if (hi >= 0) out[n++] = {hi, 2 * ax + 1, 2 * ax + 1, 2 * ax + 1, 2 * ax};
      if (lo >= 0) out[n++] = {lo, 2 * ax, 2 * ax, 2 * ax, 2 * ax + 1};
    }
  }
  return n;
}

namespace fl {

// ------------------------------------------------------------------------------------------------ momentum operator
// A = I + dt C - (mu dt / 2 rho) L  (NSFormJacobian, cnlinearcart3d.c:2930-2941).  Every row of L
// (ComputeVelocityLaplacianOperator_Private, cnlinearcart3d.c:425-632) and of C (ComputeConvectionOperator_Private,
// cnlinearcart3d.c:873-1294) is again a sum of 1-D rows, so each axis is described by MOM_NTAB numbers per cell:
//   slot  0.. 3  second-derivative row, "tangential" rule : columns i-1, i, i+1, far   (far = i+2 at the low wall, i-2 at
//   slot  4.. 7  second-derivative row, "normal" rule       the high wall: the one-sided Dirichlet rows)
//   slot  8..10  convection, low  face, tangential : columns i-1, i, i+1, already times -+0.5/h, to be multiplied by the
//   slot 11..13  convection, low  face, normal       face value (V0 or v0interp)
//   slot 14..16  convection, high face, tangential
//   slot 17..19  convection, high face, normal
// "normal" = the velocity component along this axis (c == d) and every second-term row; "tangential" = first-term rows
// of the other two components.  The two differ only on a SYMMETRY boundary (zero normal velocity / zero tangential
// gradient).  tab is slot-major: tab[slot*n + i].
int build_axis_momentum(const Axis &a, std::vector<double> &tab)
{
  const int64_t n = a.n;
  tab.assign((size_t)MOM_NTAB * (size_t)n, 0.);
  auto T = [&](int slot, int64_t i) -> double & { return tab[(size_t)slot * (size_t)n + (size_t)i]; };
  for (int rule = 0; rule < 2; ++rule)
    for (int64_t i = 0; i < n; ++i) {
      // ---- second derivative (cartdiscret.c:167-303)
      int kind = 0;  // 0 central, 1/3 one-sided Dirichlet (low/high), 2/4 Neumann (low/high)
      if (i == 0 && !a.periodic) {
        if (a.bc_lo == FL_BC_VELOCITY) kind = 1;
        else if (a.bc_lo == FL_BC_PRESSURE_OUTLET) kind = 2;
        else if (a.bc_lo == FL_BC_SYMMETRY) kind = rule ? 1 : 2;
        else return FL_ERR_ARG_WRONG;
      } else if (i == n - 1 && !a.periodic) {
        if (a.bc_hi == FL_BC_VELOCITY) kind = 3;
        else if (a.bc_hi == FL_BC_PRESSURE_OUTLET) kind = 4;
        else if (a.bc_hi == FL_BC_SYMMETRY) kind = rule ? 3 : 4;
        else return FL_ERR_ARG_WRONG;
      }
      if ((kind == 1 || kind == 3) && n < 3) return FL_ERR_SUP;  // the reference reads cell i+-2
      if ((kind == 2 || kind == 4) && n < 2) return FL_ERR_SUP;
      const int b = rule * 4;
      if (kind == 1) {
        const double h1 = a.xcc(i) - a.xf[i], h2 = a.xcc(i + 1) - a.xcc(i), h3 = a.xcc(i + 2) - a.xcc(i);
        T(b + 1, i) = 2. * (h1 - h2 - h3) / (h1 * h2 * h3);
        T(b + 2, i) = 2. * (h1 - h3) / (h2 * (h1 + h2) * (h2 - h3));
        T(b + 3, i) = 2. * (h2 - h1) / (h3 * (h1 + h3) * (h2 - h3));
      } else if (kind == 2) {
        const double h1 = a.xcc(i + 1) - a.xcc(i), h2 = a.xf[i + 1] - a.xf[i];
        T(b + 1, i) = -1. / (h1 * h2);
        T(b + 2, i) = 1. / (h1 * h2);
      } else if (kind == 3) {
        const double h1 = a.xf[i + 1] - a.xcc(i), h2 = a.xcc(i) - a.xcc(i - 1), h3 = a.xcc(i) - a.xcc(i - 2);
        T(b + 3, i) = 2. * (h2 - h1) / (h3 * (h1 + h3) * (h2 - h3));
        T(b + 0, i) = 2. * (h1 - h3) / (h2 * (h1 + h2) * (h2 - h3));
        T(b + 1, i) = 2. * (h1 - h2 - h3) / (h1 * h2 * h3);
      } else if (kind == 4) {
        const double h1 = a.xcc(i) - a.xcc(i - 1), h2 = a.xf[i + 1] - a.xf[i];
        T(b + 0, i) = 1. / (h1 * h2);
        T(b + 1, i) = -1. / (h1 * h2);
      } else {
        const double h1 = a.xcc(i) - a.xcc(i - 1), h2 = a.xcc(i + 1) - a.xcc(i), h3 = a.xf[i + 1] - a.xf[i];
        T(b + 0, i) = 1. / (h1 * h3);
        T(b + 1, i) = -(1. / (h1 * h3) + 1. / (h2 * h3));
        T(b + 2, i) = 1. / (h2 * h3);
      }
      // ---- convection (cartdiscret.c:305-371); vf = 1 here
      const double h = a.xf[i + 1] - a.xf[i];
      for (int side = 0; side < 2; ++side) {
        int ck = 0;  // 0 interpolate, 1 Neumann extrapolation, 2 no entries
        const int bc = side == 0 ? a.bc_lo : a.bc_hi;
        if (!a.periodic && ((side == 0 && i == 0) || (side == 1 && i == n - 1))) {
          if (bc == FL_BC_VELOCITY) ck = 2;
          else if (bc == FL_BC_PRESSURE_OUTLET) ck = 1;
          else if (bc == FL_BC_SYMMETRY) ck = rule ? 2 : 1;
          else return FL_ERR_ARG_WRONG;
        }
        if (ck == 1 && n < 2) return FL_ERR_SUP;
        const int s0 = 8 + (side * 2 + rule) * 3;
        if (side == 0) {
          if (ck == 0) {
            const double xW = a.xcc(i - 1), xw = a.xf[i], xP = a.xcc(i);
            T(s0 + 0, i) = -0.5 / h * (xP - xw) / (xP - xW);
            T(s0 + 1, i) = -0.5 / h * (xw - xW) / (xP - xW);
          } else if (ck == 1) {
            // the reference's low-side extrapolation row (cartdiscret.c:335-352) as it stands, sign included
            const double h1 = a.xcc(i) - a.xf[i], h2 = a.xcc(i + 1) - a.xf[i];
            T(s0 + 1, i) = -0.5 / h * (h2 * h2) / ((h1 + h2) * (h1 - h2));
            T(s0 + 2, i) = 0.5 / h * (h1 * h1) / ((h1 + h2) * (h1 - h2));
          }
        } else {
          if (ck == 0) {
            const double xP = a.xcc(i), xe = a.xf[i + 1], xE = a.xcc(i + 1);
            T(s0 + 1, i) = 0.5 / h * (xE - xe) / (xE - xP);
            T(s0 + 2, i) = 0.5 / h * (xe - xP) / (xE - xP);
          } else if (ck == 1) {
            const double h1 = a.xf[i + 1] - a.xcc(i), h2 = a.xf[i + 1] - a.xcc(i - 1);
            T(s0 + 0, i) = 0.5 / h * (h1 * h1) / ((h1 + h2) * (h1 - h2));
            T(s0 + 1, i) = -0.5 / h * (h2 * h2) / ((h1 + h2) * (h1 - h2));
          }
        }
      }
    }
  return 0;
}

// Cell-to-face velocity interpolation rows, V_f = w0 v[c0] + w1 v[c0+1] for every face f = 0..n of a grid line (face n
// of a periodic axis is face 0 and is not stored).  Linear interpolation inside; at a wall either no row (the value
// comes from a boundary-condition vector, or is zero) or a zero-gradient extrapolation through the two nearest cells.
//   kind 0: T, face-normal component, ComputeFaceNormalVelocityInterpolationOperator_Private   cnlinearcart3d.c:1934-2140
//   kind 1: B, component along the face normal (c == d)                                        cnlinearcart3d.c:1513-1747
//   kind 2: B, tangential component (c != d): a SYMMETRY wall extrapolates it
// Rows: cartdiscret.c:373-423.  T's high-side outlet row is restated as written (:1993): the backward extrapolation is
// handed (centre n-1, face n, centre n) for (xWW, xW, xw); centre n is the ghost coordinate.  B has the intended order.
int build_axis_faceinterp(const Axis &a, int kind, std::vector<double> &w0, std::vector<double> &w1, std::vector<int> &c0)
{
  const int64_t n = a.n;
  w0.assign((size_t)n + 1, 0.);
  w1.assign((size_t)n + 1, 0.);
  c0.assign((size_t)n + 1, 0);
  auto extrapolates = [&](int bc, int &ex) {
    if (bc == FL_BC_PRESSURE_OUTLET) ex = 1;
    else if (bc == FL_BC_VELOCITY) ex = 0;
    else if (bc == FL_BC_SYMMETRY) ex = kind == 2;
    else return FL_ERR_ARG_WRONG;
    return 0;
  };
  for (int64_t f = 0; f <= n; ++f) {
    if (f == 0 && !a.periodic) {
      int ex = 0;
      if (int rc = extrapolates(a.bc_lo, ex)) return rc;
      c0[f] = 0;
      if (ex) {
        if (n < 2) return FL_ERR_SUP;
        const double h1 = a.xcc(0) - a.xf[0], h2 = a.xcc(1) - a.xf[0];
        w0[f] = -(h2 * h2) / ((h1 + h2) * (h1 - h2));
        w1[f] = (h1 * h1) / ((h1 + h2) * (h1 - h2));
      }
    } else if (f == n && !a.periodic) {
      int ex = 0;
      if (int rc = extrapolates(a.bc_hi, ex)) return rc;
      c0[f] = (int)std::max<int64_t>(n - 2, 0);
      if (ex) {
        if (n < 2) return FL_ERR_SUP;
        const double h1 = kind == 0 ? a.xcc(n) - a.xf[n] : a.xf[n] - a.xcc(n - 1);
        const double h2 = kind == 0 ? a.xcc(n) - a.xcc(n - 1) : a.xf[n] - a.xcc(n - 2);
        w0[f] = (h1 * h1) / ((h1 + h2) * (h1 - h2));
        w1[f] = -(h2 * h2) / ((h1 + h2) * (h1 - h2));
      }
    } else {
      const int64_t ff = f == n ? 0 : f;  // periodic: face n == face 0 (its row is never read)
      const double  xW = a.xcc(ff - 1), xw = a.xf[ff], xP = a.xcc(ff);
      w0[f] = (xP - xw) / (xP - xW);
      w1[f] = (xw - xW) / (xP - xW);
      c0[f] = (int)(f - 1);
    }
  }
  return 0;
}

}  // namespace fl
