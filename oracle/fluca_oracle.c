/*
 * fluca_oracle.c -- CPU restatement of Fluca's pressure-Poisson path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under fluca_amd/ (the product) may call,
 * link or import this file.  Only tests/, __graft_entry__.smoke() and the
 * `cpu_baseline` leg of bench.py use it, and only as the checker / the timed
 * CPU baseline -- never as the thing shipped.
 *
 * What it restates (file:line relative to /root/reference/):
 *   - staggered divergence  D    fluca/src/ns/impl/linearcn/cnlinearcart3d.c:2314-2408
 *   - staggered gradient    Gst  cnlinearcart3d.c:2410-2600 (rows), scaled by
 *                                dt/rho at cnlinearcart3d.c:2907
 *   - Gst boundary vector        cnlinearcart3d.c:2602-2805
 *   - cell pressure gradient G   cnlinearcart3d.c:4-217 (x block shown; y,z alike)
 *   - 1-D stencil rows           fluca/src/ns/utils/cartdiscret.c:3-137,425-476
 *   - Schur complement S = D((-T)G - (-R)) == -kappa D Gst   (Ainv = ID)
 *                                fluca/src/ns/utils/abfpc/abfpc.c:150-171
 *   - PCApply_ABF stage 1/2      abfpc.c:71-101
 *   - constant null space        abfpc.c:173-177, nsbasic.c:214-244
 *   widened rows, at the end of this file:
 *   - velocity Laplacian L       cnlinearcart3d.c:425-632      (1-D rows cartdiscret.c:167-303)
 *   - convection operator C      cnlinearcart3d.c:873-1294     (1-D rows cartdiscret.c:305-371)
 *   - A = I + dt C - (mu dt/2 rho) L, assembled CSR on 3N unknowns   cnlinearcart3d.c:2930-2941
 *   - face interpolations T, B   cnlinearcart3d.c:1934-2140, 1513-1747   (rows cartdiscret.c:373-423)
 *   (fluca_oracle.py composes these into PCApply_ABF, the block Jacobian product and one whole CNLinear time step --
 *    StepOracle, cnlinearcart3d.c:2807-3060 -- and restates the build's own multigrid cycle, MgOracle.)
 *
 * PARITY PIN STATUS
 *   operator coefficients : pinned by the reference's own FlucaFD golden files
 *                           (fluca/tests/fd/output/<case>.out, copied as data into
 *                           tests/golden/flucafd/), see tests/test_oracle_golden.py and, for
 *                           the second-derivative rows of L, tests/test_oracle_momentum.py.
 *                           The convection rows and the T / B rows have no golden in the
 *                           reference; they are checked through the properties they imply
 *                           (skew form on a periodic uniform grid, exactness on linears).
 *   Krylov solve          : PARITY UNPINNED.  The solve runs inside PETSc
 *                           (>= 3.23, fluca/CMakeLists.txt:9-11) which is not
 *                           vendored, not installed here and cannot be built
 *                           (no network).  KSPCG / KSPBCGS / KSPCHEBYSHEV /
 *                           PCJACOBI / MatNullSpaceRemove below are restated
 *                           from PETSc's published algorithms and documented
 *                           defaults ("unverified vs PETSc source").
 *
 * Layouts (all fp64, x fastest):
 *   cell  (i,j,k)            -> (k*N + j)*M + i
 *   x-face(i,j,k) i in [0,Fx) -> (k*N + j)*Fx + i        Fx = M+1, or M if periodic
 *   y-face(i,j,k) j in [0,Fy) -> (k*Fy + j)*M + i        Fy = N+1 / N
 *   z-face(i,j,k) k in [0,Fz) -> (k*N + j)*M + i         Fz = P+1 / P
 *   Face f along an axis sits between cells f-1 and f (DMSTAG_LEFT/DOWN/BACK of
 *   cell f).  On a periodic axis face 0 is also the right face of cell n-1.
 */
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

enum { FO_BC_NONE = 0, FO_BC_VELOCITY = 1, FO_BC_PRESSURE_OUTLET = 2, FO_BC_PERIODIC = 3, FO_BC_SYMMETRY = 4 }; /* flucansbc.h:5-11 */

typedef struct {
  int     n[3];
  int     periodic[3];
  int     bc[6]; /* 0 left 1 right 2 down 3 up 4 back 5 front  (cart.c:564-591) */
  double  kappa; /* dt/rho */
  double *xf[3]; /* n+1 face coordinates (arrc[i][iprev]) */
  double *xcbuf[3];
  double *xc[3]; /* centres, valid for index -1..n (ghosts only meaningful when periodic) */
  int     nf[3]; /* faces per grid line */
  int64_t ncell;
  int64_t nface[3];
} fo_grid;

typedef struct {
  int64_t  nrow, nnz;
  int64_t *rowptr;
  int32_t *col;
  double  *val;
} fo_csr;

/* ------------------------------------------------------------------ grid */

fo_grid *fo_grid_create(const int n[3], const double *xf0, const double *xf1, const double *xf2, const double *xc0, const double *xc1, const double *xc2, const int bc[6], double kappa)
{
  const double *xfin[3] = {xf0, xf1, xf2};
  const double *xcin[3] = {xc0, xc1, xc2};
  fo_grid      *g       = (fo_grid *)calloc(1, sizeof(*g));
  for (int d = 0; d < 3; ++d) {
    int lo = bc[2 * d], hi = bc[2 * d + 1];
    if ((lo == FO_BC_PERIODIC) != (hi == FO_BC_PERIODIC) || n[d] < 1) {
      free(g);
      return NULL;
    }
    g->n[d]        = n[d];
    g->periodic[d] = (lo == FO_BC_PERIODIC);
    g->bc[2 * d]   = lo;
    g->bc[2 * d + 1] = hi;
    g->nf[d]       = n[d] + (g->periodic[d] ? 0 : 1);
    g->xf[d]       = (double *)malloc(sizeof(double) * (n[d] + 1));
    memcpy(g->xf[d], xfin[d], sizeof(double) * (n[d] + 1));
    g->xcbuf[d] = (double *)malloc(sizeof(double) * (n[d] + 2));
    g->xc[d]    = g->xcbuf[d] + 1;
    /* cell centres: midpoints (cart.c:136) unless the caller gives them (uniform product coordinates) */
    for (int i = 0; i < n[d]; ++i) g->xc[d][i] = xcin[d] ? xcin[d][i] : (g->xf[d][i] + g->xf[d][i + 1]) / 2.;
    double L       = g->xf[d][n[d]] - g->xf[d][0];
    g->xc[d][-1]   = g->xc[d][n[d] - 1] - L; /* periodic ghost centres */
    g->xc[d][n[d]] = g->xc[d][0] + L;
  }
  g->kappa    = kappa;
  g->ncell    = (int64_t)n[0] * n[1] * n[2];
  g->nface[0] = (int64_t)g->nf[0] * n[1] * n[2];
  g->nface[1] = (int64_t)n[0] * g->nf[1] * n[2];
  g->nface[2] = (int64_t)n[0] * n[1] * g->nf[2];
  return g;
}

void fo_grid_destroy(fo_grid *g)
{
  if (!g) return;
  for (int d = 0; d < 3; ++d) {
    free(g->xf[d]);
    free(g->xcbuf[d]);
  }
  free(g);
}

int64_t fo_grid_ncell(const fo_grid *g) { return g->ncell; }
int64_t fo_grid_nface(const fo_grid *g, int d) { return g->nface[d]; }

/* ------------------------------------------------- 1-D stencil rows (cartdiscret.c) */

/* NSComputeFaceNormalFirstDerivCentralDiff_Cart, cartdiscret.c:444-457 */
static void facenormal_central(int f, double xW, double xP, int *nc, int col[2], double v[2])
{
  v[0]   = -1. / (xP - xW);
  v[1]   = 1. / (xP - xW);
  col[0] = f - 1;
  col[1] = f;
  *nc    = 2;
}
/* NSComputeFaceNormalFirstDerivForwardDiffDirichletCond_Cart, cartdiscret.c:425-442 */
static void facenormal_fwd_dirichlet(int f, double xw, double xP, double xE, int *nc, int col[2], double v[2])
{
  double h1 = xP - xw, h2 = xE - xw;
  v[0]   = -h2 / (h1 * (h1 - h2));
  v[1]   = h1 / (h2 * (h1 - h2));
  col[0] = f;
  col[1] = f + 1;
  *nc    = 2;
}
/* NSComputeFaceNormalFirstDerivBackwardDiffDirichletCond_Cart, cartdiscret.c:459-476 */
static void facenormal_bwd_dirichlet(int f, double xWW, double xW, double xw, int *nc, int col[2], double v[2])
{
  double h1 = xw - xW, h2 = xw - xWW;
  v[0]   = -h1 / (h2 * (h1 - h2));
  v[1]   = h2 / (h1 * (h1 - h2));
  col[0] = f - 2;
  col[1] = f - 1;
  *nc    = 2;
}

/*
 * One row of the (unscaled) staggered pressure gradient along axis d at face f.
 * Follows cnlinearcart3d.c:2446-2480 (x; y at :2499-2533, z at :2552-2586).
 * Columns are UNWRAPPED cell indices along the axis (-1 on a periodic face 0,
 * exactly what the reference's stencil holds before DMStag wraps it).
 * Returns ncols (0 on a VELOCITY/SYMMETRY wall: zero pressure gradient), -1 on
 * an unsupported BC.
 */
int fo_gst_row_1d(const fo_grid *g, int d, int f, int col[2], double v[2])
{
  const double *xf = g->xf[d], *xc = g->xc[d];
  int           n = g->n[d], nc = 0;
  if (f == 0) {
    switch (g->bc[2 * d]) {
    case FO_BC_VELOCITY:
    case FO_BC_SYMMETRY:
      break;
    case FO_BC_PRESSURE_OUTLET:
      facenormal_fwd_dirichlet(f, xf[f], xc[f], xc[f + 1], &nc, col, v);
      break;
    case FO_BC_PERIODIC:
      facenormal_central(f, xc[f - 1], xc[f], &nc, col, v);
      break;
    default:
      return -1;
    }
  } else if (f == n) {
    switch (g->bc[2 * d + 1]) {
    case FO_BC_VELOCITY:
    case FO_BC_SYMMETRY:
      break;
    case FO_BC_PRESSURE_OUTLET:
      facenormal_bwd_dirichlet(f, xc[f - 2], xc[f - 1], xf[f], &nc, col, v);
      break;
    default: /* PERIODIC cannot happen: face n does not exist */
      return -1;
    }
  } else {
    facenormal_central(f, xc[f - 1], xc[f], &nc, col, v);
  }
  return nc;
}

/* Coefficient of the boundary pressure p_b in the Gst boundary vector,
 * cnlinearcart3d.c:2643-2646 (low side) and :2671-2674 (high side). */
double fo_gst_bc_coeff_1d(const fo_grid *g, int d, int side)
{
  const double *xf = g->xf[d], *xc = g->xc[d];
  int           n = g->n[d];
  if (g->bc[2 * d + side] != FO_BC_PRESSURE_OUTLET) return 0.;
  if (side == 0) {
    double h1 = xc[0] - xf[0], h2 = xc[1] - xf[0];
    return -(h1 + h2) / (h1 * h2);
  } else {
    double h1 = xf[n] - xc[n - 1], h2 = xf[n] - xc[n - 2];
    return (h1 + h2) / (h1 * h2);
  }
}

/* Divergence row of cell i along axis d: -1/dx on the left face, +1/dx on the
 * right face (cnlinearcart3d.c:2348-2362).  Faces are UNWRAPPED (i and i+1). */
void fo_div_row_1d(const fo_grid *g, int d, int i, int face[2], double v[2])
{
  double dx = g->xf[d][i + 1] - g->xf[d][i];
  face[0]   = i;
  face[1]   = i + 1;
  v[0]      = -1. / dx;
  v[1]      = 1. / dx;
}

/* cell-centred first derivative rows (cartdiscret.c:3-137) used by G */
static int cellgrad_row_1d(const fo_grid *g, int d, int i, int col[3], double v[3])
{
  const double *xf = g->xf[d], *xc = g->xc[d];
  int           n = g->n[d];
  double        h1, h2;
  if (i == 0 && g->bc[2 * d] != FO_BC_PERIODIC) {
    switch (g->bc[2 * d]) {
    case FO_BC_VELOCITY: /* NSComputeFirstDerivForwardDiffNoCond_Cart :3-24 */
      h1 = xc[1] - xc[0];
      h2 = xc[2] - xc[0];
      v[0] = -(h1 + h2) / (h1 * h2);
      v[1] = -h2 / (h1 * (h1 - h2));
      v[2] = h1 / (h2 * (h1 - h2));
      col[0] = 0; col[1] = 1; col[2] = 2;
      return 3;
    case FO_BC_PRESSURE_OUTLET: /* ...ForwardDiffDirichletCond :26-43 */
      h1 = xc[0] - xf[0];
      h2 = xc[1] - xc[0];
      v[0] = (h2 - h1) / (h1 * h2);
      v[1] = h1 / (h2 * (h1 + h2));
      col[0] = 0; col[1] = 1;
      return 2;
    case FO_BC_SYMMETRY: /* ...ForwardDiffNeumannCond :45-62 */
      h1 = xc[0] - xf[0];
      h2 = xc[1] - xc[0];
      v[0] = -2. * h1 / (h2 * (2. * h1 + h2));
      v[1] = 2. * h1 / (h2 * (2. * h1 + h2));
      col[0] = 0; col[1] = 1;
      return 2;
    default:
      return -1;
    }
  } else if (i == n - 1 && g->bc[2 * d + 1] != FO_BC_PERIODIC) {
    switch (g->bc[2 * d + 1]) {
    case FO_BC_VELOCITY: /* ...BackwardDiffNoCond :79-100 */
      h1 = xc[i] - xc[i - 1];
      h2 = xc[i] - xc[i - 2];
      v[0] = -h1 / (h2 * (h1 - h2));
      v[1] = h2 / (h1 * (h1 - h2));
      v[2] = (h1 + h2) / (h1 * h2);
      col[0] = i - 2; col[1] = i - 1; col[2] = i;
      return 3;
    case FO_BC_PRESSURE_OUTLET: /* ...BackwardDiffDirichletCond :102-119 */
      h1 = xf[i + 1] - xc[i];
      h2 = xc[i] - xc[i - 1];
      v[0] = -h1 / (h2 * (h1 + h2));
      v[1] = (h1 - h2) / (h1 * h2);
      col[0] = i - 1; col[1] = i;
      return 2;
    case FO_BC_SYMMETRY: /* ...BackwardDiffNeumannCond :120-137 */
      h1 = xf[i + 1] - xc[i];
      h2 = xc[i] - xc[i - 1];
      v[0] = -2. * h1 / (h2 * (2. * h1 + h2));
      v[1] = 2. * h1 / (h2 * (2. * h1 + h2));
      col[0] = i - 1; col[1] = i;
      return 2;
    default:
      return -1;
    }
  }
  /* NSComputeFirstDerivCentralDiff_Cart :64-77 (interior and periodic) */
  v[0]   = -1. / (xc[i + 1] - xc[i - 1]);
  v[1]   = 1. / (xc[i + 1] - xc[i - 1]);
  col[0] = i - 1;
  col[1] = i + 1;
  return 2;
}

/* --------------------------------------------------------- index helpers */

static inline int wrap(int i, int n) { return i < 0 ? i + n : (i >= n ? i - n : i); }

static inline int64_t cell_index(const fo_grid *g, int i, int j, int k) { return ((int64_t)k * g->n[1] + j) * g->n[0] + i; }

static inline int64_t face_index(const fo_grid *g, int d, int i, int j, int k)
{
  /* (i,j,k) with the d-th entry being the face number, already wrapped */
  switch (d) {
  case 0:
    return ((int64_t)k * g->n[1] + j) * g->nf[0] + i;
  case 1:
    return ((int64_t)k * g->nf[1] + j) * g->n[0] + i;
  default:
    return ((int64_t)k * g->n[1] + j) * g->n[0] + i;
  }
}

/* ------------------------------------------- S = D * (-kappa Gst), assembled CSR */

/*
 * Builds S exactly the way the reference gets it (abfpc.c:150-171 with
 * Ainv = ID): row of D (6 entries) times the rows of tmp = -(kappa*Gst).
 * The (-T)G - (-R) detour of the reference cancels to -kappa*Gst up to
 * round-off; its O(eps) ghost entries at i+-2 are not reproduced.
 */
fo_csr *fo_assemble_S(const fo_grid *g)
{
  fo_csr *A  = (fo_csr *)calloc(1, sizeof(*A));
  A->nrow    = g->ncell;
  A->rowptr  = (int64_t *)malloc(sizeof(int64_t) * (A->nrow + 1));
  int64_t cap = A->nrow * 7 + 16;
  A->col     = (int32_t *)malloc(sizeof(int32_t) * cap);
  A->val     = (double *)malloc(sizeof(double) * cap);
  int64_t nnz = 0;
  for (int k = 0; k < g->n[2]; ++k)
    for (int j = 0; j < g->n[1]; ++j)
      for (int i = 0; i < g->n[0]; ++i) {
        int     idx[3] = {i, j, k};
        int64_t ecol[12];
        double  eval[12];
        int     ne  = 0;
        int64_t row = cell_index(g, i, j, k);
        A->rowptr[row] = nnz;
        for (int d = 0; d < 3; ++d) {
          int    face[2];
          double dv[2];
          fo_div_row_1d(g, d, idx[d], face, dv);
          for (int s = 0; s < 2; ++s) {
            int    gc[2], nc;
            double gv[2];
            int    f = face[s];
            if (g->periodic[d] && f == g->n[d]) f = 0; /* right face of the last cell is face 0 */
            nc = fo_gst_row_1d(g, d, f, gc, gv);
            for (int c = 0; c < nc; ++c) {
              int cidx[3] = {i, j, k};
              int cc      = gc[c];
              /* face f==0 reached as the right face of cell n-1: its stencil cells are (-1,0) == (n-1, 0) */
              cidx[d]     = wrap(cc, g->n[d]);
              ecol[ne]    = cell_index(g, cidx[0], cidx[1], cidx[2]);
              eval[ne]    = dv[s] * (-(g->kappa * gv[c]));
              ++ne;
            }
          }
        }
        /* sort by column, merge duplicates */
        for (int a = 1; a < ne; ++a) {
          int64_t c = ecol[a];
          double  v = eval[a];
          int     b = a - 1;
          while (b >= 0 && ecol[b] > c) {
            ecol[b + 1] = ecol[b];
            eval[b + 1] = eval[b];
            --b;
          }
          ecol[b + 1] = c;
          eval[b + 1] = v;
        }
        for (int a = 0; a < ne;) {
          int64_t c = ecol[a];
          double  v = 0.;
          while (a < ne && ecol[a] == c) v += eval[a++];
          if (nnz >= cap) {
            cap    = cap * 2;
            A->col = (int32_t *)realloc(A->col, sizeof(int32_t) * cap);
            A->val = (double *)realloc(A->val, sizeof(double) * cap);
          }
          A->col[nnz] = (int32_t)c;
          A->val[nnz] = v;
          ++nnz;
        }
      }
  A->rowptr[A->nrow] = nnz;
  A->nnz             = nnz;
  return A;
}

void fo_csr_destroy(fo_csr *A)
{
  if (!A) return;
  free(A->rowptr);
  free(A->col);
  free(A->val);
  free(A);
}
int64_t        fo_csr_nnz(const fo_csr *A) { return A->nnz; }
int64_t        fo_csr_nrow(const fo_csr *A) { return A->nrow; }
const int64_t *fo_csr_rowptr(const fo_csr *A) { return A->rowptr; }
const int32_t *fo_csr_col(const fo_csr *A) { return A->col; }
const double  *fo_csr_val(const fo_csr *A) { return A->val; }

void fo_csr_mult(const fo_csr *A, const double *x, double *y)
{
#pragma omp parallel for schedule(static)
  for (int64_t r = 0; r < A->nrow; ++r) {
    double s = 0.;
    for (int64_t p = A->rowptr[r]; p < A->rowptr[r + 1]; ++p) s += A->val[p] * x[A->col[p]];
    y[r] = s;
  }
}

void fo_csr_diag(const fo_csr *A, double *d)
{
#pragma omp parallel for schedule(static)
  for (int64_t r = 0; r < A->nrow; ++r) {
    double s = 0.;
    for (int64_t p = A->rowptr[r]; p < A->rowptr[r + 1]; ++p)
      if (A->col[p] == r) s = A->val[p];
    d[r] = s;
  }
}

/* ------------------------------------------- matrix-free pieces of PCApply_ABF */

/* b = contrhs - D V   (abfpc.c:75-76; contrhs may be NULL == 0, cnlinearcart3d.c:3035) */
void fo_rhs(const fo_grid *g, const double *Vx, const double *Vy, const double *Vz, const double *contrhs, double *b)
{
  const double *V[3] = {Vx, Vy, Vz};
#pragma omp parallel for schedule(static)
  for (int k = 0; k < g->n[2]; ++k)
    for (int j = 0; j < g->n[1]; ++j)
      for (int i = 0; i < g->n[0]; ++i) {
        int    idx[3] = {i, j, k};
        double s      = 0.;
        for (int d = 0; d < 3; ++d) {
          int    face[2], fi[3];
          double dv[2];
          fo_div_row_1d(g, d, idx[d], face, dv);
          for (int t = 0; t < 2; ++t) {
            fi[0] = i; fi[1] = j; fi[2] = k;
            fi[d] = (g->periodic[d] && face[t] == g->n[d]) ? 0 : face[t];
            s += dv[t] * V[d][face_index(g, d, fi[0], fi[1], fi[2])];
          }
        }
        int64_t c = cell_index(g, i, j, k);
        b[c]      = (contrhs ? contrhs[c] : 0.) - s;
      }
}

/* G_d = kappa * Gst p on the faces of axis d (homogeneous part, MatScale at cnlinearcart3d.c:2907) */
void fo_apply_gst(const fo_grid *g, const double *p, double *Gx, double *Gy, double *Gz)
{
  double *G[3] = {Gx, Gy, Gz};
  for (int d = 0; d < 3; ++d) {
    int nn[3] = {g->n[0], g->n[1], g->n[2]};
    nn[d]     = g->nf[d];
#pragma omp parallel for schedule(static)
    for (int k = 0; k < nn[2]; ++k)
      for (int j = 0; j < nn[1]; ++j)
        for (int i = 0; i < nn[0]; ++i) {
          int    fi[3] = {i, j, k}, col[2], nc;
          double v[2], s = 0.;
          nc = fo_gst_row_1d(g, d, fi[d], col, v);
          for (int c = 0; c < nc; ++c) {
            int ci[3] = {i, j, k};
            ci[d]     = wrap(col[c], g->n[d]);
            s += g->kappa * v[c] * p[cell_index(g, ci[0], ci[1], ci[2])];
          }
          G[d][face_index(g, d, i, j, k)] = s;
        }
  }
}

/* w_d = kappa * (G p)_d at cell centres, d = 0,1,2 (cnlinearcart3d.c:4-217, MatScale :2890) */
int fo_apply_G(const fo_grid *g, const double *p, double *wx, double *wy, double *wz)
{
  double *W[3] = {wx, wy, wz};
  int     err  = 0;
  for (int d = 0; d < 3; ++d) {
#pragma omp parallel for schedule(static)
    for (int k = 0; k < g->n[2]; ++k)
      for (int j = 0; j < g->n[1]; ++j)
        for (int i = 0; i < g->n[0]; ++i) {
          int    ci[3] = {i, j, k}, col[3], nc;
          double v[3], s = 0.;
          nc = cellgrad_row_1d(g, d, ci[d], col, v);
          if (nc < 0) {
            err = 1;
            continue;
          }
          for (int c = 0; c < nc; ++c) {
            int cc[3] = {i, j, k};
            cc[d]     = wrap(col[c], g->n[d]);
            s += g->kappa * v[c] * p[cell_index(g, cc[0], cc[1], cc[2])];
          }
          W[d][cell_index(g, i, j, k)] = s;
        }
  }
  return err;
}

/* ------------------------------------------------------------------- KSP */

enum { FO_KSP_CG = 0, FO_KSP_BCGS = 1, FO_KSP_CHEBYSHEV = 2 };
enum { FO_PC_NONE = 0, FO_PC_JACOBI = 1 };
enum { FO_NORM_PRECONDITIONED = 0, FO_NORM_UNPRECONDITIONED = 1, FO_NORM_NATURAL = 2, FO_NORM_NONE = 3 };
/* KSPConvergedReason values (petscksp.h) */
enum {
  FO_CONVERGED_RTOL          = 2,
  FO_CONVERGED_ATOL          = 3,
  FO_CONVERGED_ITS           = 4,
  FO_DIVERGED_ITS            = -3,
  FO_DIVERGED_DTOL           = -4,
  FO_DIVERGED_BREAKDOWN      = -5,
  FO_DIVERGED_INDEFINITE_PC  = -8,
  FO_DIVERGED_NANORINF       = -9,
  FO_DIVERGED_INDEFINITE_MAT = -10
};

typedef struct {
  int    type, pc, norm_type;
  int    remove_nullspace; /* constant null space attached to S (abfpc.c:173-177) */
  int    maxit;
  double rtol, atol, dtol;
  double emin, emax; /* Chebyshev bounds of the preconditioned operator; both 0 -> Gershgorin estimate * (0.1, 1.1) */
  int    cg_single_reduction; /* KSPCG -ksp_cg_single_reduction (reachable in the reference as -ns_abf_schur_ksp_cg_single_reduction, abfpc.c:206) */
} fo_ksp_opts;

typedef struct {
  int    iters, reason;
  double rnorm0, rnorm, seconds;
} fo_ksp_stats;

static double now_seconds(void)
{
#ifdef _OPENMP
  return omp_get_wtime();
#else
  return 0.;
#endif
}

void fo_set_num_threads(int n)
{
#ifdef _OPENMP
  if (n > 0) omp_set_num_threads(n);
#else
  (void)n;
#endif
}

int fo_num_threads(void)
{
#ifdef _OPENMP
  return omp_get_max_threads();
#else
  return 1;
#endif
}

/* deterministic for a given thread count: static schedule, per-thread partial sums combined in thread order
 * (an OpenMP reduction clause combines in an unspecified order, which BiCGStab amplifies into +-several iterations) */
#define FO_MAXT 1024
static double vdot(int64_t n, const double *a, const double *b)
{
  double part[FO_MAXT];
  int    nt = 1;
#pragma omp parallel
  {
#ifdef _OPENMP
    const int t = omp_get_thread_num();
#pragma omp single
    nt = omp_get_num_threads();
#else
    const int t = 0;
#endif
    double s = 0.;
#pragma omp for schedule(static) nowait
    for (int64_t i = 0; i < n; ++i) s += a[i] * b[i];
    if (t < FO_MAXT) part[t] = s;
  }
  double s = 0.;
  for (int t = 0; t < nt && t < FO_MAXT; ++t) s += part[t];
  return s;
}
static double vsum(int64_t n, const double *a)
{
  double part[FO_MAXT];
  int    nt = 1;
#pragma omp parallel
  {
#ifdef _OPENMP
    const int t = omp_get_thread_num();
#pragma omp single
    nt = omp_get_num_threads();
#else
    const int t = 0;
#endif
    double s = 0.;
#pragma omp for schedule(static) nowait
    for (int64_t i = 0; i < n; ++i) s += a[i];
    if (t < FO_MAXT) part[t] = s;
  }
  double s = 0.;
  for (int t = 0; t < nt && t < FO_MAXT; ++t) s += part[t];
  return s;
}
static void vaxpy(int64_t n, double a, const double *x, double *y)
{
#pragma omp parallel for schedule(static)
  for (int64_t i = 0; i < n; ++i) y[i] += a * x[i];
}
static void vaypx(int64_t n, double a, const double *x, double *y) /* y = x + a y */
{
#pragma omp parallel for schedule(static)
  for (int64_t i = 0; i < n; ++i) y[i] = x[i] + a * y[i];
}
static void vcopy(int64_t n, const double *x, double *y)
{
#pragma omp parallel for schedule(static)
  for (int64_t i = 0; i < n; ++i) y[i] = x[i];
}
static void vset(int64_t n, double a, double *y)
{
#pragma omp parallel for schedule(static)
  for (int64_t i = 0; i < n; ++i) y[i] = a;
}

/* MatNullSpaceRemove for the constant null space: y -= mean(y) */
void fo_remove_constant(int64_t n, double *y)
{
  double m = vsum(n, y) / (double)n;
#pragma omp parallel for schedule(static)
  for (int64_t i = 0; i < n; ++i) y[i] -= m;
}

/* KSP_PCApply: z = B r, then null-space removal */
static void pc_apply(const fo_ksp_opts *o, int64_t n, const double *dinv, const double *r, double *z)
{
  if (o->pc == FO_PC_JACOBI) {
#pragma omp parallel for schedule(static)
    for (int64_t i = 0; i < n; ++i) z[i] = r[i] * dinv[i]; /* PCJACOBI: VecPointwiseMult with the stored inverse diagonal */
  } else {
    vcopy(n, r, z);
  }
  if (o->remove_nullspace) fo_remove_constant(n, z);
}

/* KSPConvergedDefault (zero initial guess): ttol = max(rtol*rnorm0, atol) */
static int converged_default(const fo_ksp_opts *o, int it, double rnorm, double *rnorm0, double *ttol)
{
  if (it == 0) {
    *rnorm0 = rnorm;
    *ttol   = fmax(o->rtol * rnorm, o->atol);
  }
  if (isnan(rnorm) || isinf(rnorm)) return FO_DIVERGED_NANORINF;
  if (rnorm <= *ttol) return (rnorm < o->atol) ? FO_CONVERGED_ATOL : FO_CONVERGED_RTOL;
  if (rnorm >= o->dtol * *rnorm0) return FO_DIVERGED_DTOL;
  return 0;
}

/* Gershgorin bound of the Jacobi-preconditioned operator: max_i sum_j |a_ij| / |a_ii| */
double fo_gershgorin_dinvA(const fo_csr *A, int pc)
{
  double m = 0.;
  for (int64_t r = 0; r < A->nrow; ++r) {
    double s = 0., d = 1.;
    for (int64_t p = A->rowptr[r]; p < A->rowptr[r + 1]; ++p) {
      s += fabs(A->val[p]);
      if (A->col[p] == r) d = fabs(A->val[p]);
    }
    if (pc != FO_PC_JACOBI) d = 1.;
    if (s / d > m) m = s / d;
  }
  return m;
}

/*
 * KSPSolve(kspS, b, x) with zero initial guess (abfpc.c:77).
 * hist (may be NULL) receives the monitored norm of iterations 0..iters, at most nhist entries.
 */
int fo_ksp_solve(const fo_csr *A, const double *b, double *x, const fo_ksp_opts *o, fo_ksp_stats *st, double *hist, int nhist)
{
  int64_t n    = A->nrow;
  double *dinv = (double *)malloc(sizeof(double) * n);
  double  t0, rnorm0 = 0., ttol = 0., dp = 0.;
  int     it = 0, reason = 0;

  fo_csr_diag(A, dinv);
#pragma omp parallel for schedule(static)
  for (int64_t i = 0; i < n; ++i) dinv[i] = 1. / dinv[i];
  vset(n, 0., x);
  t0 = now_seconds();

  if (o->type == FO_KSP_CG && o->cg_single_reduction) {
    /* KSPCG with -ksp_cg_single_reduction, restated from PETSc's published algorithm (KSPCGUseSingleReduction manual page and the
     * D'Azevedo / Eijkhout / Romine rearrangement it cites; PETSc source not available here: "unverified vs PETSc source" like the
     * other Krylov restatements).  Besides z = B r it keeps S = A z and W = A p, the latter by the recurrence W = S + b W, so that
     * beta = (z, r), delta = (z, S) and the monitored norm come out of ONE reduction and
     *     (p, A p) = delta - b^2 (p, A p)_old      for p = z + b p_old, b = beta / beta_old
     * needs none of its own.  Same iterates as the two-reduction form in exact arithmetic. */
    double *r = (double *)malloc(sizeof(double) * n), *z = (double *)malloc(sizeof(double) * n), *S = (double *)malloc(sizeof(double) * n);
    double *p = (double *)calloc(n, sizeof(double)), *W = (double *)calloc(n, sizeof(double));
    double  beta, betaold = 1., delta, a, bb, dpi = 0., dpiold;
    vcopy(n, b, r); /* r = b - A*0 */
    pc_apply(o, n, dinv, r, z);
    fo_csr_mult(A, z, S);
    beta  = vdot(n, z, r);
    delta = vdot(n, z, S);
    switch (o->norm_type) {
    case FO_NORM_PRECONDITIONED: dp = sqrt(vdot(n, z, z)); break;
    case FO_NORM_UNPRECONDITIONED: dp = sqrt(vdot(n, r, r)); break;
    case FO_NORM_NATURAL: dp = sqrt(fabs(beta)); break;
    default: dp = 0.;
    }
    if (hist && 0 < nhist) hist[0] = dp;
    reason = converged_default(o, 0, dp, &rnorm0, &ttol);
    while (!reason) {
      if (it >= o->maxit) {
        reason = FO_DIVERGED_ITS;
        break;
      }
      if (beta < 0.) {
        reason = FO_DIVERGED_INDEFINITE_PC;
        break;
      }
      dpiold = dpi;
      if (it == 0) {
        bb  = 0.;
        dpi = delta;
      } else {
        bb  = beta / betaold;
        dpi = delta - bb * bb * dpiold;
      }
      if (dpi <= 0.) {
        reason = FO_DIVERGED_INDEFINITE_MAT;
        break;
      }
      vaypx(n, bb, z, p); /* p = z + b p */
      vaypx(n, bb, S, W); /* W = S + b W  (= A p) */
      betaold = beta;
      a       = beta / dpi;
      vaxpy(n, a, p, x);
      vaxpy(n, -a, W, r);
      pc_apply(o, n, dinv, r, z);
      fo_csr_mult(A, z, S);
      beta  = vdot(n, z, r);
      delta = vdot(n, z, S);
      switch (o->norm_type) {
      case FO_NORM_PRECONDITIONED: dp = sqrt(vdot(n, z, z)); break;
      case FO_NORM_UNPRECONDITIONED: dp = sqrt(vdot(n, r, r)); break;
      case FO_NORM_NATURAL: dp = sqrt(fabs(beta)); break;
      default: dp = 0.;
      }
      ++it;
      if (hist && it < nhist) hist[it] = dp;
      reason = converged_default(o, it, dp, &rnorm0, &ttol);
    }
    free(r); free(z); free(S); free(p); free(W);
  } else if (o->type == FO_KSP_CG) {
    /* KSPCG, Hestenes-Stiefel, left preconditioning */
    double *r = (double *)malloc(sizeof(double) * n), *z = (double *)malloc(sizeof(double) * n);
    double *p = (double *)malloc(sizeof(double) * n), *w = (double *)malloc(sizeof(double) * n);
    double  beta, betaold = 1., a, dpi;
    vcopy(n, b, r); /* r = b - A*0 */
    pc_apply(o, n, dinv, r, z);
    beta = vdot(n, r, z);
    switch (o->norm_type) {
    case FO_NORM_PRECONDITIONED: dp = sqrt(vdot(n, z, z)); break;
    case FO_NORM_UNPRECONDITIONED: dp = sqrt(vdot(n, r, r)); break;
    case FO_NORM_NATURAL: dp = sqrt(fabs(beta)); break;
    default: dp = 0.;
    }
    if (hist && 0 < nhist) hist[0] = dp;
    reason = converged_default(o, 0, dp, &rnorm0, &ttol);
    while (!reason) {
      if (it >= o->maxit) {
        reason = FO_DIVERGED_ITS;
        break;
      }
      if (beta < 0.) {
        reason = FO_DIVERGED_INDEFINITE_PC;
        break;
      }
      if (it == 0) vcopy(n, z, p);
      else vaypx(n, beta / betaold, z, p); /* p = z + (beta/betaold) p */
      betaold = beta;
      fo_csr_mult(A, p, w);
      dpi = vdot(n, p, w);
      if (dpi <= 0.) {
        reason = FO_DIVERGED_INDEFINITE_MAT;
        break;
      }
      a = beta / dpi;
      vaxpy(n, a, p, x);
      vaxpy(n, -a, w, r);
      pc_apply(o, n, dinv, r, z);
      beta = vdot(n, r, z);
      switch (o->norm_type) {
      case FO_NORM_PRECONDITIONED: dp = sqrt(vdot(n, z, z)); break;
      case FO_NORM_UNPRECONDITIONED: dp = sqrt(vdot(n, r, r)); break;
      case FO_NORM_NATURAL: dp = sqrt(fabs(beta)); break;
      default: dp = 0.;
      }
      ++it;
      if (hist && it < nhist) hist[it] = dp;
      reason = converged_default(o, it, dp, &rnorm0, &ttol);
    }
    free(r); free(z); free(p); free(w);
  } else if (o->type == FO_KSP_BCGS) {
    /* KSPBCGS (van der Vorst), left preconditioning: iterates on B A, shadow residual = initial residual */
    double *R = (double *)malloc(sizeof(double) * n), *RP = (double *)malloc(sizeof(double) * n);
    double *P = (double *)calloc(n, sizeof(double)), *V = (double *)calloc(n, sizeof(double));
    double *S = (double *)malloc(sizeof(double) * n), *T = (double *)malloc(sizeof(double) * n), *tmp = (double *)malloc(sizeof(double) * n);
    double  rho, rhoold = 1., alpha = 1., omega, omegaold = 1., beta, d1, d2;
    pc_apply(o, n, dinv, b, R); /* R = B (b - A*0) */
    dp = sqrt(vdot(n, R, R));
    if (hist && 0 < nhist) hist[0] = dp;
    reason = converged_default(o, 0, dp, &rnorm0, &ttol);
    vcopy(n, R, RP);
    while (!reason) {
      if (it >= o->maxit) {
        reason = FO_DIVERGED_ITS;
        break;
      }
      rho = vdot(n, R, RP);
      if (rho == 0.) {
        reason = FO_DIVERGED_BREAKDOWN;
        break;
      }
      beta = (rho / rhoold) * (alpha / omegaold);
#pragma omp parallel for schedule(static)
      for (int64_t i = 0; i < n; ++i) P[i] = R[i] - omegaold * beta * V[i] + beta * P[i]; /* VecAXPBYPCZ */
      fo_csr_mult(A, P, tmp);
      pc_apply(o, n, dinv, tmp, V);
      d1 = vdot(n, V, RP);
      if (d1 == 0.) {
        reason = FO_DIVERGED_BREAKDOWN;
        break;
      }
      alpha = rho / d1;
#pragma omp parallel for schedule(static)
      for (int64_t i = 0; i < n; ++i) S[i] = R[i] - alpha * V[i]; /* VecWAXPY */
      fo_csr_mult(A, S, tmp);
      pc_apply(o, n, dinv, tmp, T);
      d1 = vdot(n, S, T);
      d2 = vdot(n, T, T);
      if (d2 == 0.) {
        /* S == 0: x + alpha P is the solution */
        vaxpy(n, alpha, P, x);
        ++it;
        dp = 0.;
        if (hist && it < nhist) hist[it] = dp;
        reason = FO_CONVERGED_RTOL;
        break;
      }
      omega = d1 / d2;
#pragma omp parallel for schedule(static)
      for (int64_t i = 0; i < n; ++i) {
        x[i] += alpha * P[i] + omega * S[i]; /* VecAXPBYPCZ */
        R[i] = S[i] - omega * T[i];          /* VecWAXPY */
      }
      dp       = sqrt(vdot(n, R, R));
      rhoold   = rho;
      omegaold = omega;
      ++it;
      if (hist && it < nhist) hist[it] = dp;
      reason = converged_default(o, it, dp, &rnorm0, &ttol);
    }
    free(R); free(RP); free(P); free(V); free(S); free(T); free(tmp);
  } else if (o->type == FO_KSP_CHEBYSHEV) {
    /* KSPCHEBYSHEV three-term recurrence, left preconditioning */
    double  emin = o->emin, emax = o->emax;
    double *pkm1 = (double *)malloc(sizeof(double) * n), *pk = (double *)malloc(sizeof(double) * n), *pkp1 = (double *)malloc(sizeof(double) * n);
    double *r = (double *)malloc(sizeof(double) * n), *z = (double *)malloc(sizeof(double) * n);
    double  scale, alpha, Gamma, mu, omegaprod, ckm1, ck, ckp1, omega;
    if (emin == 0. && emax == 0.) {
      double lam = fo_gershgorin_dinvA(A, o->pc);
      emin       = 0.1 * lam; /* -ksp_chebyshev_esteig 0,0.1,0,1.1 applied to the bound */
      emax       = 1.1 * lam;
    }
    scale     = 2. / (emax + emin);
    alpha     = 1. - scale * emin;
    Gamma     = 1.;
    mu        = 1. / alpha;
    omegaprod = 2. / alpha;
    ckm1      = 1.;
    ck        = mu;
    vset(n, 0., pkm1);      /* p[km1] = x = 0 */
    pc_apply(o, n, dinv, b, z); /* z = B (b - A x) */
    switch (o->norm_type) {
    case FO_NORM_PRECONDITIONED: dp = sqrt(vdot(n, z, z)); break;
    case FO_NORM_UNPRECONDITIONED: dp = sqrt(vdot(n, b, b)); break;
    default: dp = 0.;
    }
    if (hist && 0 < nhist) hist[0] = dp;
    reason = (o->norm_type == FO_NORM_NONE) ? 0 : converged_default(o, 0, dp, &rnorm0, &ttol);
#pragma omp parallel for schedule(static)
    for (int64_t i = 0; i < n; ++i) pk[i] = scale * z[i] + pkm1[i]; /* VecAYPX(p[k], scale, p[km1]) */
    if (!reason && o->maxit <= 0) reason = (o->norm_type == FO_NORM_NONE) ? FO_CONVERGED_ITS : FO_DIVERGED_ITS;
    while (!reason) {
      double *t;
      ++it;
      /* r = b - A p[k] */
      fo_csr_mult(A, pk, r);
#pragma omp parallel for schedule(static)
      for (int64_t i = 0; i < n; ++i) r[i] = b[i] - r[i];
      pc_apply(o, n, dinv, r, z);
      if (o->norm_type != FO_NORM_NONE) {
        dp = (o->norm_type == FO_NORM_UNPRECONDITIONED) ? sqrt(vdot(n, r, r)) : sqrt(vdot(n, z, z));
        if (hist && it < nhist) hist[it] = dp;
        reason = converged_default(o, it, dp, &rnorm0, &ttol);
        if (reason) break; /* solution is p[k] */
      }
      if (it >= o->maxit) {
        reason = (o->norm_type == FO_NORM_NONE) ? FO_CONVERGED_ITS : FO_DIVERGED_ITS;
        break;
      }
      ckp1  = 2. * mu * ck - ckm1;
      omega = omegaprod * ck / ckp1;
#pragma omp parallel for schedule(static)
      for (int64_t i = 0; i < n; ++i) pkp1[i] = (1. - omega) * pkm1[i] + omega * pk[i] + omega * Gamma * scale * z[i];
      ckm1 = ck;
      ck   = ckp1;
      t    = pkm1;
      pkm1 = pk;
      pk   = pkp1;
      pkp1 = t;
    }
    vcopy(n, pk, x);
    free(pkm1); free(pk); free(pkp1); free(r); free(z);
  } else {
    free(dinv);
    return -1;
  }
  st->seconds = now_seconds() - t0;
  st->iters   = it;
  st->reason  = reason;
  st->rnorm0  = rnorm0;
  st->rnorm   = dp;
  free(dinv);
  return 0;
}

/* ----------------------------------------------------------------- IBM (own spec; no reference counterpart) */

enum { FO_DELTA_PESKIN4 = 0, FO_DELTA_ROMA3 = 1 };

static double phi_peskin4(double r)
{
  r = fabs(r);
  if (r <= 1.) return (3. - 2. * r + sqrt(1. + 4. * r - 4. * r * r)) / 8.;
  if (r <= 2.) return (5. - 2. * r - sqrt(-7. + 12. * r - 4. * r * r)) / 8.;
  return 0.;
}
static double phi_roma3(double r)
{
  r = fabs(r);
  if (r <= 0.5) return (1. + sqrt(1. - 3. * r * r)) / 3.;
  if (r <= 1.5) return (5. - 3. * r - sqrt(1. - 3. * (1. - r) * (1. - r))) / 6.;
  return 0.;
}

/* support of marker coordinate X along axis d: first cell index i0 and weights w[0..S) (phi only, the
 * 1/h of delta_h cancels against the h of the quadrature); S = 4 or 3.  Uniform spacing required. */
/* Uniform axis (spacing equal to 1e-10 relative): the spec of round 1.  Stretched axis (round 2): the delta function lives in INDEX
 * space -- the marker position is mapped to the continuous cell-centre index s (piecewise linear through the cell centres; beyond the
 * first / last centre through the mirror image of that centre in the wall, or the periodic image) and phi is evaluated at s - i.  On a
 * uniform axis both definitions coincide. */
static int ibm_axis_uniform(const fo_grid *g, int d)
{
  const double h = (g->xf[d][g->n[d]] - g->xf[d][0]) / g->n[d];
  for (int i = 0; i < g->n[d]; ++i)
    if (fabs((g->xf[d][i + 1] - g->xf[d][i]) - h) > 1e-10 * h) return 0;
  return 1;
}
static double ibm_centre(const fo_grid *g, int d, int i) /* i = -1 .. n */
{
  const int n = g->n[d];
  if (i >= 0 && i < n) return g->xc[d][i];
  if (g->periodic[d]) return g->xc[d][i]; /* the periodic images are stored */
  return i < 0 ? 2. * g->xf[d][0] - g->xc[d][0] : 2. * g->xf[d][n] - g->xc[d][n - 1];
}
static int ibm_weights_1d(const fo_grid *g, int d, int kind, double X, int *i0, double w[4])
{
  int    S  = (kind == FO_DELTA_PESKIN4) ? 4 : 3;
  double s;
  if (ibm_axis_uniform(g, d)) {
    double h = (g->xf[d][g->n[d]] - g->xf[d][0]) / g->n[d];
    s        = (X - g->xf[d][0]) / h - 0.5; /* position in cell-centre index units */
  } else {
    /* the interval [centre(c), centre(c+1)) that holds X, c = -1 .. n-1 (clamped: linear extension beyond the ghost centres) */
    int lo = -1, hi = g->n[d] - 1;
    while (lo < hi) {
      int mid = lo + (hi - lo + 1) / 2;
      if (ibm_centre(g, d, mid) <= X) lo = mid;
      else hi = mid - 1;
    }
    const double a = ibm_centre(g, d, lo), b = ibm_centre(g, d, lo + 1);
    s              = (double)lo + (X - a) / (b - a);
  }
  int    i  = (kind == FO_DELTA_PESKIN4) ? (int)floor(s) - 1 : (int)floor(s + 0.5) - 1;
  for (int a = 0; a < S; ++a) {
    double r = s - (double)(i + a);
    w[a]     = (kind == FO_DELTA_PESKIN4) ? phi_peskin4(r) : phi_roma3(r);
  }
  *i0 = i;
  return S;
}

/* U_l = sum_x u(x) delta_h(x - X_l) h^3 ; u has ncomp components, component-major u[c*ncell + cell].
 * Cells outside a non-periodic domain contribute nothing. */
void fo_ibm_interp(const fo_grid *g, int kind, int64_t L, const double *X, const double *Y, const double *Z, int ncomp, const double *u, double *U)
{
#pragma omp parallel for schedule(static)
  for (int64_t l = 0; l < L; ++l) {
    int    i0[3], S;
    double w[3][4];
    S = ibm_weights_1d(g, 0, kind, X[l], &i0[0], w[0]);
    ibm_weights_1d(g, 1, kind, Y[l], &i0[1], w[1]);
    ibm_weights_1d(g, 2, kind, Z[l], &i0[2], w[2]);
    for (int c = 0; c < ncomp; ++c) {
      double s = 0.;
      for (int c3 = 0; c3 < S; ++c3)
        for (int b = 0; b < S; ++b)
          for (int a = 0; a < S; ++a) {
            int ii = i0[0] + a, jj = i0[1] + b, kk = i0[2] + c3;
            if (g->periodic[0]) ii = ((ii % g->n[0]) + g->n[0]) % g->n[0];
            if (g->periodic[1]) jj = ((jj % g->n[1]) + g->n[1]) % g->n[1];
            if (g->periodic[2]) kk = ((kk % g->n[2]) + g->n[2]) % g->n[2];
            if (ii < 0 || ii >= g->n[0] || jj < 0 || jj >= g->n[1] || kk < 0 || kk >= g->n[2]) continue;
            s += w[0][a] * w[1][b] * w[2][c3] * u[(int64_t)c * g->ncell + cell_index(g, ii, jj, kk)];
          }
      U[(int64_t)c * L + l] = s;
    }
  }
}

/* f(x) += sum_l F_l delta_h(x - X_l) dV_l ; delta_h = prod phi(r_d)/h_d.  Serial over markers (deterministic). */
void fo_ibm_spread(const fo_grid *g, int kind, int64_t L, const double *X, const double *Y, const double *Z, const double *dV, int ncomp, const double *F, double *f)
{
  double hx = (g->xf[0][g->n[0]] - g->xf[0][0]) / g->n[0];
  double hy = (g->xf[1][g->n[1]] - g->xf[1][0]) / g->n[1];
  double hz = (g->xf[2][g->n[2]] - g->xf[2][0]) / g->n[2];
  double ih = 1. / (hx * hy * hz);
  /* stretched grids: the force density of a cell is per ITS volume: 1 / (dx_i dy_j dz_k), formed as a product of reciprocals */
  const int uni = ibm_axis_uniform(g, 0) && ibm_axis_uniform(g, 1) && ibm_axis_uniform(g, 2);
  for (int64_t l = 0; l < L; ++l) {
    int    i0[3], S;
    double w[3][4];
    S = ibm_weights_1d(g, 0, kind, X[l], &i0[0], w[0]);
    ibm_weights_1d(g, 1, kind, Y[l], &i0[1], w[1]);
    ibm_weights_1d(g, 2, kind, Z[l], &i0[2], w[2]);
    for (int c3 = 0; c3 < S; ++c3)
      for (int b = 0; b < S; ++b)
        for (int a = 0; a < S; ++a) {
          int ii = i0[0] + a, jj = i0[1] + b, kk = i0[2] + c3;
          if (g->periodic[0]) ii = ((ii % g->n[0]) + g->n[0]) % g->n[0];
          if (g->periodic[1]) jj = ((jj % g->n[1]) + g->n[1]) % g->n[1];
          if (g->periodic[2]) kk = ((kk % g->n[2]) + g->n[2]) % g->n[2];
          if (ii < 0 || ii >= g->n[0] || jj < 0 || jj >= g->n[1] || kk < 0 || kk >= g->n[2]) continue;
          const double vinv = uni ? ih : (1. / (g->xf[0][ii + 1] - g->xf[0][ii])) * (1. / (g->xf[1][jj + 1] - g->xf[1][jj])) * (1. / (g->xf[2][kk + 1] - g->xf[2][kk]));
          double       wt   = w[0][a] * w[1][b] * w[2][c3] * vinv * dV[l];
          for (int c = 0; c < ncomp; ++c) f[(int64_t)c * g->ncell + cell_index(g, ii, jj, kk)] += wt * F[(int64_t)c * L + l];
        }
  }
}

double fo_ibm_phi(int kind, double r) { return kind == FO_DELTA_PESKIN4 ? phi_peskin4(r) : phi_roma3(r); }

/* ================================================================================================
 * Momentum operator  A = I + dt*C - (mu*dt/(2 rho)) * L     
 *
 *   NSFormJacobian_CNLinear_Cart3d_Internal   cnlinearcart3d.c:2930-2941   (MatScale(A,dt); MatAXPY(L); MatShift(1))
 *   ComputeVelocityLaplacianOperator_Private  cnlinearcart3d.c:425-632     (L, ADD_VALUES of one 1-D row per axis)
 *   ComputeConvectionOperator_Private         cnlinearcart3d.c:873-1294    ((Cv)_i = 1/2 d/dx_j (v_i V0_j + v0interp_i v_j))
 *   1-D rows                                  cartdiscret.c:167-232,262-371
 *
 * Unknown ordering here: component-major, row = c*ncell + cell (the reference's DMStag vector interleaves the three
 * components per element; this is a fixed permutation of it).  V0[d]: face-normal velocity on the d-faces; W[c*3+d]:
 * component c of v0interp on the d-faces (arrv0interp[..][iv0interp[c][d]]).  Face layouts as at the top of this file.
 * Restated literally -- including the sign of the low-side Neumann extrapolation row (cartdiscret.c:335-352), which
 * differs from the high-side one; the product reproduces it, see DESIGN.md.
 * ================================================================================================ */

typedef struct {
  int    nc;
  int    off[4]; /* column offsets along the axis, relative to the row's cell (unwrapped) */
  double v[4];
} fo_row1d;

/* One axis' contribution to row (cell i along axis d, component c) of L; cnlinearcart3d.c:466-520 (x), :522-576 (y),
 * :578-632 (z). */
void fo_lap_row_1d(const fo_grid *g, int d, int i, int c, fo_row1d *r)
{
  const double *xf = g->xf[d], *xc = g->xc[d];
  int           n = g->n[d];
  int           kind = 0; /* 0 central, 1 fwd dirichlet, 2 fwd neumann, 3 bwd dirichlet, 4 bwd neumann */
  double        h1, h2, h3;
  if (i == 0) {
    switch (g->bc[2 * d]) {
    case FO_BC_VELOCITY: kind = 1; break;
    case FO_BC_PRESSURE_OUTLET: kind = 2; break;
    case FO_BC_PERIODIC: kind = 0; break;
    case FO_BC_SYMMETRY: kind = (c == d) ? 1 : 2; break;
    default: r->nc = -1; return;
    }
  } else if (i == n - 1) {
    switch (g->bc[2 * d + 1]) {
    case FO_BC_VELOCITY: kind = 3; break;
    case FO_BC_PRESSURE_OUTLET: kind = 4; break;
    case FO_BC_PERIODIC: kind = 0; break;
    case FO_BC_SYMMETRY: kind = (c == d) ? 3 : 4; break;
    default: r->nc = -1; return;
    }
  }
  switch (kind) {
  case 1: /* NSComputeSecondDerivForwardDiffDirichletCond_Cart, cartdiscret.c:167-189 */
    h1 = xc[i] - xf[i];
    h2 = xc[i + 1] - xc[i];
    h3 = xc[i + 2] - xc[i];
    r->v[0] = 2. * (h1 - h2 - h3) / (h1 * h2 * h3);
    r->v[1] = 2. * (h1 - h3) / (h2 * (h1 + h2) * (h2 - h3));
    r->v[2] = 2. * (h2 - h1) / (h3 * (h1 + h3) * (h2 - h3));
    r->off[0] = 0; r->off[1] = 1; r->off[2] = 2;
    r->nc = 3;
    break;
  case 2: /* ...ForwardDiffNeumannCond, :191-208 */
    h1 = xc[i + 1] - xc[i];
    h2 = xf[i + 1] - xf[i];
    r->v[0] = -1. / (h1 * h2);
    r->v[1] = 1. / (h1 * h2);
    r->off[0] = 0; r->off[1] = 1;
    r->nc = 2;
    break;
  case 3: /* ...BackwardDiffDirichletCond, :262-284 */
    h1 = xf[i + 1] - xc[i];
    h2 = xc[i] - xc[i - 1];
    h3 = xc[i] - xc[i - 2];
    r->v[0] = 2. * (h2 - h1) / (h3 * (h1 + h3) * (h2 - h3));
    r->v[1] = 2. * (h1 - h3) / (h2 * (h1 + h2) * (h2 - h3));
    r->v[2] = 2. * (h1 - h2 - h3) / (h1 * h2 * h3);
    r->off[0] = -2; r->off[1] = -1; r->off[2] = 0;
    r->nc = 3;
    break;
  case 4: /* ...BackwardDiffNeumannCond, :286-303 */
    h1 = xc[i] - xc[i - 1];
    h2 = xf[i + 1] - xf[i];
    r->v[0] = 1. / (h1 * h2);
    r->v[1] = -1. / (h1 * h2);
    r->off[0] = -1; r->off[1] = 0;
    r->nc = 2;
    break;
  default: /* NSComputeSecondDerivCentralDiff_Cart, :210-232 (interior and periodic) */
    h1 = xc[i] - xc[i - 1];
    h2 = xc[i + 1] - xc[i];
    h3 = xf[i + 1] - xf[i];
    r->v[0] = 1. / (h1 * h3);
    r->v[1] = -(1. / (h1 * h3) + 1. / (h2 * h3));
    r->v[2] = 1. / (h2 * h3);
    r->off[0] = -1; r->off[1] = 0; r->off[2] = 1;
    r->nc = 3;
  }
}

/* One face's contribution to a row of C: cell i along axis d, side 0 = low face / 1 = high face, face value vf.
 * normal != 0: the row interpolates the face-normal component (second term, or first term with c == d);
 * normal == 0: a tangential component (first term, c != d).  They differ only on a SYMMETRY boundary.
 * cnlinearcart3d.c:931-1040 (x), :1042-1166 (y), :1168-1292 (z). */
void fo_conv_row_1d(const fo_grid *g, int d, int i, int side, int normal, double vf, fo_row1d *r)
{
  const double *xf = g->xf[d], *xc = g->xc[d];
  int           n = g->n[d];
  double        h = xf[i + 1] - xf[i];
  int           kind; /* 0 interpolate, 1 Neumann extrapolation, 2 nothing */
  double        h1, h2;
  r->nc = 0;
  if (side == 0) {
    kind = 0;
    if (i == 0) {
      switch (g->bc[2 * d]) {
      case FO_BC_VELOCITY: kind = 2; break;
      case FO_BC_PRESSURE_OUTLET: kind = 1; break;
      case FO_BC_PERIODIC: kind = 0; break;
      case FO_BC_SYMMETRY: kind = normal ? 2 : 1; break;
      default: r->nc = -1; return;
      }
    }
    if (kind == 0) { /* NSComputeConvectionLinearInterpolationPrev_Cart, cartdiscret.c:305-318 */
      double xW = xc[i - 1], xw = xf[i], xP = xc[i];
      r->v[0] = -0.5 * vf / h * (xP - xw) / (xP - xW);
      r->v[1] = -0.5 * vf / h * (xw - xW) / (xP - xW);
      r->off[0] = -1; r->off[1] = 0;
      r->nc = 2;
    } else if (kind == 1) { /* ...LinearForwardExtrapolationNeumannCond_Cart, :335-352 */
      h1 = xc[i] - xf[i];
      h2 = xc[i + 1] - xf[i];
      r->v[0] = -0.5 * vf / h * (h2 * h2) / ((h1 + h2) * (h1 - h2));
      r->v[1] = 0.5 * vf / h * (h1 * h1) / ((h1 + h2) * (h1 - h2));
      r->off[0] = 0; r->off[1] = 1;
      r->nc = 2;
    }
  } else {
    kind = 0;
    if (i == n - 1) {
      switch (g->bc[2 * d + 1]) {
      case FO_BC_VELOCITY: kind = 2; break;
      case FO_BC_PRESSURE_OUTLET: kind = 1; break;
      case FO_BC_PERIODIC: kind = 0; break;
      case FO_BC_SYMMETRY: kind = normal ? 2 : 1; break;
      default: r->nc = -1; return;
      }
    }
    if (kind == 0) { /* NSComputeConvectionLinearInterpolationNext_Cart, :320-333 */
      double xP = xc[i], xe = xf[i + 1], xE = xc[i + 1];
      r->v[0] = 0.5 * vf / h * (xE - xe) / (xE - xP);
      r->v[1] = 0.5 * vf / h * (xe - xP) / (xE - xP);
      r->off[0] = 0; r->off[1] = 1;
      r->nc = 2;
    } else if (kind == 1) { /* ...LinearBackwardExtrapolationNeumannCond_Cart, :354-371 */
      h1 = xf[i + 1] - xc[i];
      h2 = xf[i + 1] - xc[i - 1];
      r->v[0] = 0.5 * vf / h * (h1 * h1) / ((h1 + h2) * (h1 - h2));
      r->v[1] = -0.5 * vf / h * (h2 * h2) / ((h1 + h2) * (h1 - h2));
      r->off[0] = -1; r->off[1] = 0;
      r->nc = 2;
    }
  }
}

typedef struct {
  int64_t col;
  double  c, l; /* accumulated C and L entries (ADD_VALUES order of the reference) */
} mom_ent;

static void mom_add(mom_ent *e, int *ne, int64_t col, double c, double l)
{
  for (int a = 0; a < *ne; ++a)
    if (e[a].col == col) {
      e[a].c += c;
      e[a].l += l;
      return;
    }
  e[*ne].col = col;
  e[*ne].c   = c;
  e[*ne].l   = l;
  ++*ne;
}

/* A = cI*I + cC*C + cL*L.  The reference's A: cI = 1, cC = dt, cL = -0.5*mu*dt/rho (cnlinearcart3d.c:2937-2940).
 * V0[3], W[9] may be NULL when cC == 0. */
fo_csr *fo_assemble_momentum(const fo_grid *g, double cI, double cC, double cL, const double *const *V0, const double *const *W)
{
  int64_t N = g->ncell;
  fo_csr *A = (fo_csr *)calloc(1, sizeof(*A));
  A->nrow   = 3 * N;
  A->rowptr = (int64_t *)malloc(sizeof(int64_t) * (A->nrow + 1));
  int64_t cap = A->nrow * 16 + 16;
  A->col    = (int32_t *)malloc(sizeof(int32_t) * cap);
  A->val    = (double *)malloc(sizeof(double) * cap);
  int64_t nnz = 0;
  for (int c = 0; c < 3; ++c)
    for (int k = 0; k < g->n[2]; ++k)
      for (int j = 0; j < g->n[1]; ++j)
        for (int i = 0; i < g->n[0]; ++i) {
          int      idx[3] = {i, j, k};
          mom_ent  e[48];
          int      ne  = 0;
          int64_t  row = (int64_t)c * N + cell_index(g, i, j, k);
          fo_row1d r;
          A->rowptr[row] = nnz;
          mom_add(e, &ne, row, 0., 0.);
          /* L */
          for (int d = 0; d < 3; ++d) {
            fo_lap_row_1d(g, d, idx[d], c, &r);
            for (int a = 0; a < r.nc; ++a) {
              int cidx[3] = {i, j, k};
              cidx[d]     = wrap(idx[d] + r.off[a], g->n[d]);
              mom_add(e, &ne, (int64_t)c * N + cell_index(g, cidx[0], cidx[1], cidx[2]), 0., r.v[a]);
            }
          }
          /* C */
          if (cC != 0.)
            for (int d = 0; d < 3; ++d)
              for (int side = 0; side < 2; ++side) {
                int fidx[3] = {i, j, k};
                fidx[d]     = idx[d] + side;
                if (g->periodic[d] && fidx[d] == g->n[d]) fidx[d] = 0;
                int64_t f = face_index(g, d, fidx[0], fidx[1], fidx[2]);
                for (int term = 0; term < 2; ++term) {
                  int    cc = term == 0 ? c : d; /* column component */
                  double vf = term == 0 ? V0[d][f] : W[c * 3 + d][f];
                  fo_conv_row_1d(g, d, idx[d], side, term == 1 || c == d, vf, &r);
                  for (int a = 0; a < r.nc; ++a) {
                    int cidx[3] = {i, j, k};
                    cidx[d]     = wrap(idx[d] + r.off[a], g->n[d]);
                    mom_add(e, &ne, (int64_t)cc * N + cell_index(g, cidx[0], cidx[1], cidx[2]), r.v[a], 0.);
                  }
                }
              }
          /* sort by column */
          for (int a = 1; a < ne; ++a) {
            mom_ent t = e[a];
            int     b = a - 1;
            while (b >= 0 && e[b].col > t.col) {
              e[b + 1] = e[b];
              --b;
            }
            e[b + 1] = t;
          }
          for (int a = 0; a < ne; ++a) {
            if (nnz >= cap) {
              cap    = cap * 2;
              A->col = (int32_t *)realloc(A->col, sizeof(int32_t) * cap);
              A->val = (double *)realloc(A->val, sizeof(double) * cap);
            }
            double v = cC * e[a].c + cL * e[a].l; /* MatScale(A, dt); MatAXPY(A, cL, L) */
            if (e[a].col == row) v += cI;        /* MatShift(A, 1) */
            A->col[nnz] = (int32_t)e[a].col;
            A->val[nnz] = v;
            ++nnz;
          }
        }
  A->rowptr[A->nrow] = nnz;
  A->nnz             = nnz;
  return A;
}

/* ================================================================================================
 * Face-normal velocity interpolation  T  (cell-centred v_d -> d-faces), the remaining operator of PCApply_ABF:
 *   V* = interprhs - (-T) v*          abfpc.c:73-74
 *   ComputeFaceNormalVelocityInterpolationOperator_Private   cnlinearcart3d.c:1934-2140
 *   rows: NSComputeLinearInterpolation_Cart / ...ForwardExtrapolationNeumannCond / ...BackwardExtrapolationNeumannCond
 *         cartdiscret.c:373-423
 * ================================================================================================ */

/* One row of T along axis d at face f (0..n, or 0..n-1 when periodic): columns are UNWRAPPED cell indices.
 * Returns ncols (0: VELOCITY / SYMMETRY wall, the value comes from the boundary-condition vector or is zero). */
int fo_T_row_1d(const fo_grid *g, int d, int f, int col[2], double v[2])
{
  const double *xf = g->xf[d], *xc = g->xc[d];
  int           n = g->n[d];
  double        h1, h2;
  if (f == 0 && !g->periodic[d]) {
    if (g->bc[2 * d] != FO_BC_PRESSURE_OUTLET) return (g->bc[2 * d] == FO_BC_VELOCITY || g->bc[2 * d] == FO_BC_SYMMETRY) ? 0 : -1;
    /* NSComputeLinearForwardExtrapolationNeumannCond_Cart(xw = face 0, xP = centre 0, xE = centre 1), :1974 */
    h1 = xc[0] - xf[0];
    h2 = xc[1] - xf[0];
    v[0] = -(h2 * h2) / ((h1 + h2) * (h1 - h2));
    v[1] = (h1 * h1) / ((h1 + h2) * (h1 - h2));
    col[0] = 0; col[1] = 1;
    return 2;
  }
  if (f == n && !g->periodic[d]) {
    if (g->bc[2 * d + 1] != FO_BC_PRESSURE_OUTLET) return (g->bc[2 * d + 1] == FO_BC_VELOCITY || g->bc[2 * d + 1] == FO_BC_SYMMETRY) ? 0 : -1;
    /* As written at :1993: NSComputeLinearBackwardExtrapolationNeumannCond_Cart(xWW, xW, xw) is handed
     * (centre n-1, face n, centre n) -- centre n is the ghost coordinate DMStagSetUniformCoordinatesProduct leaves
     * there (face n + h/2 on a uniform grid; here g->xc[n]) -- with columns n-2, n-1. */
    h1 = xc[n] - xf[n];
    h2 = xc[n] - xc[n - 1];
    v[0] = (h1 * h1) / ((h1 + h2) * (h1 - h2));
    v[1] = -(h2 * h2) / ((h1 + h2) * (h1 - h2));
    col[0] = n - 2; col[1] = n - 1;
    return 2;
  }
  /* NSComputeLinearInterpolation_Cart(xW = centre f-1, xw = face f, xP = centre f), :2005 (interior and periodic) */
  v[0]   = (xc[f] - xf[f]) / (xc[f] - xc[f - 1]);
  v[1]   = (xf[f] - xc[f - 1]) / (xc[f] - xc[f - 1]);
  col[0] = f - 1;
  col[1] = f;
  return 2;
}

/* V_d = rhs_d + T v_d on every face  (v component-major: v[d*ncell + cell]; rhs may be NULL == 0) */
int fo_apply_T(const fo_grid *g, const double *v, const double *rx, const double *ry, const double *rz, double *Vx, double *Vy, double *Vz)
{
  const double *R[3] = {rx, ry, rz};
  double       *V[3] = {Vx, Vy, Vz};
  int           err = 0;
  for (int d = 0; d < 3; ++d) {
    int nfz = d == 2 ? g->nf[2] : g->n[2], nfy = d == 1 ? g->nf[1] : g->n[1], nfx = d == 0 ? g->nf[0] : g->n[0];
#pragma omp parallel for schedule(static)
    for (int k = 0; k < nfz; ++k)
      for (int j = 0; j < nfy; ++j)
        for (int i = 0; i < nfx; ++i) {
          int     fi[3] = {i, j, k}, col[2], nc;
          double  w[2];
          int64_t q = face_index(g, d, i, j, k);
          double  s = R[d] ? R[d][q] : 0.;
          nc = fo_T_row_1d(g, d, fi[d], col, w);
          if (nc < 0) {
            err = 1;
            continue;
          }
          for (int c = 0; c < nc; ++c) {
            int cc[3] = {i, j, k};
            cc[d]     = wrap(col[c], g->n[d]);
            s += w[c] * v[(int64_t)d * g->ncell + cell_index(g, cc[0], cc[1], cc[2])];
          }
          V[d][q] = s;
        }
  }
  return err;
}

/* ================================================================================================
 * Face velocity interpolation  B  (every velocity component to every face): v0interp = B v0 + vbc,
 * NSStep_CNLinear_Cart3d_Internal, cnlinearcart3d.c:2826-2831;
 * ComputeFaceVelocityInterpolationOperator_Private, cnlinearcart3d.c:1513-1747.
 * Differs from T (above) in two places: a SYMMETRY wall extrapolates the tangential components (c != d) with zero
 * gradient, and the high-side outlet row gets the arguments in the intended order (centres n-2, n-1 and face n).
 * ================================================================================================ */
int fo_B_row_1d(const fo_grid *g, int d, int f, int c, int col[2], double v[2])
{
  const double *xf = g->xf[d], *xc = g->xc[d];
  int           n = g->n[d];
  double        h1, h2;
  if ((f == 0 || f == n) && !g->periodic[d]) {
    int bc = g->bc[2 * d + (f == n)];
    int extrap;
    switch (bc) {
    case FO_BC_VELOCITY: extrap = 0; break;
    case FO_BC_PRESSURE_OUTLET: extrap = 1; break;
    case FO_BC_SYMMETRY: extrap = (c != d); break;
    default: return -1;
    }
    if (!extrap) return 0;
    if (f == 0) { /* NSComputeLinearForwardExtrapolationNeumannCond_Cart(face 0, centre 0, centre 1), :1551 */
      h1 = xc[0] - xf[0];
      h2 = xc[1] - xf[0];
      v[0] = -(h2 * h2) / ((h1 + h2) * (h1 - h2));
      v[1] = (h1 * h1) / ((h1 + h2) * (h1 - h2));
      col[0] = 0; col[1] = 1;
    } else { /* NSComputeLinearBackwardExtrapolationNeumannCond_Cart(centre n-2, centre n-1, face n), :1577 */
      h1 = xf[n] - xc[n - 1];
      h2 = xf[n] - xc[n - 2];
      v[0] = (h1 * h1) / ((h1 + h2) * (h1 - h2));
      v[1] = -(h2 * h2) / ((h1 + h2) * (h1 - h2));
      col[0] = n - 2; col[1] = n - 1;
    }
    return 2;
  }
  v[0]   = (xc[f] - xf[f]) / (xc[f] - xc[f - 1]); /* NSComputeLinearInterpolation_Cart, :1597 */
  v[1]   = (xf[f] - xc[f - 1]) / (xc[f] - xc[f - 1]);
  col[0] = f - 1;
  col[1] = f;
  return 2;
}

/* out[c*3+d] = B v_c on the d-faces (v component-major) */
int fo_apply_B(const fo_grid *g, const double *v, double *const *out)
{
  int err = 0;
  for (int c = 0; c < 3; ++c)
    for (int d = 0; d < 3; ++d) {
      int     nfz = d == 2 ? g->nf[2] : g->n[2], nfy = d == 1 ? g->nf[1] : g->n[1], nfx = d == 0 ? g->nf[0] : g->n[0];
      double *O = out[c * 3 + d];
#pragma omp parallel for schedule(static)
      for (int k = 0; k < nfz; ++k)
        for (int j = 0; j < nfy; ++j)
          for (int i = 0; i < nfx; ++i) {
            int    fi[3] = {i, j, k}, col[2], nc;
            double w[2], s = 0.;
            nc = fo_B_row_1d(g, d, fi[d], c, col, w);
            if (nc < 0) {
              err = 1;
              continue;
            }
            for (int a = 0; a < nc; ++a) {
              int cc[3] = {i, j, k};
              cc[d]     = wrap(col[a], g->n[d]);
              s += w[a] * v[(int64_t)c * g->ncell + cell_index(g, cc[0], cc[1], cc[2])];
            }
            O[face_index(g, d, i, j, k)] = s;
          }
    }
  return err;
}
