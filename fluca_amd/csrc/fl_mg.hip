// fl_mg.hip -- geometric multigrid preconditioner for the Schur complement (BASELINE.json
// config 3 names its smoother: Chebyshev-Jacobi).  There is no reference function behind it -- the reference reaches
// multigrid only through PETSc options on an assembled matrix -- so the algorithm is specified here and in DESIGN.md
// section 10, and restated on the CPU by the oracle (oracle/fluca_oracle.py: MgOracle) for the parity tests.
//
//   levels     : every axis whose cell count is even and >= 8 is halved (coarse faces = every other fine face, coarse
//                centres = midpoints); the coarse operator is the SAME discretisation on the coarse grid (S = -kappa D Gst
//                from fl_coeff.cpp on the coarse coordinates, same boundary conditions) -- no Galerkin product
//   smoother   : nu steps of Chebyshev over [0.1, 1.1] x (Gershgorin bound of D^-1 S), Jacobi inside, zero initial guess
//                (KSPCHEBYSHEV + PCJACOBI as in fl_ksp.hip; -mg_levels_ksp_type chebyshev -mg_levels_pc_type jacobi)
//   restriction: volume-weighted average of the children's residuals;  prolongation: tri-linear between the parent and its neighbours on
//                the child's side (tuning knob "mg_prolong" = 1, the default; 0 = piecewise constant, child += parent)
//   cycle      : V(nu, nu):  x = smooth(b); r = b - S x; e = V(R r); x += P e; r = b - S x; x += smooth(r)
//   coarsest   : Jacobi-PCG to rtol 1e-2 (at most 200 iterations)
//   outer      : KSPCG, left preconditioning, preconditioned norm ||z||, KSPConvergedDefault, constant null space
//                removed from every preconditioner output; beta in the Polak-Ribiere form (flexible CG, "mg_flexible" below)
//
// Shape: every level is a full fl_poisson handle on the fine handle's stream, and the cycle works on the PADDED work
// vectors of those handles (right-hand side h->r, iterate h->xp, scratch h->q) through the padded entry points of
// fl_ksp.hip -- no pad / unpad copies between the pieces.  Post-smoothing is Chebyshev started from the current iterate
// (the same polynomial in the residual as "e = smooth(b - S x); x += e", without forming the residual).  The constant null
// space is handled once per cycle, algebraically: the outer CG needs z' = z - mean(z) only inside inner products and in
// the direction update, so one pass over (z, r) yields sum z, z.z, r.z, sum r, r.r and the shift is folded into
// p = (z - m) + beta p.  Scalars of the outer CG live on the DEVICE (MgScal, round 5): alpha, beta, the shift m and r.z are formed by one-thread
// kernels from the reduced sums and read from there by the kernels that use them; the host looks at the monitored norm once per iteration, through
// a page-locked slot, and by then the GPU already holds the first third of the next iteration (one deferred wait, none on the critical path).
// Several ranks: every level keeps the fine decomposition (block boundaries coincide with coarse faces), so restriction
// and prolongation stay local; an axis is coarsened only while every rank's share stays even and >= 8 cells; the levels
// borrow the fine handle's communicator for their halo exchanges and reductions.
#include <new>

#include "fl_handle.h"
#include "fl_device.h"

namespace fl {

// All multigrid vectors are padded (GridP addressing, PADX doubles in front of each row => rows are 128-B aligned and a
// pair of cells (i even) is one aligned double2).  One thread per pair of owned cells, grid-stride over (pair, row).
struct Owned {
  int64_t npair_row, npairs;  // pairs per row (last one may be half), pairs in the block
};
__device__ __forceinline__ Owned owned_of(const GridP &g)
{
  Owned o;
  o.npair_row = (g.nx + 1) / 2;
  o.npairs    = o.npair_row * (int64_t)g.ny * g.nz;
  return o;
}
// pair q -> offset of its first cell in a padded array, and whether the second cell exists
__device__ __forceinline__ int64_t pair_off(const GridP &g, const Owned &o, int64_t q, bool &two)
{
  const int     ip = (int)(q % o.npair_row);
  const int64_t row = q / o.npair_row;
  const int     j = (int)(row % g.ny), k = (int)(row / g.ny);
  two = 2 * ip + 1 < g.nx;
  return g.off0 + (int64_t)k * g.sxy + (int64_t)j * g.sx + 2 * ip;
}
__device__ __forceinline__ double2 ldp(const double *a, int64_t off, bool two) { return two ? *reinterpret_cast<const double2 *>(a + off) : make_double2(a[off], 0.); }
__device__ __forceinline__ void    stp(double *a, int64_t off, bool two, double2 v)
{
  if (two) *reinterpret_cast<double2 *>(a + off) = v;
  else a[off] = v.x;
}

// coarse(I,J,K) = sum over children of wx wy wz * fine(child); w = child extent / parent extent along each axis
__global__ void __launch_bounds__(256) k_mg_restrict(GridP gc, GridP gf, int rx, int ry, int rz, const double *__restrict__ wx, const double *__restrict__ wy, const double *__restrict__ wz,
                                                     const double *__restrict__ fine, double *__restrict__ coarse)
{
  const int64_t n = (int64_t)gc.nx * gc.ny * gc.nz;
  for (int64_t q = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; q < n; q += (int64_t)gridDim.x * blockDim.x) {
    const int     I = (int)(q % gc.nx);
    const int64_t t = q / gc.nx;
    const int     J = (int)(t % gc.ny), K = (int)(t / gc.ny);
    double        s = 0.;
    for (int c = 0; c < rz; ++c)
      for (int b = 0; b < ry; ++b) {
        const int     j = J * ry + b, k = K * rz + c;
        const int64_t ro = gf.off0 + (int64_t)k * gf.sxy + (int64_t)j * gf.sx + (int64_t)I * rx;
        const double  wyz = wy[j] * wz[k];
        if (rx == 2) {
          const double2 f = *reinterpret_cast<const double2 *>(fine + ro);
          s += wyz * (wx[2 * I] * f.x + wx[2 * I + 1] * f.y);
        } else s += wyz * wx[I] * fine[ro];
      }
    coarse[gc.off0 + (int64_t)K * gc.sxy + (int64_t)J * gc.sx + I] = s;
  }
}

// fine(child) += coarse(parent)
__global__ void __launch_bounds__(256) k_mg_prolong_add(GridP gf, GridP gc, int rx, int ry, int rz, const double *__restrict__ coarse, double *__restrict__ fine)
{
  const Owned o = owned_of(gf);
  for (int64_t q = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; q < o.npairs; q += (int64_t)gridDim.x * blockDim.x) {
    const int     ip = (int)(q % o.npair_row);
    const int64_t row = q / o.npair_row;
    const int     j = (int)(row % gf.ny), k = (int)(row / gf.ny), i = 2 * ip;
    const bool    two = i + 1 < gf.nx;
    const int64_t fo = gf.off0 + (int64_t)k * gf.sxy + (int64_t)j * gf.sx + i;
    const int64_t co = gc.off0 + (int64_t)(k / rz) * gc.sxy + (int64_t)(j / ry) * gc.sx;
    double2       f = ldp(fine, fo, two);
    f.x += coarse[co + i / rx];
    if (two) f.y += coarse[co + (i + 1) / rx];
    stp(fine, fo, two, f);
  }
}

// fine(child) += tensor-product linear interpolation of the coarse correction between the parent and, along every coarsened axis, the
// neighbour of the parent on the child's side: per axis  (1 - w) coarse(parent) + w coarse(parent + o),  o = -1 / +1, w from the cell
// centres (1/4 on a uniform axis); w = 0 where there is no such neighbour (behind a wall: the correction is continued as a constant,
// which is what a zero normal derivative asks for) and on axes that are not coarsened.  The coarse vector needs its edge and corner
// ghosts (fl_fill_ghosts_full).  pn / pw: per fine cell of this rank's block and axis.
struct MgLin {
  const int    *pn[3];
  const double *pw[3];
};
__global__ void __launch_bounds__(256) k_mg_prolong_lin_add(GridP gf, GridP gc, int rx, int ry, int rz, MgLin t, const double *__restrict__ coarse, double *__restrict__ fine)
{
  // branch-free: where a cell has no neighbour to interpolate with its offset is 0 and its weight 0, so the "neighbour" read is the
  // parent itself.  The two cells of a pair share their four (J, K) lines of the coarse grid.
  const Owned o = owned_of(gf);
  for (int64_t q = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; q < o.npairs; q += (int64_t)gridDim.x * blockDim.x) {
    const int     ip = (int)(q % o.npair_row);
    const int64_t row = q / o.npair_row;
    const int     j = (int)(row % gf.ny), k = (int)(row / gf.ny), i = 2 * ip;
    const bool    two = i + 1 < gf.nx;
    const int     i1 = two ? i + 1 : i;
    const int64_t fo = gf.off0 + (int64_t)k * gf.sxy + (int64_t)j * gf.sx + i;
    const int64_t cjk = gc.off0 + (int64_t)(k / rz) * gc.sxy + (int64_t)(j / ry) * gc.sx;
    const double  wy = t.pw[1][j], wz = t.pw[2][k];
    const int64_t oy = (int64_t)t.pn[1][j] * gc.sx, oz = (int64_t)t.pn[2][k] * gc.sxy;
    const int64_t c0 = cjk + i / rx, c1 = cjk + i1 / rx;
    const int     ox0 = t.pn[0][i], ox1 = t.pn[0][i1];
    const double  wx0 = t.pw[0][i], wx1 = t.pw[0][i1];
    double        v0[4], v1[4];
#pragma unroll
    for (int a = 0; a < 4; ++a) {
      const int64_t sh = ((a & 1) ? oy : 0) + ((a & 2) ? oz : 0);
      const double  p0 = coarse[c0 + sh], p1 = coarse[c1 + sh];
      v0[a] = fma(wx0, coarse[c0 + sh + ox0] - p0, p0);
      v1[a] = fma(wx1, coarse[c1 + sh + ox1] - p1, p1);
    }
    const double a0 = fma(wy, v0[1] - v0[0], v0[0]), b0 = fma(wy, v0[3] - v0[2], v0[2]);
    const double a1 = fma(wy, v1[1] - v1[0], v1[0]), b1 = fma(wy, v1[3] - v1[2], v1[2]);
    double2      f = ldp(fine, fo, two);
    f.x += fma(wz, b0 - a0, a0);
    if (two) f.y += fma(wz, b1 - a1, a1);
    stp(fine, fo, two, f);
  }
}

// The same arithmetic on a tile walk: block = 4 rows x 128 cells (a lane owns a pair) marching through ZC planes, so that row and plane
// numbers are wave-uniform (their table entries arrive by scalar loads, no 64-bit division per pair as in the grid-stride form above) and
// the coarse lines of two consecutive planes / rows are the same lines (L1 / L2 hits).  R2: every axis is coarsened by two (shifts).
// 512^3: 1.1 ms -> see profiles/r03_mg_bench.txt (the fine level moves 17 B/cell: 0.38 ms at 6 TB/s).
template <bool R2>
__global__ void __launch_bounds__(256) k_mg_prolong_lin_tile(GridP gf, GridP gc, int rx, int ry, int rz, MgLin t, const double *__restrict__ coarse, double *__restrict__ fine, int zc)
{
  const int lane = threadIdx.x & 63, w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int i = ((int)blockIdx.x * 64 + lane) * 2, j = (int)blockIdx.y * 4 + w;
  if (j >= gf.ny || i >= gf.nx) return;
  const int     k0 = (int)blockIdx.z * zc, k1 = min(k0 + zc, gf.nz);
  const bool    two = i + 1 < gf.nx;
  const int     i1 = two ? i + 1 : i;
  const int     I0 = R2 ? i >> 1 : i / rx, I1 = R2 ? i1 >> 1 : i1 / rx, J = R2 ? j >> 1 : j / ry;
  const int     ox0 = t.pn[0][i], ox1 = t.pn[0][i1];
  const double  wx0 = t.pw[0][i], wx1 = t.pw[0][i1];
  const double  wy = t.pw[1][j];
  const int64_t oy = (int64_t)t.pn[1][j] * gc.sx;
  const double *cj = coarse + gc.off0 + (int64_t)J * gc.sx;
  double       *fj = fine + gf.off0 + (int64_t)j * gf.sx + i;
  for (int k = k0; k < k1; ++k) {
    const int     K = R2 ? k >> 1 : k / rz;
    const double  wz = t.pw[2][k];
    const int64_t oz = (int64_t)t.pn[2][k] * gc.sxy;
    const double *ck = cj + (int64_t)K * gc.sxy;
    double        v0[4], v1[4];
#pragma unroll
    for (int a = 0; a < 4; ++a) {
      const double *line = ck + ((a & 1) ? oy : 0) + ((a & 2) ? oz : 0);
      const double  p0 = line[I0], p1 = line[I1];
      v0[a] = fma(wx0, line[I0 + ox0] - p0, p0);
      v1[a] = fma(wx1, line[I1 + ox1] - p1, p1);
    }
    const double a0 = fma(wy, v0[1] - v0[0], v0[0]), b0 = fma(wy, v0[3] - v0[2], v0[2]);
    const double a1 = fma(wy, v1[1] - v1[0], v1[0]), b1 = fma(wy, v1[3] - v1[2], v1[2]);
    const int64_t fo = (int64_t)k * gf.sxy;
    double2       f = ldp(fj, fo, two);
    f.x += fma(wz, b0 - a0, a0);
    if (two) f.y += fma(wz, b1 - a1, a1);
    stp(fj, fo, two, f);
  }
}

// The same arithmetic once more, parent-centred, for the usual case that every axis is coarsened by two: a lane owns ONE coarse cell and
// writes its eight children; a wave owns a coarse row (lanes 0 and 63 only carry the x-neighbours of lanes 1 and 62), a block four rows,
// marching through KC coarse planes.  Per coarse cell and plane step: three 8-byte reads of the coarse vector (rows J-1, J, J+1 of the
// plane ahead; x-neighbours by DPP) where the fine-cell-centred kernels issue 64, plus the four 16-byte read-modify-writes of the children.
// The x interpolation is done as a plane arrives (two values per row: one per x-child), so three planes of 3 x 2 values ride in registers.
template <int CTRL>
__device__ __forceinline__ double mg_lane_shift(double v)
{
  const int lo = __builtin_amdgcn_update_dpp(__double2loint(v), __double2loint(v), CTRL, 0xf, 0xf, false);
  const int hi = __builtin_amdgcn_update_dpp(__double2hiint(v), __double2hiint(v), CTRL, 0xf, 0xf, false);
  return __hiloint2double(hi, lo);
}
__global__ void __launch_bounds__(256) k_mg_prolong_lin_cc(GridP gf, GridP gc, MgLin t, const double *__restrict__ coarse, double *__restrict__ fine, int kc)
{
  const int lane = threadIdx.x & 63, w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int J = (int)blockIdx.y * 4 + w;
  if (J >= gc.ny) return;
  const int  I = (int)blockIdx.x * 62 - 1 + lane, Ic = min(I, gc.nx);  // lane 0 of the first block: the low ghost column
  const bool act = lane >= 1 && lane <= 62 && I < gc.nx;
  const int  K0 = (int)blockIdx.z * kc, K1 = min(K0 + kc, gc.nz);
  const int  ia = act ? 2 * I : 0;
  const int    ox[2] = {t.pn[0][ia], t.pn[0][ia + 1]};
  const double wx[2] = {t.pw[0][ia], t.pw[0][ia + 1]};
  const int    oy[2] = {t.pn[1][2 * J], t.pn[1][2 * J + 1]};
  const double wy[2] = {t.pw[1][2 * J], t.pw[1][2 * J + 1]};
  const double *cb = coarse + gc.off0 + (int64_t)J * gc.sx + Ic;
  // x-interpolated values of the three rows of coarse plane K: X[jj + 1][a] for the x-child a
  auto plane = [&](int K, double (&X)[3][2]) {
    const double *cp = cb + (int64_t)K * gc.sxy;
#pragma unroll
    for (int jj = 0; jj < 3; ++jj) {
      const double C = cp[(int64_t)(jj - 1) * gc.sx];
      const double L = mg_lane_shift<0x138>(C), R = mg_lane_shift<0x130>(C);  // lane - 1, lane + 1
#pragma unroll
      for (int a = 0; a < 2; ++a) {
        const double N = ox[a] < 0 ? L : (ox[a] > 0 ? R : C);
        X[jj][a] = fma(wx[a], N - C, C);
      }
    }
  };
  double Xm[3][2], Xc[3][2], Xp[3][2];
  plane(K0 - 1, Xm);
  plane(K0, Xc);
  for (int K = K0; K < K1; ++K) {
    plane(K + 1, Xp);
#pragma unroll
    for (int c = 0; c < 2; ++c) {
      const int    k = 2 * K + c, oz = t.pn[2][k];
      const double wz = t.pw[2][k];
#pragma unroll
      for (int b = 0; b < 2; ++b) {
        double2 add;
#pragma unroll
        for (int a = 0; a < 2; ++a) {
          // the four lines of k_mg_prolong_lin_add: (0,0), (oy,0), (0,oz), (oy,oz)
          const double v0 = Xc[1][a], v1 = oy[b] < 0 ? Xc[0][a] : (oy[b] > 0 ? Xc[2][a] : Xc[1][a]);
          const double z0 = oz < 0 ? Xm[1][a] : (oz > 0 ? Xp[1][a] : Xc[1][a]);
          const double z1 = oz < 0 ? (oy[b] < 0 ? Xm[0][a] : (oy[b] > 0 ? Xm[2][a] : Xm[1][a]))
                                   : (oz > 0 ? (oy[b] < 0 ? Xp[0][a] : (oy[b] > 0 ? Xp[2][a] : Xp[1][a])) : v1);
          const double A = fma(wy[b], v1 - v0, v0), B = fma(wy[b], z1 - z0, z0);
          (a ? add.y : add.x) = fma(wz, B - A, A);
        }
        if (act) {
          double2 *f = reinterpret_cast<double2 *>(fine + gf.off0 + (int64_t)k * gf.sxy + (int64_t)(2 * J + b) * gf.sx + 2 * I);
          double2  v = *f;
          v.x += add.x;
          v.y += add.y;
          *f = v;
        }
      }
    }
#pragma unroll
    for (int jj = 0; jj < 3; ++jj)
#pragma unroll
      for (int a = 0; a < 2; ++a) {
        Xm[jj][a] = Xc[jj][a];
        Xc[jj][a] = Xp[jj][a];
      }
  }
}

// slots: 0 sum z   1 z.z   2 r.z   3 sum r   4 r.r      (owned cells only)
__global__ void __launch_bounds__(256) k_mg_dots(GridP g, const double *__restrict__ z, const double *__restrict__ r, double *__restrict__ partial, int stride)
{
  __shared__ double red[5 * 4];
  double            v[5] = {0., 0., 0., 0., 0.};
  const Owned       o = owned_of(g);
  for (int64_t q = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; q < o.npairs; q += (int64_t)gridDim.x * blockDim.x) {
    bool          two;
    const int64_t off = pair_off(g, o, q, two);
    const double2 zv = ldp(z, off, two), rv = ldp(r, off, two);
    v[0] += zv.x + zv.y;
    v[1] += zv.x * zv.x + zv.y * zv.y;
    v[2] += rv.x * zv.x + rv.y * zv.y;
    v[3] += rv.x + rv.y;
    v[4] += rv.x * rv.x + rv.y * rv.y;
  }
  block_sum<5>(v, red);
  if (threadIdx.x == 0)
#pragma unroll
    for (int a = 0; a < 5; ++a) partial[(int64_t)a * stride + blockIdx.x] = v[a];
}

// OP 0:  y0 = a x0 + b (y0 - c)                (p = (z - m) + beta p  as  y0 = p, x0 = z: y0 = 1*(x0 - c) + b y0, see launch)
// OP 1:  y0 += a x0 ;  y1 -= a x1              (x += alpha p ; r -= alpha q)
// OP 2:  y0 = x0 - c                           (p = z - m)
// OP 3:  y0 += a x0                            (x += alpha p, the update still owed when the iteration stops)
// OP 4:  y1 += a y0 ;  y0 = (x0 - c) + b y0    (x += alpha p with the OLD p, then p = (z - m) + beta p: the x-update rides on the pass that
//                                               rewrites p anyway; r -= alpha q rides on the cycle's first smoothing step, k_cheb_first)
// OP 5:  y1 -= a x1                            (r -= alpha q alone: hierarchies whose level 0 has no smoother)
template <int OP>
__global__ void __launch_bounds__(256) k_mg_pw(GridP g, double a, double b, double c, const double *__restrict__ x0, const double *__restrict__ x1, double *__restrict__ y0, double *__restrict__ y1)
{
  // a wave per 128-cell row segment (row and plane numbers wave-uniform: scalar index arithmetic instead of 64-bit divisions per pair)
  const int     lane = threadIdx.x & 63, w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), nw = blockDim.x >> 6;
  const int     nseg = (g.nx + 127) / 128;
  const int64_t nitem = (int64_t)nseg * g.ny * g.nz;
  for (int64_t it = (int64_t)blockIdx.x * nw + w; it < nitem; it += (int64_t)gridDim.x * nw) {
    const int seg = (int)(it % nseg), row = (int)(it / nseg);
    const int j = row % g.ny, k = row / g.ny, i = seg * 128 + 2 * lane;
    if (i >= g.nx) continue;
    const bool    two = i + 1 < g.nx;
    const int64_t off = g.off0 + (int64_t)k * g.sxy + (int64_t)j * g.sx + i;
    if (OP == 0) {
      const double2 xv = ldp(x0, off, two), yv = ldp(y0, off, two);
      stp(y0, off, two, make_double2((xv.x - c) + b * yv.x, (xv.y - c) + b * yv.y));
    } else if (OP == 1) {
      const double2 pv = ldp(x0, off, two), qv = ldp(x1, off, two), xv = ldp(y0, off, two), rv = ldp(y1, off, two);
      stp(y0, off, two, make_double2(xv.x + a * pv.x, xv.y + a * pv.y));
      stp(y1, off, two, make_double2(rv.x - a * qv.x, rv.y - a * qv.y));
    } else if (OP == 3) {
      const double2 pv = ldp(x0, off, two), xv = ldp(y0, off, two);
      stp(y0, off, two, make_double2(xv.x + a * pv.x, xv.y + a * pv.y));
    } else if (OP == 4) {
      const double2 zv = ldp(x0, off, two), pv = ldp(y0, off, two), xv = ldp(y1, off, two);
      stp(y1, off, two, make_double2(xv.x + a * pv.x, xv.y + a * pv.y));
      stp(y0, off, two, make_double2((zv.x - c) + b * pv.x, (zv.y - c) + b * pv.y));
    } else if (OP == 5) {
      const double2 qv = ldp(x1, off, two), rv = ldp(y1, off, two);
      stp(y1, off, two, make_double2(rv.x - a * qv.x, rv.y - a * qv.y));
    } else {
      const double2 xv = ldp(x0, off, two);
      stp(y0, off, two, make_double2(xv.x - c, xv.y - c));
    }
  }
}

// ---- scalars of the outer (flexible) CG, device-resident --------------------------------------------------------------------------------------
struct MgScal {
  double alpha, beta, m, rz, rz_old, pq, dp, N;
  int    ns, pnorm, bad_pq, pad_;
};
// what the host reads once per iteration (page-locked, one slot per iteration parity)
struct MgSlot {
  double dp, rz, pq;
  int    bad_pq, pad_;
};
// STEP 0: after q = S p and the reduction of p.q (sums[2]):  alpha = r.z / p.q.  A non-positive or NaN p.q is flagged and alpha set to 0, so that the
//         cycle already enqueued behind this kernel changes nothing (r -= 0 q) until the host has seen the flag.
// STEP 1: after a cycle, sums = {sum z, z.z, r.z, sum r, r.r} of k_mg_dots:  m = mean(z) (null space), the monitored norm, r.z' with the shift folded in;
//         the previous r.z moves to rz_old.  FIRST: the cycle in front of the loop (no previous r.z).
// STEP 2: after the dots of (z, q):  beta = -alpha q.z' / (z_old.r_old)  (Polak-Ribiere);  STEP 3: beta = r.z / rz_old (KSPCG's form, no dots needed)
template <int STEP, bool FIRST = false>
__global__ void k_mg_scal(MgScal *__restrict__ S, const double *__restrict__ sums, MgSlot *__restrict__ slot_dev)
{
  if (STEP == 0) {
    const double pq = sums[2];
    S->pq     = pq;
    S->bad_pq = !(pq > 0.);
    S->alpha  = pq > 0. ? S->rz / pq : 0.;
  } else if (STEP == 1) {
    const double d0 = sums[0], d1 = sums[1], d2 = sums[2], d3 = sums[3], d4 = sums[4], N = S->N;
    const double m  = S->ns ? d0 / N : 0.;
    const double zz = d1 - N * m * m;
    double       dp = S->pnorm ? sqrt(zz > 0. ? zz : 0.) : sqrt(d4);
    if (isnan(d1)) dp = d1;
    if (!FIRST) S->rz_old = S->rz;
    S->m  = m;
    S->dp = dp;
    S->rz = d2 - m * d3;
    slot_dev->dp     = dp;
    slot_dev->rz     = S->rz;
    slot_dev->pq     = S->pq;
    slot_dev->bad_pq = FIRST ? 0 : S->bad_pq;
  } else if (STEP == 2) {
    S->beta = -S->alpha * (sums[2] - S->m * sums[3]) / S->rz_old;
  } else {
    S->beta = S->rz / S->rz_old;
  }
}

__global__ void k_mg_scal_set(MgScal *dst, MgScal v) { *dst = v; }

// The coarsest level's Jacobi-PCG (KSPCG + PCJACOBI, preconditioned norm, KSPConvergedDefault with rtol / maxit, constant null space removed from every
// preconditioner output: the algorithm of fl_poisson_solve and of the oracle's fo_ksp_solve) as ONE workgroup: a coarsest grid of a few thousand cells
// is a handful of waves' worth of work per iteration, and through the public solver it cost about sixty launches and a host poll per cycle.  x, r and
// 1 / diag live in registers (CPT cells per thread), the direction p in LDS (its six neighbours are read from there; a periodic axis wraps its index, a
// wall's coefficient is 0), two block reductions per iteration.  b: padded right-hand side; x: padded answer (owned cells written, zero initial guess).
constexpr int MG_COARSE_MAX = 4096, MG_COARSE_CPT = MG_COARSE_MAX / 256;
__global__ void __launch_bounds__(256) k_mg_coarse_cg(GridP g, int perbits, int ns, double rtol, double atol, double dtol, int maxit, const double *__restrict__ b, double *__restrict__ x)
{
  constexpr int CPT = MG_COARSE_CPT;
  __shared__ double P[MG_COARSE_MAX];
  __shared__ double red[4 * 4], bc[4];
  const int tid = threadIdx.x, n = g.nx * g.ny * g.nz;
  double    xr[CPT], rr[CPT], di[CPT];
  int       cell_i[CPT], cell_j[CPT], cell_k[CPT];
  // block-wide sums of up to four numbers, the same bits in every thread
  auto sum4 = [&](double (&v)[4]) {
    block_sum<4>(v, red);
    if (tid == 0)
#pragma unroll
      for (int a = 0; a < 4; ++a) bc[a] = v[a];
    __syncthreads();
#pragma unroll
    for (int a = 0; a < 4; ++a) v[a] = bc[a];
    __syncthreads();
  };
#pragma unroll
  for (int m = 0; m < CPT; ++m) {
    const int c = tid + 256 * m, cc = min(c, n - 1);
    const int i = cc % g.nx, t = cc / g.nx, j = t % g.ny, k = t / g.ny;
    cell_i[m] = i;
    cell_j[m] = j;
    cell_k[m] = k;
    di[m]     = 1. / (g.sc[0][i] + g.sc[1][j] + g.sc[2][k]);
    xr[m]     = 0.;
    rr[m]     = c < n ? b[pidx(g, i, j, k)] : 0.;
  }
  const double N = (double)n;
  double       beta, betaold = 1., mz, dp, rnorm0, ttol;
  // z = M r with the constant removed, folded into the sums:  z' = z - m,  r.z' = r.z - m sum r,  z'.z' = z.z - N m^2
  auto pc_sums = [&]() {
    double v[4] = {0., 0., 0., 0.};
#pragma unroll
    for (int m = 0; m < CPT; ++m)
      if (tid + 256 * m < n) {
        const double z = rr[m] * di[m];
        v[0] += z;
        v[1] += z * z;
        v[2] += rr[m] * z;
        v[3] += rr[m];
      }
    sum4(v);
    mz              = ns ? v[0] / N : 0.;
    const double zz = v[1] - N * mz * mz;
    dp              = sqrt(zz > 0. ? zz : 0.);
    if (isnan(v[1])) dp = v[1];
    beta = v[2] - mz * v[3];
  };
  pc_sums();
  rnorm0     = dp;
  ttol       = fmax(rtol * dp, atol);
  auto done = [&](double v) { return isnan(v) || isinf(v) || v <= ttol || v >= dtol * rnorm0; };
  int  it   = 0;
  bool stop = done(dp);
  while (!stop && it < maxit && !(beta < 0.)) {
    // p = z' + (beta / betaold) p
    const double bb = it == 0 ? 0. : beta / betaold;
#pragma unroll
    for (int m = 0; m < CPT; ++m) {
      const int c = tid + 256 * m;
      if (c < n) P[c] = (rr[m] * di[m] - mz) + (it == 0 ? 0. : bb * P[c]);
    }
    betaold = beta;
    __syncthreads();
    // w = S p ; p.w
    double wv[CPT], v[4] = {0., 0., 0., 0.};
#pragma unroll
    for (int m = 0; m < CPT; ++m) {
      const int c = tid + 256 * m;
      wv[m]       = 0.;
      if (c < n) {
        const int i = cell_i[m], j = cell_j[m], k = cell_k[m];
        const int im = i > 0 ? i - 1 : ((perbits & 1) ? g.nx - 1 : i), ip = i < g.nx - 1 ? i + 1 : ((perbits & 1) ? 0 : i);
        const int jm = j > 0 ? j - 1 : ((perbits & 2) ? g.ny - 1 : j), jp = j < g.ny - 1 ? j + 1 : ((perbits & 2) ? 0 : j);
        const int km = k > 0 ? k - 1 : ((perbits & 4) ? g.nz - 1 : k), kp = k < g.nz - 1 ? k + 1 : ((perbits & 4) ? 0 : k);
        const int row = (k * g.ny + j) * g.nx;
        double    acc = (g.sc[0][i] + g.sc[1][j] + g.sc[2][k]) * P[c];
        acc += g.sl[0][i] * P[row + im] + g.sh[0][i] * P[row + ip];
        acc += g.sl[1][j] * P[(k * g.ny + jm) * g.nx + i] + g.sh[1][j] * P[(k * g.ny + jp) * g.nx + i];
        acc += g.sl[2][k] * P[(km * g.ny + j) * g.nx + i] + g.sh[2][k] * P[(kp * g.ny + j) * g.nx + i];
        wv[m] = acc;
        v[0] += P[c] * acc;
      }
    }
    sum4(v);
    const double dpi = v[0];
    if (!(dpi > 0.)) break;  // KSP_DIVERGED_INDEFINITE_MAT / NaN: the correction found so far is what the cycle gets
    const double a = beta / dpi;
#pragma unroll
    for (int m = 0; m < CPT; ++m) {
      const int c = tid + 256 * m;
      if (c < n) {
        xr[m] += a * P[c];
        rr[m] -= a * wv[m];
      }
    }
    pc_sums();
    ++it;
    stop = done(dp);
  }
#pragma unroll
  for (int m = 0; m < CPT; ++m)
    if (tid + 256 * m < n) x[pidx(g, cell_i[m], cell_j[m], cell_k[m])] = xr[m];
}

// k_mg_pw with its scalars read from MgScal.  OP 2: y0 = x0 - m ;  OP 4: y1 += alpha y0 ; y0 = (x0 - m) + beta y0 ;  OP 5: y1 -= alpha x1
template <int OP>
__global__ void __launch_bounds__(256) k_mg_pwd(GridP g, const MgScal *__restrict__ S, const double *__restrict__ x0, const double *__restrict__ x1, double *__restrict__ y0, double *__restrict__ y1)
{
  const double  a = S->alpha, b = S->beta, c = S->m;
  const int     lane = threadIdx.x & 63, w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), nw = blockDim.x >> 6;
  const int     nseg = (g.nx + 127) / 128;
  const int64_t nitem = (int64_t)nseg * g.ny * g.nz;
  for (int64_t it = (int64_t)blockIdx.x * nw + w; it < nitem; it += (int64_t)gridDim.x * nw) {
    const int seg = (int)(it % nseg), row = (int)(it / nseg);
    const int j = row % g.ny, k = row / g.ny, i = seg * 128 + 2 * lane;
    if (i >= g.nx) continue;
    const bool    two = i + 1 < g.nx;
    const int64_t off = g.off0 + (int64_t)k * g.sxy + (int64_t)j * g.sx + i;
    if (OP == 2) {
      const double2 xv = ldp(x0, off, two);
      stp(y0, off, two, make_double2(xv.x - c, xv.y - c));
    } else if (OP == 4) {
      const double2 zv = ldp(x0, off, two), pv = ldp(y0, off, two), xv = ldp(y1, off, two);
      stp(y1, off, two, make_double2(xv.x + a * pv.x, xv.y + a * pv.y));
      stp(y0, off, two, make_double2((zv.x - c) + b * pv.x, (zv.y - c) + b * pv.y));
    } else {
      const double2 qv = ldp(x1, off, two), rv = ldp(y1, off, two);
      stp(y1, off, two, make_double2(rv.x - a * qv.x, rv.y - a * qv.y));
    }
  }
}

}  // namespace fl

using namespace fl;

struct MgLevel {
  fl_poisson *h = nullptr;  // level 0: the caller's handle (not owned)
  int         r[3] = {1, 1, 1};  // refinement ratio to the NEXT (coarser) level
  double     *w[3] = {nullptr, nullptr, nullptr};  // restriction weights of this level's cells along each axis
  int        *pn[3] = {nullptr, nullptr, nullptr};  // tri-linear prolongation onto this level: neighbour offset (-1 / +1 / 0) ...
  double     *pw[3] = {nullptr, nullptr, nullptr};  // ... and its weight, per cell of this rank's block
  double     *x = nullptr, *b = nullptr;  // unpadded cell arrays: the coarsest level's solve goes through the public entry point
};

struct fl_mg {
  std::vector<MgLevel> lv;
  int                  requested = 0;       // fl_ksp_opts.mg_levels the hierarchy was built for (0 = as deep as possible)
  MgScal              *scal = nullptr;      // device: the outer CG's scalars
  MgSlot              *slot_dev = nullptr;  // device staging of what the host reads ...
  MgSlot              *slot_host = nullptr; // ... and its page-locked copies, one per iteration parity
  hipEvent_t           ev_slot[2] = {nullptr, nullptr};
};

void fl_mg_destroy(fl_poisson *h);

// "mg_prolong" (fl_tuning_set): 1 (default since round 3) tri-linear prolongation, 0 piecewise
// constant.  512^3 cavity, rtol 1e-8, nu = 3: 7 iterations / 0.078 - 0.080 s against 8 / 0.082 - 0.083 s on the same box; nu = 2: 10 against 22
// iterations, nu = 1: 20 against 132 (profiles/r03_mg_bench.txt)
static inline int fl_mg_prolong_mode() { return knob(K_mg_prolong); }

// "mg_flexible" (fl_tuning_set): 1 (default since round 4) the outer CG takes the Polak-Ribiere form
// of beta, beta = z_new . (r_new - r_old) / (z_old . r_old) = -alpha (q . z_new) / (z_old . r_old) -- KSPFCG truncated to one direction
// (-ksp_fcg_mmax 1) --, 0 the Fletcher-Reeves form of KSPCG, r_new . z_new / (r_old . z_old).  The two agree for a fixed symmetric
// preconditioner; the V-cycle is neither once the restriction is not a multiple of the transposed prolongation (tri-linear against
// volume-weighted; any transfer pair on a stretched grid): with two smoothing steps a stretched channel needed 100 iterations with the
// KSPCG form and 17 with this one (oracle study, profiles/r04_mg_flexible.txt).  Costs one more pass over (z, q) per iteration.
static inline int fl_mg_flexible_mode() { return knob(K_mg_flexible); }

// "mg_coarse" (fl_tuning_set): 1 (default) a coarsest level of at most 4096 cells on one rank is solved by one workgroup (k_mg_coarse_cg); 0: through
// the public Jacobi-PCG like every other size (A/B runs; the same algorithm, other summation order)
static inline int fl_mg_coarse_mode() { return knob(K_mg_coarse); }

namespace {

constexpr int MG_BLOCKS = 4096;

int nblk(int64_t n) { return (int)std::max<int64_t>(1, std::min<int64_t>((n + 255) / 256, MG_BLOCKS)); }
int nblk_pairs(const GridP &g) { return nblk((int64_t)((g.nx + 1) / 2) * g.ny * g.nz); }

// the five sums of k_mg_dots over all ranks, on the host (one wait)
int dots(fl_poisson *h, const double *z, const double *r, double out[5])
{
  const int nb = nblk_pairs(h->g);
  hipLaunchKernelGGL(k_mg_dots, dim3(nb), dim3(256), 0, h->stream, h->g, z, r, h->partial, h->partial_stride);
  launch_reduce(h->stream, h->partial, nb, h->partial_stride, 5, h->sums);
  if (h->multi) FL_CHK(h->comm.allreduce(h->stream, h->sums, NSLOT));
  FL_HIP(hipMemcpyAsync(out, h->sums, sizeof(double) * 5, hipMemcpyDeviceToHost, h->stream));
  FL_HIP(hipStreamSynchronize(h->stream));
  return 0;
}

int alloc_cells(fl_poisson *h, double **p) { return fl_dev_alloc(h, (void **)p, sizeof(double) * (size_t)h->ncell, true); }

int mg_build_levels(fl_poisson *h, fl_mg *mg, int max_levels);

// builds h->mg; on any failure everything created so far (coarse handles included) is released again
int mg_build(fl_poisson *h, int max_levels)
{
  h->mg = new (std::nothrow) fl_mg;
  if (!h->mg) return FL_ERR_MEM;
  const int rc = mg_build_levels(h, h->mg, max_levels);
  if (rc != 0) fl_mg_destroy(h);
  return rc;
}

int mg_build_levels(fl_poisson *h, fl_mg *mg, int max_levels)
{
  MgLevel L0;
  L0.h = h;
  mg->lv.push_back(L0);
  for (int l = 0; max_levels <= 0 || l + 1 < max_levels; ++l) {
    fl_poisson *hf = mg->lv[l].h;
    int         r[3];
    bool        any = false;
    for (int d = 0; d < 3; ++d) {
      const int64_t n = hf->ax[d].n, m = hf->dec.ranks[d];
      // every rank's share must stay even and >= 8: with DMStag's default split that means n divisible by 2 m
      r[d] = (n % (2 * m) == 0 && n / m >= 8 && hf->dec.lo[d] % 2 == 0 && hf->dec.len[d] % 2 == 0) ? 2 : 1;
    }
    if (hf->multi) {
      // the ranks must build the SAME hierarchy (their halo exchanges and all-reduces pair up level by level): with a caller-supplied
      // uneven split the local parity tests can differ from rank to rank, so an axis is coarsened only if every rank can
      FL_HIP(hipStreamSynchronize(hf->stream));
      double ok[NSLOT] = {0., 0., 0., 0., 0., 0., 0., 0.};
      for (int d = 0; d < 3; ++d) ok[d] = r[d] == 2 ? 0. : 1.;  // number of ranks that cannot
      FL_HIP(hipMemcpy(hf->sums, ok, sizeof(ok), hipMemcpyHostToDevice));
      FL_CHK(hf->comm.allreduce(hf->stream, hf->sums, NSLOT));
      FL_HIP(hipMemcpyAsync(ok, hf->sums, sizeof(ok), hipMemcpyDeviceToHost, hf->stream));
      FL_HIP(hipStreamSynchronize(hf->stream));
      for (int d = 0; d < 3; ++d)
        if (ok[d] != 0.) r[d] = 1;
    }
    for (int d = 0; d < 3; ++d) any |= r[d] == 2;
    if (!any) break;
    // coarse grid: every other face of the coarsened axes, centres = midpoints
    std::vector<double> xf[3];
    fl_grid             cg;
    for (int d = 0; d < 3; ++d) {
      const int64_t nc = hf->ax[d].n / r[d];
      xf[d].resize((size_t)nc + 1);
      for (int64_t i = 0; i <= nc; ++i) xf[d][(size_t)i] = hf->ax[d].xf[(size_t)(i * r[d])];
      cg.n[d]  = nc;
      cg.xf[d] = xf[d].data();
      cg.xc[d] = nullptr;
    }
    fl_poisson *hc = nullptr;
    fl_decomp   cd = hf->dec;
    for (int d = 0; d < 3; ++d) {
      cd.lo[d] /= r[d];
      cd.len[d] /= r[d];
    }
    if (fl_poisson_create(&cg, h->bc, h->kappa, h->multi ? &cd : nullptr, h->device, &hc) != 0) break;  // cannot be discretised any coarser
    if (h->multi) hc->comm.borrow(h->comm);
    MgLevel Lc;
    Lc.h = hc;
    mg->lv.push_back(Lc);
    FL_CHK(fl_poisson_set_stream(hc, (void *)h->stream));
    // restriction weights of the fine cells: extent of the cell / extent of its parent
    for (int d = 0; d < 3; ++d) {
      mg->lv[l].r[d] = r[d];
      std::vector<double> w((size_t)hf->dec.len[d]);  // this rank's cells
      for (int64_t il = 0; il < hf->dec.len[d]; ++il) {
        const int64_t i = hf->dec.lo[d] + il, I = i / r[d];
        w[(size_t)il]   = (hf->ax[d].xf[(size_t)i + 1] - hf->ax[d].xf[(size_t)i]) / (xf[d][(size_t)I + 1] - xf[d][(size_t)I]);
      }
      FL_HIP(hipMalloc((void **)&mg->lv[l].w[d], sizeof(double) * w.size()));
      FL_HIP(hipMemcpy(mg->lv[l].w[d], w.data(), sizeof(double) * w.size(), hipMemcpyHostToDevice));
      // tri-linear prolongation: the parent's neighbour on the child's side and its weight, from the cell centres
      std::vector<int>    pn((size_t)hf->dec.len[d], 0);
      std::vector<double> pw((size_t)hf->dec.len[d], 0.);
      if (r[d] == 2) {
        const int64_t nc = hf->ax[d].n / 2;
        const double  L = xf[d][(size_t)nc] - xf[d][0];
        auto          Xc = [&](int64_t I) {  // coarse cell centre, periodic images included
          if (I < 0) return 0.5 * (xf[d][(size_t)(I + nc)] + xf[d][(size_t)(I + nc) + 1]) - L;
          if (I >= nc) return 0.5 * (xf[d][(size_t)(I - nc)] + xf[d][(size_t)(I - nc) + 1]) + L;
          return 0.5 * (xf[d][(size_t)I] + xf[d][(size_t)I + 1]);
        };
        for (int64_t il = 0; il < hf->dec.len[d]; ++il) {
          const int64_t i = hf->dec.lo[d] + il, I = i / 2, In = (i % 2) ? I + 1 : I - 1;
          if (!hf->ax[d].periodic && (In < 0 || In >= nc)) continue;  // no neighbour behind a wall: constant continuation
          const double xc = 0.5 * (hf->ax[d].xf[(size_t)i] + hf->ax[d].xf[(size_t)i + 1]);
          pn[(size_t)il] = (i % 2) ? 1 : -1;
          pw[(size_t)il] = (xc - Xc(I)) / (Xc(In) - Xc(I));
        }
      }
      FL_HIP(hipMalloc((void **)&mg->lv[l].pn[d], sizeof(int) * std::max<size_t>(pn.size(), 1)));
      FL_HIP(hipMalloc((void **)&mg->lv[l].pw[d], sizeof(double) * std::max<size_t>(pw.size(), 1)));
      FL_HIP(hipMemcpy(mg->lv[l].pn[d], pn.data(), sizeof(int) * pn.size(), hipMemcpyHostToDevice));
      FL_HIP(hipMemcpy(mg->lv[l].pw[d], pw.data(), sizeof(double) * pw.size(), hipMemcpyHostToDevice));
    }
  }
  for (size_t l = 0; l < mg->lv.size(); ++l) {
    MgLevel    &L = mg->lv[l];
    fl_poisson *hl = L.h;
    if (hl->nv_il > 1) return FL_ERR_SUP;  // the experimental row-interleaved vector layout is not wired into the cycle
    for (double **v : {&hl->r, &hl->P0, &hl->q, &hl->xp}) FL_CHK(fl_ensure_vec(hl, v));
    if (l + 1 == mg->lv.size()) {
      FL_CHK(alloc_cells(hl, &L.x));
      FL_CHK(alloc_cells(hl, &L.b));
    }
    FL_CHK(fl_ensure_partials(hl, MG_BLOCKS));
  }
  for (double **v : {&h->w0, &h->w1, &h->w2}) FL_CHK(fl_ensure_vec(h, v));  // outer CG: x, p, q
  FL_HIP(hipMalloc((void **)&mg->scal, sizeof(MgScal)));
  FL_HIP(hipMalloc((void **)&mg->slot_dev, 2 * sizeof(MgSlot)));
  FL_HIP(hipHostMalloc((void **)&mg->slot_host, 2 * sizeof(MgSlot)));
  for (hipEvent_t &e : mg->ev_slot) FL_HIP(hipEventCreateWithFlags(&e, hipEventDisableTiming));
  return 0;
}

// V-cycle on level l: right-hand side in the level handle's h->r, answer (zero initial guess) in its h->xp -- both padded
// sums (level 0 only; in: asked for, out: delivered): the five sums of k_mg_dots over (answer, right-hand side) left in h->sums by the last sweep
// subq (level 0 only): the right-hand side is first updated in place, b -= alpha * subq with the outer CG's alpha read from mg->scal on the device, on the
// first smoothing step's pass over it
int vcycle(fl_mg *mg, size_t l, const fl_ksp_opts *o, bool *sums = nullptr, const double *subq = nullptr)
{
  const bool want = sums && *sums;
  if (sums) *sums = false;
  MgLevel    &L = mg->lv[l];
  fl_poisson *h = L.h;
  if (l + 1 == mg->lv.size()) {
    // coarsest level: Jacobi-PCG to rtol 1e-2 through the public entry point (which uses the padded work vectors itself)
    fl_ksp_stats st;
    fl_ksp_opts  so;
    fl_ksp_opts_default(&so);
    so.remove_nullspace = o->remove_nullspace;
    so.type  = FL_KSP_CG;
    so.pc    = FL_PC_JACOBI;
    so.rtol  = 1e-2;
    so.maxit = 200;
    if (subq) hipLaunchKernelGGL(k_mg_pwd<5>, dim3(nblk_pairs(h->g)), dim3(256), 0, h->stream, h->g, (const MgScal *)mg->scal, (const double *)nullptr, subq, (double *)nullptr, h->r);
    if (!h->multi && h->ncell <= MG_COARSE_MAX && h->nv_il == 1 && fl_mg_coarse_mode() != 0) {
      // one workgroup, no host poll, straight from the padded right-hand side into the padded answer
      const int per = (h->ax[0].periodic ? 1 : 0) | (h->ax[1].periodic ? 2 : 0) | (h->ax[2].periodic ? 4 : 0);
      hipLaunchKernelGGL(k_mg_coarse_cg, dim3(1), dim3(256), 0, h->stream, h->g, per, so.remove_nullspace, so.rtol, so.atol, so.dtol, so.maxit, (const double *)h->r, h->xp);
      return 0;
    }
    launch_unpad_copy(h->stream, h->g, h->r, L.b, nullptr);
    // a one-level hierarchy runs this solve on the OUTER handle, whose h->r is the outer residual and is the inner
    // solve's work vector as well: keep it aside
    if (l == 0) {
      FL_CHK(fl_ensure_vec(h, &h->cd1));
      FL_HIP(hipMemcpyAsync(h->cd1, h->r, sizeof(double) * h->padlen, hipMemcpyDeviceToDevice, h->stream));
    }
    FL_CHK(fl_poisson_solve(h, L.b, L.x, &so, &st));
    if (l == 0) FL_HIP(hipMemcpyAsync(h->r, h->cd1, sizeof(double) * h->padlen, hipMemcpyDeviceToDevice, h->stream));
    launch_pad_copy(h->stream, h->g, L.x, h->xp);
    return 0;
  }
  const int nu = o->mg_smooth_its > 0 ? o->mg_smooth_its : 3;
  MgLevel  &C  = mg->lv[l + 1];
  FL_CHK(fl_cheb_smooth_padded(h, nu, true, true, nullptr, subq, subq ? &mg->scal->alpha : nullptr));  // x = smooth(b), zero initial guess
  int fused = 1;
  if (L.r[0] == 2 && L.r[1] == 2 && L.r[2] == 2) {
    fused = fl_residual_restrict_padded(h, h->xp, h->r, L.w[0], L.w[1], L.w[2], C.h, C.h->r);       // coarse b = R (b - S x) in one pass
    if (fused < 0) return fused;
  }
  if (fused != 0) {
    FL_CHK(fl_residual_padded(h, h->xp, h->r, h->q));                                // q = b - S x   (q: the smoother's scratch)
    hipLaunchKernelGGL(k_mg_restrict, dim3(nblk(C.h->ncell)), dim3(256), 0, h->stream, C.h->g, h->g, L.r[0], L.r[1], L.r[2], L.w[0], L.w[1], L.w[2], h->q, C.h->r);
  }
  FL_CHK(vcycle(mg, l + 1, o));                                                      // e_c = V(R r)
  if (fl_mg_prolong_mode() == 1) {  // x += P e_c, tri-linear: the coarse correction with its edge and corner ghosts
    FL_CHK(fl_fill_ghosts_full(C.h, C.h->xp));
    MgLin t;
    for (int d = 0; d < 3; ++d) {
      t.pn[d] = L.pn[d];
      t.pw[d] = L.pw[d];
    }
    const int tiled = FL_VARIANT(mg_prolong_tile, 2);  // 2 (shipped): parent-centred where every axis is halved, else 1: the tile walk; 0: the grid-stride kernel
    if (tiled >= 2 && L.r[0] == 2 && L.r[1] == 2 && L.r[2] == 2) {
      const int  kc = 4;
      const GridP &gcs = C.h->g;
      hipLaunchKernelGGL(k_mg_prolong_lin_cc, dim3((gcs.nx + 61) / 62, (gcs.ny + 3) / 4, (gcs.nz + kc - 1) / kc), dim3(256), 0, h->stream, h->g, gcs, t, (const double *)C.h->xp, h->xp, kc);
    } else if (tiled) {
      const int  zc = 8;
      const dim3 grid((h->g.nx + 127) / 128, (h->g.ny + 3) / 4, (h->g.nz + zc - 1) / zc);
      if (L.r[0] == 2 && L.r[1] == 2 && L.r[2] == 2) hipLaunchKernelGGL((k_mg_prolong_lin_tile<true>), grid, dim3(256), 0, h->stream, h->g, C.h->g, L.r[0], L.r[1], L.r[2], t, (const double *)C.h->xp, h->xp, zc);
      else hipLaunchKernelGGL((k_mg_prolong_lin_tile<false>), grid, dim3(256), 0, h->stream, h->g, C.h->g, L.r[0], L.r[1], L.r[2], t, (const double *)C.h->xp, h->xp, zc);
    } else hipLaunchKernelGGL(k_mg_prolong_lin_add, dim3(nblk_pairs(h->g)), dim3(256), 0, h->stream, h->g, C.h->g, L.r[0], L.r[1], L.r[2], t, (const double *)C.h->xp, h->xp);
  } else hipLaunchKernelGGL(k_mg_prolong_add, dim3(nblk_pairs(h->g)), dim3(256), 0, h->stream, h->g, C.h->g, L.r[0], L.r[1], L.r[2], C.h->xp, h->xp);  // x += P e_c
  bool got = want;
  const int nu_post = knob(K_mg_post_smooth) > 0 ? knob(K_mg_post_smooth) : nu;
  FL_CHK(fl_cheb_smooth_padded(h, nu_post, true, false, &got));                      // nu (or "mg_post_smooth") more steps from x
  if (sums) *sums = got;
  return 0;
}

}  // namespace

void fl_mg_destroy(fl_poisson *h)
{
  fl_mg *mg = h->mg;
  if (!mg) return;
  for (size_t l = 0; l < mg->lv.size(); ++l) {
    MgLevel &L = mg->lv[l];
    for (double *p : {L.x, L.b, L.w[0], L.w[1], L.w[2], L.pw[0], L.pw[1], L.pw[2]})
      if (p) (void)hipFree(p);
    for (int *p : L.pn)
      if (p) (void)hipFree(p);
    if (l > 0 && L.h) fl_poisson_destroy(L.h);
  }
  if (mg->scal) (void)hipFree(mg->scal);
  if (mg->slot_dev) (void)hipFree(mg->slot_dev);
  if (mg->slot_host) (void)hipHostFree(mg->slot_host);
  for (hipEvent_t e : mg->ev_slot)
    if (e) (void)hipEventDestroy(e);
  delete mg;
  h->mg = nullptr;
}

void fl_mg_set_stream(fl_poisson *h)
{
  if (!h->mg) return;
  for (size_t l = 1; l < h->mg->lv.size(); ++l) (void)fl_poisson_set_stream(h->mg->lv[l].h, (void *)h->stream);
}

// KSPCG with the V-cycle as (left) preconditioner; PETSc's KSPSolve_CG with KSP_NORM_PRECONDITIONED
int fl_solve_cg_mg(fl_poisson *h, const double *b, double *x, const fl_ksp_opts *o, fl_ksp_stats *st)
{
  if (h->multi && h->comm.kind == Comm::NONE) return FL_ERR_ARG_WRONGSTATE;
  if (o->norm_type != FL_NORM_PRECONDITIONED && o->norm_type != FL_NORM_UNPRECONDITIONED) return FL_ERR_SUP;
  // another depth than the hierarchy was built for: build again (round 5: comparing with the number of levels that EXIST never let a hierarchy grow
  // back once a solve had asked for a shallower one)
  if (h->mg && h->mg->requested != o->mg_levels) fl_mg_destroy(h);
  if (!h->mg) {
    FL_CHK(mg_build(h, o->mg_levels));
    h->mg->requested = o->mg_levels;
  }
  // the placement step (and nothing else) may have dropped the fine level's vectors since the hierarchy was built
  for (double **v : {&h->r, &h->P0, &h->q, &h->xp, &h->w0, &h->w1, &h->w2}) FL_CHK(fl_ensure_vec(h, v));
  fl_mg       *mg = h->mg;
  const GridP &g = h->g;
  const bool   ns = o->remove_nullspace != 0;
  const bool   pnorm = o->norm_type == FL_NORM_PRECONDITIONED;
  const double N = (double)h->ax[0].n * (double)h->ax[1].n * (double)h->ax[2].n;  // global cell count
  const int    nb = nblk_pairs(g);
  hipStream_t  s = h->stream;
  // the level-0 smoother runs on this very handle and may use h->ev0 / h->ev1: own events for the timing
  struct Events {  // released on every return path
    hipEvent_t a = nullptr, b = nullptr;
    ~Events()
    {
      if (a) (void)hipEventDestroy(a);
      if (b) (void)hipEventDestroy(b);
    }
  } ev;
  FL_HIP(hipEventCreate(&ev.a));
  FL_HIP(hipEventCreate(&ev.b));
  hipEvent_t e0 = ev.a, e1 = ev.b;
  FL_HIP(hipEventRecord(e0, s));
  double *X = h->w0, *P = h->w1, *Q = h->w2;  // padded; r = h->r (the cycle's right-hand side), z = h->xp after the cycle
  double  d[5], rz = 0., dp = 0., pq = 0.;
  int     bad_pq = 0;
  std::vector<double> hist;
  // z' = z - m 1 with m = mean(z):  z'.z' = z.z - N m^2,  r.z' = r.z - m sum r   (k_mg_scal<1>)
  const bool fused_dots = FL_VARIANT(mg_fused_dots, 1) != 0;  // 0: always the separate k_mg_dots pass (A/B runs)
  MgScal *S = mg->scal;
  {
    MgScal init;
    std::memset(&init, 0, sizeof(init));
    init.N      = N;
    init.ns     = ns ? 1 : 0;
    init.pnorm  = pnorm ? 1 : 0;
    init.rz_old = 1.;
    hipLaunchKernelGGL(k_mg_scal_set, dim3(1), dim3(1), 0, s, S, init);  // by value: nothing reads the stack frame after the launch
  }
  // the five sums of k_mg_dots over all ranks, left in h->sums (no host wait)
  auto dots_dev = [&](const double *z, const double *r) -> int {
    hipLaunchKernelGGL(k_mg_dots, dim3(nb), dim3(256), 0, s, g, z, r, h->partial, h->partial_stride);
    launch_reduce(s, h->partial, nb, h->partial_stride, 5, h->sums);
    if (h->multi) FL_CHK(h->comm.allreduce(s, h->sums, NSLOT));
    return 0;
  };
  const double *subq = nullptr;  // r -= alpha * subq is owed (taken care of inside the next cycle, alpha from the device)
  // [r -= alpha q ;] z = M^-1 r, the five sums, the scalars that follow from them, and their copy on the way to the host (slot `it & 1`)
  auto cycle_and_sums = [&](int it, bool first) -> int {
    bool got = fused_dots;
    FL_CHK(vcycle(mg, 0, o, &got, subq));
    if (!got) FL_CHK(dots_dev(h->xp, h->r));  // (else the cycle's last smoothing sweep formed the sums on its way)
    const int a = it & 1;
    if (first) hipLaunchKernelGGL((k_mg_scal<1, true>), dim3(1), dim3(1), 0, s, S, (const double *)h->sums, mg->slot_dev + a);
    else hipLaunchKernelGGL((k_mg_scal<1, false>), dim3(1), dim3(1), 0, s, S, (const double *)h->sums, mg->slot_dev + a);
    FL_HIP(hipMemcpyAsync(mg->slot_host + a, mg->slot_dev + a, sizeof(MgSlot), hipMemcpyDeviceToHost, s));
    FL_HIP(hipEventRecord(mg->ev_slot[a], s));
    return 0;
  };
  // the host's one look per iteration: by now the GPU holds the work that follows the cycle, so it does not idle while the host decides
  auto look = [&](int it) -> int {
    const int a = it & 1;
    FL_HIP(hipEventSynchronize(mg->ev_slot[a]));
    const MgSlot &L = mg->slot_host[a];
    dp     = L.dp;
    rz     = L.rz;
    pq     = L.pq;
    bad_pq = L.bad_pq;
    return 0;
  };
  launch_pad_copy(s, g, b, h->r);                                           // r = b (x = 0)
  FL_CHK(fl_zero_vec(h, X));
  FL_CHK(cycle_and_sums(0, true));
  hipLaunchKernelGGL(k_mg_pwd<2>, dim3(nb), dim3(256), 0, s, g, (const MgScal *)S, (const double *)h->xp, (const double *)nullptr, P, (double *)nullptr);  // p = z'
  FL_CHK(look(0));
  const double rnorm0 = dp, ttol = std::max(o->rtol * dp, o->atol);
  hist.push_back(dp);
  int  it = 0, reason = 0;
  auto converged = [&](double v) {
    if (std::isnan(v) || std::isinf(v)) return (int)FL_DIVERGED_NANORINF;
    if (v <= ttol) return v < o->atol ? (int)FL_CONVERGED_ATOL : (int)FL_CONVERGED_RTOL;
    if (v >= o->dtol * rnorm0) return (int)FL_DIVERGED_DTOL;
    return 0;
  };
  reason = converged(dp);
  if (!reason && o->maxit <= 0) reason = FL_DIVERGED_ITS;
  const bool flexible = fl_mg_flexible_mode() != 0;
  while (!reason) {
    FL_CHK(fl_apply_padded_dot(h, P, Q, nullptr));                          // q = S p ; p.q stays in h->sums[2]
    hipLaunchKernelGGL((k_mg_scal<0>), dim3(1), dim3(1), 0, s, S, (const double *)h->sums, (MgSlot *)nullptr);  // alpha = r.z / p.q
    // x += alpha p ; r -= alpha q -- neither as a pass of its own: r on the first smoothing step of the cycle that follows, x together with p below
    subq = Q;
    FL_CHK(cycle_and_sums(it + 1, false));                                  // r -= alpha q, z = M^-1 r, the five sums, m, ||z'||, r.z'
    subq = nullptr;
    // The direction update is enqueued BEFORE the host has looked at the norm: x += alpha p is owed whether the iteration stops or not, and a
    // p that is rewritten after the last iteration is never read.  beta in the Polak-Ribiere way: z'.(r_new - r_old) = -alpha q.z'
    if (flexible) {
      FL_CHK(dots_dev(h->xp, Q));
      hipLaunchKernelGGL((k_mg_scal<2>), dim3(1), dim3(1), 0, s, S, (const double *)h->sums, (MgSlot *)nullptr);
    } else hipLaunchKernelGGL((k_mg_scal<3>), dim3(1), dim3(1), 0, s, S, (const double *)h->sums, (MgSlot *)nullptr);
    hipLaunchKernelGGL(k_mg_pwd<4>, dim3(nb), dim3(256), 0, s, g, (const MgScal *)S, (const double *)h->xp, (const double *)nullptr, P, X);  // x += alpha p ; p = z' + beta p
    FL_CHK(look(it + 1));
    if (bad_pq) {  // p.q <= 0 or NaN: alpha was 0 on the device, the cycle and the update behind it changed nothing; the iteration does not count
      reason = std::isnan(pq) ? FL_DIVERGED_NANORINF : FL_DIVERGED_INDEFINITE_MAT;
      break;
    }
    ++it;
    hist.push_back(dp);
    reason = converged(dp);
    if (!reason && it >= o->maxit) reason = FL_DIVERGED_ITS;
    if (!reason && !(rz > 0.)) reason = std::isnan(rz) ? FL_DIVERGED_NANORINF : FL_DIVERGED_INDEFINITE_PC;
  }
  // answer, with the constant removed on the way out (the shift is handed over in device memory)
  double *shift = nullptr;
  if (ns) {
    FL_CHK(dots(h, X, X, d));
    const double mx = d[0] / N;
    FL_HIP(hipMemcpyAsync(h->sums, &mx, sizeof(double), hipMemcpyHostToDevice, s));
    FL_HIP(hipStreamSynchronize(s));  // mx lives on this stack frame
    shift = h->sums;
  }
  launch_unpad_copy(s, g, X, x, shift);
  FL_HIP(hipEventRecord(e1, s));
  FL_HIP(hipStreamSynchronize(s));
  float ms = 0.f;
  FL_HIP(hipEventElapsedTime(&ms, e0, e1));
  st->iters   = it;
  st->reason  = reason;
  if (reason == FL_DIVERGED_NANORINF || reason == FL_DIVERGED_DTOL || !std::isfinite(dp)) {
    for (MgLevel &L : mg->lv) L.h->poisoned = true;  // work vectors of every level may hold NaN: cleared before their next CG solve
  } else {
    // the unmonitored smoothing sweeps mark their handle (fl_cheb_smooth_padded: no norm is looked at in there), but the outer iteration did
    // look: a finite ||z|| after the last cycle says that no level's work vector holds a NaN or an Inf, so nothing needs clearing
    for (MgLevel &L : mg->lv) L.h->poisoned = false;
  }
  st->rnorm0  = rnorm0;
  st->rnorm   = dp;
  st->seconds = ms * 1e-3;
  if (o->history && o->nhistory > 0) {
    const int n = std::min<int>(o->nhistory, (int)hist.size());
    std::memcpy(o->history, hist.data(), sizeof(double) * n);
  }
  return 0;
}
