// fl_kernels.hip -- gfx950 kernels of the pressure-Poisson path (fp64, HBM-bound, no MFMA).
//
// Data layout: every solver-internal vector is a PADDED block (nx+ghosts) x (ny+2) x (nz+2), x fastest, row stride
// sx (multiple of 16 doubles, cell i=0 at a 128-B boundary), one ghost layer (star stencil, width 1 -- the reference's
// DMStag stencil, fluca/src/mesh/impl/cart/cart.c:66,91).  Ghosts hold the periodic image / the neighbour rank's
// cells / nothing (walls: the matching stencil coefficient is 0 and 1/diag is 0).  With that, S p is a branch-free
// 7-point stencil whose coefficients come from nine 1-D tables (fl_coeff.cpp).
//
// Hot kernels (Jacobi-PCG, KSPSolve(kspS) of fluca/src/ns/utils/abfpc/abfpc.c:77):
//   k_cg_A  p' = z + beta p (z = r/diag - mean), q = S p', partial p'.q, and the deferred x += alpha_prev p:
//           reads r,p,x  writes p',q,x  = 48 B/cell.  128 x (4*RY) tiles marching in z; p' planes staged in LDS
//           (2 buffers, one barrier per plane), z-neighbours in registers, global loads prefetched one plane ahead.
//   k_cg_B  r -= alpha q with the five partial sums the PETSc-style convergence test needs: 24 B/cell.
// Together 72 B/cell/iteration against the 88 B/cell of the textbook sequence.
//   k_cg_Bq the same update with q = S p' formed again from the stored direction (k_cg_A<SQ = false> writes no q): the
//           pair moves 40 + 24 = 64 B/cell/iteration (+ tile rings).  The solver's default since round 2.
#include "fl_internal.h"
#include "fl_stencil.h"
#include "fl_knobs.h"

#ifndef FL_CGA_WPE
#define FL_CGA_WPE 2  // waves per SIMD the register allocator must leave room for: 2 = up to 256 VGPRs (one 512-thread block per CU), 4 = at most 128 (two blocks)
#endif
#ifndef FL_CGB_WPE
#define FL_CGB_WPE 2
#endif
namespace fl {

// ------------------------------------------------------------------------------------------------ helpers

__device__ __forceinline__ int64_t pidx(const GridP &g, int i, int j, int k) { return g.off0 + (int64_t)k * g.sxy + (int64_t)j * g.sx + i; }

__device__ __forceinline__ double wave_sum(double v)
{
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
  return v;
}

// sum over the 256-thread block; result valid in thread 0.  Fixed order -> deterministic.
template <int NV>
__device__ __forceinline__ void block_sum(double (&v)[NV], double *red /* [NV][4] */)
{
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
#pragma unroll
  for (int a = 0; a < NV; ++a) {
    v[a] = wave_sum(v[a]);
    if (lane == 0) red[a * 4 + w] = v[a];
  }
  __syncthreads();
  if (threadIdx.x == 0) {
#pragma unroll
    for (int a = 0; a < NV; ++a) v[a] = (red[a * 4 + 0] + red[a * 4 + 1]) + (red[a * 4 + 2] + red[a * 4 + 3]);
  }
}

// ------------------------------------------------------------------------------------------------ generic kernels
// launch shape for the generic kernels: block (64,4), grid (ceil(nx/64), ceil(ny/4), nz-range)

__global__ void k_pad_copy(GridP g, const double *__restrict__ src, double *__restrict__ dst)
{
  const int i = blockIdx.x * 64 + threadIdx.x, j = blockIdx.y * 4 + threadIdx.y, k = blockIdx.z;
  if (i < g.nx && j < g.ny) dst[pidx(g, i, j, k)] = src[((int64_t)k * g.ny + j) * g.nx + i];
}

// dst = src_pad - *shift   (shift may be NULL)
__global__ void k_unpad_copy(GridP g, const double *__restrict__ src, double *__restrict__ dst, const double *__restrict__ shift)
{
  const int i = blockIdx.x * 64 + threadIdx.x, j = blockIdx.y * 4 + threadIdx.y, k = blockIdx.z;
  const double s = shift ? *shift : 0.;
  if (i < g.nx && j < g.ny) dst[((int64_t)k * g.ny + j) * g.nx + i] = src[pidx(g, i, j, k)] - s;
}

// periodic image inside one rank: ghost(-1) = cell(n-1), ghost(n) = cell(0) along `axis`
// blockIdx.z: which of several vectors laid out `vstride` doubles apart (the three components of a velocity: one launch instead of three)
__global__ void k_wrap_ghosts(GridP g, double *__restrict__ v0_, int axis, int64_t vstride)
{
  double *__restrict__ v = v0_ + (int64_t)blockIdx.z * vstride;
  const int a = blockIdx.x * 64 + threadIdx.x, b = blockIdx.y * 4 + threadIdx.y;
  int       na, nb;
  if (axis == 0) { na = g.ny; nb = g.nz; }
  else if (axis == 1) { na = g.nx; nb = g.nz; }
  else { na = g.nx; nb = g.ny; }
  if (a >= na || b >= nb) return;
  if (axis == 0) {
    v[pidx(g, -1, a, b)]   = v[pidx(g, g.nx - 1, a, b)];
    v[pidx(g, g.nx, a, b)] = v[pidx(g, 0, a, b)];
  } else if (axis == 1) {
    v[pidx(g, a, -1, b)]   = v[pidx(g, a, g.ny - 1, b)];
    v[pidx(g, a, g.ny, b)] = v[pidx(g, a, 0, b)];
  } else {
    v[pidx(g, a, b, -1)]   = v[pidx(g, a, b, g.nz - 1)];
    v[pidx(g, a, b, g.nz)] = v[pidx(g, a, b, 0)];
  }
}

// Dimension-by-dimension ghost fill (edges and corners included): axis d is handled AFTER the axes before it, over a face that is
// wider by the ghost layers those axes have just received -- ea / eb = 1 extends the first / second in-face direction by one cell on
// either side.  mode 0: periodic wrap inside the block, 1: pack the owned boundary layer `side`, 2: unpack into the ghost layer `side`.
__global__ void k_face_ext(GridP g, double *__restrict__ v, double *__restrict__ buf, int axis, int side, int ea, int eb, int mode)
{
  const int na = (axis == 0 ? g.ny : g.nx) + 2 * ea, nb = (axis == 2 ? g.ny : g.nz) + 2 * eb;
  const int ta = blockIdx.x * 64 + threadIdx.x, tb = blockIdx.y * 4 + threadIdx.y;
  if (ta >= na || tb >= nb) return;
  const int a = ta - ea, b = tb - eb;
  const int n = axis == 0 ? g.nx : (axis == 1 ? g.ny : g.nz);
  auto      at = [&](int c) { return axis == 0 ? pidx(g, c, a, b) : (axis == 1 ? pidx(g, a, c, b) : pidx(g, a, b, c)); };
  if (mode == 0) {
    v[at(-1)] = v[at(n - 1)];
    v[at(n)]  = v[at(0)];
  } else if (mode == 1) buf[(int64_t)tb * na + ta] = v[at(side ? n - 1 : 0)];
  else v[at(side ? n : -1)] = buf[(int64_t)tb * na + ta];
}

// The same for `dp` layers (a two-deep ghost exchange: the fused two-step smoother on several ranks): ea / eb = how many cells the face is
// extended on either side of the first / second in-face direction (0 .. 2: the ghost layers the axes before this one have filled).
// Layer l = 0 is the one next to the boundary: pack reads the owned layer l from `side`, unpack writes the ghost layer l beyond it;
// buffer [l][tb][ta].  mode 1 pack, 2 unpack.
__global__ void k_face_ext_deep(GridP g, double *__restrict__ v, double *__restrict__ buf, int axis, int side, int ea, int eb, int dp, int mode)
{
  const int na = (axis == 0 ? g.ny : g.nx) + 2 * ea, nb = (axis == 2 ? g.ny : g.nz) + 2 * eb;
  const int ta = blockIdx.x * 64 + threadIdx.x, tb = blockIdx.y * 4 + threadIdx.y;
  if (ta >= na || tb >= nb) return;
  const int a = ta - ea, b = tb - eb;
  const int n = axis == 0 ? g.nx : (axis == 1 ? g.ny : g.nz);
  auto      at = [&](int c) { return axis == 0 ? pidx(g, c, a, b) : (axis == 1 ? pidx(g, a, c, b) : pidx(g, a, b, c)); };
  for (int l = 0; l < dp; ++l) {
    const int64_t q = ((int64_t)l * nb + tb) * na + ta;
    if (mode == 1) buf[q] = v[at(side ? n - 1 - l : l)];
    else v[at(side ? n + l : -1 - l)] = buf[q];
  }
}

// faces <-> contiguous buffers (multi-rank halo exchange).  side 0 = low, 1 = high.  pack reads owned boundary cells,
// unpack writes the ghost layer.
__global__ void k_pack_face(GridP g, const double *__restrict__ v, double *__restrict__ buf, int axis, int side)
{
  const int a = blockIdx.x * 64 + threadIdx.x, b = blockIdx.y * 4 + threadIdx.y;
  int       na, nb;
  if (axis == 0) { na = g.ny; nb = g.nz; }
  else if (axis == 1) { na = g.nx; nb = g.nz; }
  else { na = g.nx; nb = g.ny; }
  if (a >= na || b >= nb) return;
  const int n = axis == 0 ? g.nx : (axis == 1 ? g.ny : g.nz);
  const int c = side ? n - 1 : 0;
  int64_t   p = axis == 0 ? pidx(g, c, a, b) : (axis == 1 ? pidx(g, a, c, b) : pidx(g, a, b, c));
  buf[(int64_t)b * na + a] = v[p];
}
__global__ void k_unpack_face(GridP g, double *__restrict__ v, const double *__restrict__ buf, int axis, int side)
{
  const int a = blockIdx.x * 64 + threadIdx.x, b = blockIdx.y * 4 + threadIdx.y;
  int       na, nb;
  if (axis == 0) { na = g.ny; nb = g.nz; }
  else if (axis == 1) { na = g.nx; nb = g.nz; }
  else { na = g.nx; nb = g.ny; }
  if (a >= na || b >= nb) return;
  const int n = axis == 0 ? g.nx : (axis == 1 ? g.ny : g.nz);
  const int c = side ? n : -1;
  int64_t   p = axis == 0 ? pidx(g, c, a, b) : (axis == 1 ? pidx(g, a, c, b) : pidx(g, a, b, c));
  v[p]        = buf[(int64_t)b * na + a];
}

// all faces of one ghost exchange in ONE launch: blockIdx.z = boundary 0..5; buf[b] == NULL -> that boundary is not exchanged
struct FaceBufs {
  double *buf[6];
};
__global__ void k_pack_faces(GridP g, const double *__restrict__ v, FaceBufs fb)
{
  const int bnd = blockIdx.z, axis = bnd / 2, side = bnd % 2;
  double   *buf = fb.buf[bnd];
  if (!buf) return;
  const int a = blockIdx.x * 64 + threadIdx.x, b = blockIdx.y * 4 + threadIdx.y;
  const int na = axis == 0 ? g.ny : g.nx, nb = axis == 2 ? g.ny : g.nz;
  if (a >= na || b >= nb) return;
  const int     n = axis == 0 ? g.nx : (axis == 1 ? g.ny : g.nz), c = side ? n - 1 : 0;
  const int64_t p = axis == 0 ? pidx(g, c, a, b) : (axis == 1 ? pidx(g, a, c, b) : pidx(g, a, b, c));
  buf[(int64_t)b * na + a] = v[p];
}
// The boundary layers of the NEW residual r - alpha q, packed before k_cg_B has formed it anywhere: the halo exchange then runs on
// a second stream while k_cg_B streams the whole block (same fma as k_cg_B, so the neighbour's ghost equals this rank's cell bit
// for bit).  After the iteration has stopped (reason != 0) k_cg_B leaves r alone: so does this.
__global__ void k_pack_faces_rq(GridP g, const double *__restrict__ r, const double *__restrict__ q, const KspScal *__restrict__ s, FaceBufs fb)
{
  const int bnd = blockIdx.z, axis = bnd / 2, side = bnd % 2;
  double   *buf = fb.buf[bnd];
  if (!buf) return;
  const int a = blockIdx.x * 64 + threadIdx.x, b = blockIdx.y * 4 + threadIdx.y;
  const int na = axis == 0 ? g.ny : g.nx, nb = axis == 2 ? g.ny : g.nz;
  if (a >= na || b >= nb) return;
  const double  alpha = s->reason != 0 ? 0. : s->alpha;
  const int     n = axis == 0 ? g.nx : (axis == 1 ? g.ny : g.nz), c = side ? n - 1 : 0;
  const int64_t p = axis == 0 ? pidx(g, c, a, b) : (axis == 1 ? pidx(g, a, c, b) : pidx(g, a, b, c));
  buf[(int64_t)b * na + a] = s->reason != 0 ? r[p] : fma(-alpha, q[p], r[p]);
}
// The same for the single-reduction CG (MODE 9 / 10 of st_body, fl_ksp.hip): the boundary layers of the NEW residual r - a (S + b W), from the S = A z
// MODE 9 kept on those layers, packed before MODE 10 forms it for the whole block -- the same two fma, so a neighbour's ghost equals this
// rank's cell bit for bit.
__global__ void k_pack_faces_sr(GridP g, const double *__restrict__ r, const double *__restrict__ sb, const double *__restrict__ W, const KspScal *__restrict__ s, FaceBufs fb)
{
  const int bnd = blockIdx.z, axis = bnd / 2, side = bnd % 2;
  double   *buf = fb.buf[bnd];
  if (!buf) return;
  const int a = blockIdx.x * 64 + threadIdx.x, b = blockIdx.y * 4 + threadIdx.y;
  const int na = axis == 0 ? g.ny : g.nx, nb = axis == 2 ? g.ny : g.nz;
  if (a >= na || b >= nb) return;
  const int     n = axis == 0 ? g.nx : (axis == 1 ? g.ny : g.nz), c = side ? n - 1 : 0;
  const int64_t p = axis == 0 ? pidx(g, c, a, b) : (axis == 1 ? pidx(g, a, c, b) : pidx(g, a, b, c));
  buf[(int64_t)b * na + a] = s->reason != 0 ? r[p] : fma(-s->alpha, fma(s->beta, W[p], sb[p]), r[p]);
}
__global__ void k_unpack_faces(GridP g, double *__restrict__ v, FaceBufs fb)
{
  const int     bnd = blockIdx.z, axis = bnd / 2, side = bnd % 2;
  const double *buf = fb.buf[bnd];
  if (!buf) return;
  const int a = blockIdx.x * 64 + threadIdx.x, b = blockIdx.y * 4 + threadIdx.y;
  const int na = axis == 0 ? g.ny : g.nx, nb = axis == 2 ? g.ny : g.nz;
  if (a >= na || b >= nb) return;
  const int     n = axis == 0 ? g.nx : (axis == 1 ? g.ny : g.nz), c = side ? n : -1;
  const int64_t p = axis == 0 ? pidx(g, c, a, b) : (axis == 1 ? pidx(g, a, c, b) : pidx(g, a, b, c));
  v[p]            = buf[(int64_t)b * na + a];
}

__device__ __forceinline__ double stencil7(const GridP &g, const double *__restrict__ x, int i, int j, int k)
{
  const int64_t c = pidx(g, i, j, k);
  const double  d = g.sc[0][i] + g.sc[1][j] + g.sc[2][k];
  return d * x[c] + g.sl[0][i] * x[c - 1] + g.sh[0][i] * x[c + 1] + g.sl[1][j] * x[c - g.sx] + g.sh[1][j] * x[c + g.sx] + g.sl[2][k] * x[c - g.sxy] + g.sh[2][k] * x[c + g.sxy];
}

// y = S x ; x padded with valid ghosts; y unpadded (ypad == 0) or padded
__global__ void k_apply(GridP g, const double *__restrict__ x, double *__restrict__ y, int ypad)
{
  const int i = blockIdx.x * 64 + threadIdx.x, j = blockIdx.y * 4 + threadIdx.y, k = blockIdx.z;
  if (i >= g.nx || j >= g.ny) return;
  const double v = stencil7(g, x, i, j, k);
  if (ypad) y[pidx(g, i, j, k)] = v;
  else y[((int64_t)k * g.ny + j) * g.nx + i] = v;
}

__global__ void k_diagonal(GridP g, double *__restrict__ d)
{
  const int i = blockIdx.x * 64 + threadIdx.x, j = blockIdx.y * 4 + threadIdx.y, k = blockIdx.z;
  if (i < g.nx && j < g.ny) d[((int64_t)k * g.ny + j) * g.nx + i] = g.sc[0][i] + g.sc[1][j] + g.sc[2][k];
}

// b = contrhs - D V   (abfpc.c:75-76).  Faces unpadded (layout in fluca_hip.h); ghost faces: the high-side face of the last
// owned cell belongs to the next rank (or wraps) -> passed in hi[d] (plane arrays) when this rank does not own it.
__global__ void k_rhs(GridP g, const double *__restrict__ Vx, const double *__restrict__ Vy, const double *__restrict__ Vz, const double *__restrict__ hix, const double *__restrict__ hiy,
                      const double *__restrict__ hiz, const double *__restrict__ contrhs, double *__restrict__ b)
{
  const int i = blockIdx.x * 64 + threadIdx.x, j = blockIdx.y * 4 + threadIdx.y, k = blockIdx.z;
  if (i >= g.nx || j >= g.ny) return;
  const double xl = Vx[((int64_t)k * g.ny + j) * g.fx + i];
  const double xh = (i + 1 < g.fx) ? Vx[((int64_t)k * g.ny + j) * g.fx + i + 1] : hix[(int64_t)k * g.ny + j];
  const double yl = Vy[((int64_t)k * g.fy + j) * g.nx + i];
  const double yh = (j + 1 < g.fy) ? Vy[((int64_t)k * g.fy + j + 1) * g.nx + i] : hiy[(int64_t)k * g.nx + i];
  const double zl = Vz[((int64_t)k * g.ny + j) * g.nx + i];
  const double zh = (k + 1 < g.fz) ? Vz[((int64_t)(k + 1) * g.ny + j) * g.nx + i] : hiz[(int64_t)j * g.nx + i];
  const double div = (xh - xl) * g.idx[0][i] + (yh - yl) * g.idx[1][j] + (zh - zl) * g.idx[2][k];
  const int64_t c  = ((int64_t)k * g.ny + j) * g.nx + i;
  b[c]             = (contrhs ? contrhs[c] : 0.) - div;
}

// first plane of the face array along `axis` -> contiguous buffer (the low-side neighbour needs it as its hi face)
__global__ void k_face_plane0(GridP g, const double *__restrict__ V, double *__restrict__ buf, int axis)
{
  const int a = blockIdx.x * 64 + threadIdx.x, b = blockIdx.y * 4 + threadIdx.y;
  int       na, nb;
  if (axis == 0) { na = g.ny; nb = g.nz; }
  else if (axis == 1) { na = g.nx; nb = g.nz; }
  else { na = g.nx; nb = g.ny; }
  if (a >= na || b >= nb) return;
  int64_t p = axis == 0 ? ((int64_t)b * g.ny + a) * g.fx : (axis == 1 ? ((int64_t)b * g.fy) * g.nx + a : (int64_t)b * g.nx + a);
  buf[(int64_t)b * na + a] = V[p];
}

// V_d -= kappa * Gst p on the owned faces of axis d (p padded with ghosts)
#ifdef FL_KBENCH_VARIANTS  // round 1: one projection kernel per output array
__global__ void k_project_faces(GridP g, const double *__restrict__ p, double *__restrict__ V, int axis)
{
  const int i = blockIdx.x * 64 + threadIdx.x, j = blockIdx.y * 4 + threadIdx.y, k = blockIdx.z;
  const int lx = axis == 0 ? g.fx : g.nx, ly = axis == 1 ? g.fy : g.ny;
  if (i >= lx || j >= ly) return;
  const int     f  = axis == 0 ? i : (axis == 1 ? j : k);
  const int     c0 = g.gc0[axis][f];
  const int64_t st = axis == 0 ? 1 : (axis == 1 ? (int64_t)g.sx : g.sxy);
  const int64_t pc = axis == 0 ? pidx(g, c0, j, k) : (axis == 1 ? pidx(g, i, c0, k) : pidx(g, i, j, c0));
  const double  gr = g.ga0[axis][f] * p[pc] + g.ga1[axis][f] * p[pc + st];
  const int64_t fi = axis == 0 ? ((int64_t)k * g.ny + j) * g.fx + i : (axis == 1 ? ((int64_t)k * g.fy + j) * g.nx + i : ((int64_t)k * g.ny + j) * g.nx + i);
  V[fi] -= g.kappa * gr;
}

// v_d -= kappa * (G p)_d at cell centres.  p padded with TWO-deep access only at physical walls (inside the block), so
// one ghost layer is enough: the 3-point one-sided rows start at the wall cell itself.
__global__ void k_project_cells(GridP g, const double *__restrict__ p, double *__restrict__ v, int axis)
{
  const int i = blockIdx.x * 64 + threadIdx.x, j = blockIdx.y * 4 + threadIdx.y, k = blockIdx.z;
  if (i >= g.nx || j >= g.ny) return;
  const int     c  = axis == 0 ? i : (axis == 1 ? j : k);
  const int     s0 = g.Gs[axis][c];
  const int64_t st = axis == 0 ? 1 : (axis == 1 ? (int64_t)g.sx : g.sxy);
  const int64_t pc = axis == 0 ? pidx(g, s0, j, k) : (axis == 1 ? pidx(g, i, s0, k) : pidx(g, i, j, s0));
  double        gr = g.Gv0[axis][c] * p[pc] + g.Gv1[axis][c] * p[pc + st];
  const double  v2 = g.Gv2[axis][c];
  if (v2 != 0.) gr += v2 * p[pc + 2 * st];
  v[((int64_t)k * g.ny + j) * g.nx + i] -= g.kappa * gr;
}
#endif  // FL_KBENCH_VARIANTS

// The whole stage-2 update of PCApply_ABF (abfpc.c:79-101) in ONE pass over p:  v_d -= kappa (G p)_d on the cells and V_d -= kappa (Gst p)_d on the
// owned faces of all three axes -- the rows and the arithmetic of k_project_cells / k_project_faces (same products, same order: bit-identical), but p is
// streamed once instead of six times and the three cell arrays / three face arrays are each touched by one kernel instead of their own.  A wave owns a
// 64-cell segment of a row; it keeps its segment (the grid is a multiple of the segments per row), so the x rows are loaded once per wave, and row /
// plane numbers are wave-uniform (scalar table reads).  A cell also does the face behind the last cell of an axis where this rank owns that face.
// Any of the six outputs may be NULL.
struct ProjOut {
  double *v[3], *V[3];
};
// a lane owns two x-adjacent cells: 16-byte accesses to p along y / z, to the cell arrays and to the y- / z-face arrays (pairs: the caller's arrays are
// 16-byte aligned and nx is even; the x-face array has rows of nx + 1 entries and keeps 8-byte accesses), 8-byte reads of p along x
__global__ void __launch_bounds__(256) k_project_all(GridP g, const double *__restrict__ p, ProjOut o, int pairs)
{
  const int lane = threadIdx.x & 63, w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), nw = blockDim.x >> 6;
  const int nseg = (g.nx + 127) / 128;
  const int gw = (int)blockIdx.x * nw + w, nwaves = (int)gridDim.x * nw;  // nwaves is a multiple of nseg (launch_project_all)
  const int seg = gw % nseg, i = seg * 128 + 2 * lane;
  if (i >= g.nx) return;
  const bool   two = i + 1 < g.nx, pr2 = two && pairs;
  const int    i1 = two ? i + 1 : i;
  const double kap = g.kappa;
  // x rows of this lane's cells / their low faces (and of face nx behind the last cell)
  const int    xs[2] = {g.Gs[0][i], g.Gs[0][i1]};
  const double xg0[2] = {g.Gv0[0][i], g.Gv0[0][i1]}, xg1[2] = {g.Gv1[0][i], g.Gv1[0][i1]}, xg2[2] = {g.Gv2[0][i], g.Gv2[0][i1]};
  const int    xc[2] = {g.gc0[0][i], g.gc0[0][i1]};
  const double xa0[2] = {g.ga0[0][i], g.ga0[0][i1]}, xa1[2] = {g.ga1[0][i], g.ga1[0][i1]};
  const int    il = two ? i + 1 : i;  // the lane's last cell
  const bool   xlast = il == g.nx - 1 && g.fx > g.nx;
  const int    xc0n = xlast ? g.gc0[0][g.nx] : 0;
  const double xa0n = xlast ? g.ga0[0][g.nx] : 0., xa1n = xlast ? g.ga1[0][g.nx] : 0.;
  auto sub = [&](double *a, int64_t q, double2 d) {  // a[q], a[q + 1] -= kappa * d
    if (pr2) {
      double2 t = *reinterpret_cast<double2 *>(a + q);
      t.x -= kap * d.x;
      t.y -= kap * d.y;
      *reinterpret_cast<double2 *>(a + q) = t;
    } else {
      a[q] -= kap * d.x;
      if (two) a[q + 1] -= kap * d.y;
    }
  };
  const int64_t nrows = (int64_t)g.ny * g.nz;
  for (int64_t row = gw / nseg; row < nrows; row += nwaves / nseg) {
    const int     j = (int)(row % g.ny), k = (int)(row / g.ny);
    const int64_t pr = g.off0 + (int64_t)k * g.sxy + (int64_t)j * g.sx;  // cell (0, j, k) of the padded p
    const int64_t cell = ((int64_t)k * g.ny + j) * g.nx + i;
    // ---- x
    if (o.v[0]) {
      double2 gr;
      gr.x = xg0[0] * p[pr + xs[0]] + xg1[0] * p[pr + xs[0] + 1];
      if (xg2[0] != 0.) gr.x += xg2[0] * p[pr + xs[0] + 2];
      gr.y = xg0[1] * p[pr + xs[1]] + xg1[1] * p[pr + xs[1] + 1];
      if (xg2[1] != 0.) gr.y += xg2[1] * p[pr + xs[1] + 2];
      sub(o.v[0], cell, gr);
    }
    if (o.V[0]) {
      const int64_t fi = ((int64_t)k * g.ny + j) * g.fx + i;
      o.V[0][fi] -= kap * (xa0[0] * p[pr + xc[0]] + xa1[0] * p[pr + xc[0] + 1]);
      if (two) o.V[0][fi + 1] -= kap * (xa0[1] * p[pr + xc[1]] + xa1[1] * p[pr + xc[1] + 1]);
      if (xlast) o.V[0][fi + (two ? 2 : 1)] -= kap * (xa0n * p[pr + xc0n] + xa1n * p[pr + xc0n + 1]);
    }
    // ---- y (wave-uniform rows); the padded p is 16-byte aligned at even i
    auto ldp = [&](int64_t q) { return two ? *reinterpret_cast<const double2 *>(p + q) : make_double2(p[q], 0.); };
    if (o.v[1]) {
      const int     s0 = g.Gs[1][j];
      const int64_t pc = pr + (int64_t)(s0 - j) * g.sx + i;
      const double  c0 = g.Gv0[1][j], c1 = g.Gv1[1][j], c2 = g.Gv2[1][j];
      const double2 a = ldp(pc), b = ldp(pc + g.sx);
      double2       gr = make_double2(c0 * a.x + c1 * b.x, c0 * a.y + c1 * b.y);
      if (c2 != 0.) {
        const double2 c = ldp(pc + 2 * (int64_t)g.sx);
        gr.x += c2 * c.x;
        gr.y += c2 * c.y;
      }
      sub(o.v[1], cell, gr);
    }
    if (o.V[1]) {
      const int     c0 = g.gc0[1][j];
      const int64_t pc = pr + (int64_t)(c0 - j) * g.sx + i;
      const int64_t fi = ((int64_t)k * g.fy + j) * g.nx + i;
      const double  a0 = g.ga0[1][j], a1 = g.ga1[1][j];
      const double2 a = ldp(pc), b = ldp(pc + g.sx);
      sub(o.V[1], fi, make_double2(a0 * a.x + a1 * b.x, a0 * a.y + a1 * b.y));
      if (j == g.ny - 1 && g.fy > g.ny) {
        const int     c0n = g.gc0[1][g.ny];
        const int64_t pn  = pr + (int64_t)(c0n - j) * g.sx + i;
        const double  n0 = g.ga0[1][g.ny], n1 = g.ga1[1][g.ny];
        const double2 an = ldp(pn), bn = ldp(pn + g.sx);
        sub(o.V[1], fi + g.nx, make_double2(n0 * an.x + n1 * bn.x, n0 * an.y + n1 * bn.y));
      }
    }
    // ---- z
    if (o.v[2]) {
      const int     s0 = g.Gs[2][k];
      const int64_t pc = pr + (int64_t)(s0 - k) * g.sxy + i;
      const double  c0 = g.Gv0[2][k], c1 = g.Gv1[2][k], c2 = g.Gv2[2][k];
      const double2 a = ldp(pc), b = ldp(pc + g.sxy);
      double2       gr = make_double2(c0 * a.x + c1 * b.x, c0 * a.y + c1 * b.y);
      if (c2 != 0.) {
        const double2 c = ldp(pc + 2 * g.sxy);
        gr.x += c2 * c.x;
        gr.y += c2 * c.y;
      }
      sub(o.v[2], cell, gr);
    }
    if (o.V[2]) {
      const int     c0 = g.gc0[2][k];
      const int64_t pc = pr + (int64_t)(c0 - k) * g.sxy + i;
      const double  a0 = g.ga0[2][k], a1 = g.ga1[2][k];
      const double2 a = ldp(pc), b = ldp(pc + g.sxy);
      sub(o.V[2], cell, make_double2(a0 * a.x + a1 * b.x, a0 * a.y + a1 * b.y));
      if (k == g.nz - 1 && g.fz > g.nz) {
        const int     c0n = g.gc0[2][g.nz];
        const int64_t pn  = pr + (int64_t)(c0n - k) * g.sxy + i;
        const double  n0 = g.ga0[2][g.nz], n1 = g.ga1[2][g.nz];
        const double2 an = ldp(pn), bn = ldp(pn + g.sxy);
        sub(o.V[2], cell + (int64_t)g.nx * g.ny, make_double2(n0 * an.x + n1 * bn.x, n0 * an.y + n1 * bn.y));
      }
    }
  }
}

// Round 4: the same update when all six arrays are there (PCApply_ABF's call) as straight-line code -- every load of a row (the six arrays, 25 values of
// p) is issued before the first use, the arrays with the non-temporal hint (they are touched once; p is read seven times and should stay cached) -- and
// with the work divided so that p stays in an XCD's L2: XCD x (block number mod 8) owns the rows j in [x ny / 8, (x + 1) ny / 8) of every plane and
// walks them plane by plane, so the three planes of p a row needs are three 1/8 slabs.  DIRECT: p is the caller's unpadded array (single rank: no ghost
// layer is needed -- a tap outside the block exists on periodic axes only, and is wrapped here), which saves the padded copy (16 B per cell) in front of
// the kernel.  The products and their order are those of k_project_cells / k_project_faces.  U rows per wave and pass.
template <bool DIRECT, int NT, int U>
__global__ void __launch_bounds__(256) k_project_six(GridP g, const double *__restrict__ p, ProjOut o, int per, int nxcd)
{
  const int lane = threadIdx.x & 63, w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int nseg = (g.nx + 127) / 128;
  const int xcd = (int)blockIdx.x % nxcd, lw = ((int)blockIdx.x / nxcd) * 4 + w, WX = ((int)gridDim.x / nxcd) * 4;  // WX is a multiple of nseg (launcher)
  const int j0 = (int)((int64_t)xcd * g.ny / nxcd), sn = (int)((int64_t)(xcd + 1) * g.ny / nxcd) - j0;
  const int seg = lw % nseg, i = seg * 128 + 2 * lane;  // nx is even (launcher): a lane owns cells i, i + 1
  if (i >= g.nx || sn <= 0) return;
  const double kap = g.kappa;
  const int     sy = DIRECT ? g.nx : g.sx;
  const int64_t sz = DIRECT ? (int64_t)g.nx * g.ny : g.sxy;
  auto fix = [](int s, int n, bool periodic) { return s < 0 ? (periodic ? s + n : 0) : (s >= n ? (periodic ? s - n : n - 1) : s); };
  // x rows of the lane's cells / their low faces (and of face nx behind the last cell): offsets of the taps from cell 0 of the row
  int    xv[2][3], xf[2][2], xn[2] = {0, 0};
  double xg0[2], xg1[2], xg2[2], xa0[2], xa1[2];
#pragma unroll
  for (int c = 0; c < 2; ++c) {
    const int s0 = g.Gs[0][i + c], c0 = g.gc0[0][i + c];
    xg0[c] = g.Gv0[0][i + c]; xg1[c] = g.Gv1[0][i + c]; xg2[c] = g.Gv2[0][i + c];
    xa0[c] = g.ga0[0][i + c]; xa1[c] = g.ga1[0][i + c];
#pragma unroll
    for (int m = 0; m < 3; ++m) xv[c][m] = DIRECT ? fix(s0 + m, g.nx, per & 1) : s0 + m;
#pragma unroll
    for (int m = 0; m < 2; ++m) xf[c][m] = DIRECT ? fix(c0 + m, g.nx, per & 1) : c0 + m;
  }
  const bool   xlast = i + 1 == g.nx - 1 && g.fx > g.nx;
  const double xa0n = xlast ? g.ga0[0][g.nx] : 0., xa1n = xlast ? g.ga1[0][g.nx] : 0.;
  if (xlast) {
    const int c0 = g.gc0[0][g.nx];
    xn[0] = DIRECT ? fix(c0, g.nx, per & 1) : c0;
    xn[1] = DIRECT ? fix(c0 + 1, g.nx, per & 1) : c0 + 1;
  }
  struct Row {
    int     j, k;
    int64_t pr, cell, fxi, fyi;
    double2 a0, a1, a2, b1, b2, py[3], pyf[2], pz[3], pzf[2];
    double  bx0, bx1, px[2][3], pf[2][2];
  };
  auto ldp = [&](int64_t q) { return *reinterpret_cast<const double2 *>(p + q); };
  auto ldn = [&](const double *a) { return NT ? __builtin_nontemporal_load(a) : *a; };
  auto stn = [&](double *a, double v) { if (NT) __builtin_nontemporal_store(v, a); else *a = v; };
  auto load = [&](Row &R, uint32_t rowi) {
    R.j = j0 + (int)(rowi % (uint32_t)sn);
    R.k = (int)(rowi / (uint32_t)sn);
    const int j = R.j, k = R.k;
    R.pr   = DIRECT ? ((int64_t)k * g.ny + j) * g.nx : g.off0 + (int64_t)k * g.sxy + (int64_t)j * g.sx;
    R.cell = ((int64_t)k * g.ny + j) * g.nx + i;
    R.fxi  = ((int64_t)k * g.ny + j) * g.fx + i;
    R.fyi  = ((int64_t)k * g.fy + j) * g.nx + i;
    R.a0 = ld2<NT>(o.v[0] + R.cell); R.a1 = ld2<NT>(o.v[1] + R.cell); R.a2 = ld2<NT>(o.v[2] + R.cell);
    R.b1 = ld2<NT>(o.V[1] + R.fyi);  R.b2 = ld2<NT>(o.V[2] + R.cell);
    R.bx0 = ldn(o.V[0] + R.fxi);     R.bx1 = ldn(o.V[0] + R.fxi + 1);
#pragma unroll
    for (int c = 0; c < 2; ++c) {
#pragma unroll
      for (int m = 0; m < 3; ++m) R.px[c][m] = p[R.pr + xv[c][m]];
#pragma unroll
      for (int m = 0; m < 2; ++m) R.pf[c][m] = p[R.pr + xf[c][m]];
    }
    const int ys = g.Gs[1][j], yc = g.gc0[1][j], zs = g.Gs[2][k], zc = g.gc0[2][k];
#pragma unroll
    for (int m = 0; m < 3; ++m) {
      R.py[m] = ldp(R.pr + i + (int64_t)((DIRECT ? fix(ys + m, g.ny, per & 2) : ys + m) - j) * sy);
      R.pz[m] = ldp(R.pr + i + (int64_t)((DIRECT ? fix(zs + m, g.nz, per & 4) : zs + m) - k) * sz);
    }
#pragma unroll
    for (int m = 0; m < 2; ++m) {
      R.pyf[m] = ldp(R.pr + i + (int64_t)((DIRECT ? fix(yc + m, g.ny, per & 2) : yc + m) - j) * sy);
      R.pzf[m] = ldp(R.pr + i + (int64_t)((DIRECT ? fix(zc + m, g.nz, per & 4) : zc + m) - k) * sz);
    }
  };
  auto cellrow = [&](double c0, double c1, double c2, const double2 *t) {
    double2 gr = make_double2(c0 * t[0].x + c1 * t[1].x, c0 * t[0].y + c1 * t[1].y);
    if (c2 != 0.) {
      gr.x += c2 * t[2].x;
      gr.y += c2 * t[2].y;
    }
    return gr;
  };
  auto sub2 = [&](double2 t, double2 d) {
    t.x -= kap * d.x;
    t.y -= kap * d.y;
    return t;
  };
  auto finish = [&](Row &R) {
    const int j = R.j, k = R.k;
    // ---- x
    double2 gr;
    gr.x = xg0[0] * R.px[0][0] + xg1[0] * R.px[0][1];
    if (xg2[0] != 0.) gr.x += xg2[0] * R.px[0][2];
    gr.y = xg0[1] * R.px[1][0] + xg1[1] * R.px[1][1];
    if (xg2[1] != 0.) gr.y += xg2[1] * R.px[1][2];
    st2<NT>(o.v[0] + R.cell, sub2(R.a0, gr));
    stn(o.V[0] + R.fxi, R.bx0 - kap * (xa0[0] * R.pf[0][0] + xa1[0] * R.pf[0][1]));
    stn(o.V[0] + R.fxi + 1, R.bx1 - kap * (xa0[1] * R.pf[1][0] + xa1[1] * R.pf[1][1]));
    if (xlast) o.V[0][R.fxi + 2] -= kap * (xa0n * p[R.pr + xn[0]] + xa1n * p[R.pr + xn[1]]);
    // ---- y, z (wave-uniform rows)
    st2<NT>(o.v[1] + R.cell, sub2(R.a1, cellrow(g.Gv0[1][j], g.Gv1[1][j], g.Gv2[1][j], R.py)));
    st2<NT>(o.v[2] + R.cell, sub2(R.a2, cellrow(g.Gv0[2][k], g.Gv1[2][k], g.Gv2[2][k], R.pz)));
    {
      const double a0 = g.ga0[1][j], a1 = g.ga1[1][j];
      st2<NT>(o.V[1] + R.fyi, sub2(R.b1, make_double2(a0 * R.pyf[0].x + a1 * R.pyf[1].x, a0 * R.pyf[0].y + a1 * R.pyf[1].y)));
    }
    {
      const double a0 = g.ga0[2][k], a1 = g.ga1[2][k];
      st2<NT>(o.V[2] + R.cell, sub2(R.b2, make_double2(a0 * R.pzf[0].x + a1 * R.pzf[1].x, a0 * R.pzf[0].y + a1 * R.pzf[1].y)));
    }
    if (j == g.ny - 1 && g.fy > g.ny) {  // the face behind the last row / plane, where this rank owns it
      const int     c0n = g.gc0[1][g.ny];
      const double  n0 = g.ga0[1][g.ny], n1 = g.ga1[1][g.ny];
      const double2 an = ldp(R.pr + i + (int64_t)((DIRECT ? fix(c0n, g.ny, per & 2) : c0n) - j) * sy), bn = ldp(R.pr + i + (int64_t)((DIRECT ? fix(c0n + 1, g.ny, per & 2) : c0n + 1) - j) * sy);
      double2      *q = reinterpret_cast<double2 *>(o.V[1] + R.fyi + g.nx);
      *q = sub2(*q, make_double2(n0 * an.x + n1 * bn.x, n0 * an.y + n1 * bn.y));
    }
    if (k == g.nz - 1 && g.fz > g.nz) {
      const int     c0n = g.gc0[2][g.nz];
      const double  n0 = g.ga0[2][g.nz], n1 = g.ga1[2][g.nz];
      const double2 an = ldp(R.pr + i + (int64_t)((DIRECT ? fix(c0n, g.nz, per & 4) : c0n) - k) * sz), bn = ldp(R.pr + i + (int64_t)((DIRECT ? fix(c0n + 1, g.nz, per & 4) : c0n + 1) - k) * sz);
      double2      *q = reinterpret_cast<double2 *>(o.V[2] + R.cell + (int64_t)g.nx * g.ny);
      *q = sub2(*q, make_double2(n0 * an.x + n1 * bn.x, n0 * an.y + n1 * bn.y));
    }
  };
  const uint32_t nrows = (uint32_t)sn * (uint32_t)g.nz, step = (uint32_t)(WX / nseg);
  for (uint32_t rowi = (uint32_t)(lw / nseg); rowi < nrows; rowi += U * step) {
    Row R[U];
#pragma unroll
    for (int u = 0; u < U; ++u)
      if (u == 0 || rowi + u * step < nrows) load(R[u], rowi + u * step);
#pragma unroll
    for (int u = 0; u < U; ++u)
      if (u == 0 || rowi + u * step < nrows) finish(R[u]);
  }
}

// boundary face plane of V (axis, side) = coeff * pb     (INSERT_VALUES)
__global__ void k_gst_bc(GridP g, const double *__restrict__ pb, double *__restrict__ V, int axis, int side, double coeff, int add)
{
  const int a = blockIdx.x * 64 + threadIdx.x, b = blockIdx.y * 4 + threadIdx.y;
  int       na, nb;
  if (axis == 0) { na = g.ny; nb = g.nz; }
  else if (axis == 1) { na = g.nx; nb = g.nz; }
  else { na = g.nx; nb = g.ny; }
  if (a >= na || b >= nb) return;
  const int f = side ? (axis == 0 ? g.fx : (axis == 1 ? g.fy : g.fz)) - 1 : 0;
  int64_t   p = axis == 0 ? ((int64_t)b * g.ny + a) * g.fx + f : (axis == 1 ? ((int64_t)b * g.fy + f) * g.nx + a : ((int64_t)f * g.ny + b) * g.nx + a);
  const double t = coeff * pb[(int64_t)b * na + a];
  V[p]           = add ? V[p] + t : t;
}

// cell layer next to boundary (axis, side) += coeff * plane   (ADD_VALUES of the boundary-condition vectors)
__global__ void k_bc_add_cells(GridP g, const double *__restrict__ plane, double *__restrict__ cells, int axis, int side, double coeff)
{
  const int a = blockIdx.x * 64 + threadIdx.x, b = blockIdx.y * 4 + threadIdx.y;
  int       na, nb;
  if (axis == 0) { na = g.ny; nb = g.nz; }
  else if (axis == 1) { na = g.nx; nb = g.nz; }
  else { na = g.nx; nb = g.ny; }
  if (a >= na || b >= nb) return;
  const int c = side ? (axis == 0 ? g.nx : (axis == 1 ? g.ny : g.nz)) - 1 : 0;
  const int64_t p = axis == 0 ? ((int64_t)b * g.ny + a) * g.nx + c : (axis == 1 ? ((int64_t)b * g.ny + c) * g.nx + a : ((int64_t)c * g.ny + b) * g.nx + a);
  cells[p] += coeff * plane[(int64_t)b * na + a];
}

// cnlinearcart3d.c:2846-2854
__global__ void k_pressure_update(int64_t n, int first, const double *__restrict__ dp, const double *__restrict__ p0, double *__restrict__ phalf, double *__restrict__ p)
{
  for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < n; t += (int64_t)gridDim.x * blockDim.x) {
    const double d = dp[t];
    if (first) {
      const double q = p0[t];
      p[t]           = 2. * d + q;  // VecWAXPY(solp, 2., dp, p0)
      phalf[t]       = d + q;       // VecWAXPY(phalf, 1., dp, p0)
    } else {
      const double h = phalf[t];
      p[t]           = 1.5 * d + h;  // VecWAXPY(solp, 1.5, dp, phalf)
      phalf[t]       = h + d;        // VecAXPY(phalf, 1., dp)
    }
  }
}

// ------------------------------------------------------------------------------------------------ reductions / scalar state

// sums[a] = sum_b partial[a*stride + b], a < nslot.  One block of 256 threads, fixed order.
__device__ __forceinline__ void reduce_partials(const double *__restrict__ partial, int nblocks, int stride, int nslot, double *out /* shared [NSLOT] */, double *red /* shared [NSLOT*4] */)
{
  double v[NSLOT];
#pragma unroll
  for (int a = 0; a < NSLOT; ++a) {
    v[a] = 0.;
    if (a < nslot)
      for (int b = threadIdx.x; b < nblocks; b += 256) v[a] += partial[(int64_t)a * stride + b];
  }
  block_sum<NSLOT>(v, red);
  if (threadIdx.x == 0)
    for (int a = 0; a < NSLOT; ++a) out[a] = v[a];
  __syncthreads();
}

// partials -> sums (for the multi-rank path: an all-reduce of `sums` follows)
__global__ void __launch_bounds__(256) k_reduce(const double *__restrict__ partial, int nblocks, int stride, int nslot, double *__restrict__ sums)
{
  __shared__ double out[NSLOT], red[NSLOT * 4];
  reduce_partials(partial, nblocks, stride, nslot, out, red);
  if (threadIdx.x < NSLOT) sums[threadIdx.x] = (int)threadIdx.x < nslot ? out[threadIdx.x] : 0.;
}

__device__ __forceinline__ int converged_default(KspScal *s, double dp)
{
  // KSPConvergedDefault, zero initial guess
  if (isnan(dp) || isinf(dp)) return FL_DIVERGED_NANORINF;
  if (dp <= s->ttol) return dp < s->atol ? FL_CONVERGED_ATOL : FL_CONVERGED_RTOL;
  if (dp >= s->dtol * s->rnorm0) return FL_DIVERGED_DTOL;
  return 0;
}

// slots: 0 sum r*z0, 1 sum z0^2, 2 sum z0, 3 sum r, 4 sum r^2      (z0 = r/diag before the mean removal)
__device__ __forceinline__ double cg_norms(KspScal *s, const double *sum, double &rz)
{
  const double N    = s->ncell_global;
  const double mean = s->nullspace ? sum[2] / N : 0.;
  rz                = sum[0] - mean * sum[3];
  const double zz   = sum[1] - N * mean * mean;
  s->zshift         = mean;
  switch (s->norm_type) {
  case FL_NORM_PRECONDITIONED: return sqrt(zz < 0. ? 0. : zz);  // keeps a NaN a NaN
  case FL_NORM_UNPRECONDITIONED: return sqrt(sum[4]);
  case FL_NORM_NATURAL: return sqrt(fabs(rz));
  default: return 0.;
  }
}

// mode 0: after k_cg_init.  mode 1: after k_cg_A (alpha).  mode 2: after k_cg_B / k_cg_Bq (beta, convergence).  One thread.
// Who owes x what (pending_x = 1: x still lacks alpha * p of the current direction; k_cg_finish adds it):
//   stored-q pair: k_cg_A applies the update deferred from the iteration before (mode 1 clears the flag), k_cg_B leaves one (mode 2);
//   q-free pair:   k_cg_A never touches x (mode 3 = mode 1 without clearing); k_cg_Bq applies none on even iterations (mode 2) and
//                  the two it then owes on odd ones (mode 4 = mode 2 with nothing left pending).
__device__ __forceinline__ void cg_fin_apply(int mode, const double *out, KspScal *__restrict__ s, double *__restrict__ hist, int nhist)
{
  if (mode == 1 || mode == 3) {
    if (mode == 1) s->pending_x = 0;
    const double pq = out[0];
    s->pq           = pq;
    if (!(pq > 0.)) {
      // the direction just written is never used: cur keeps pointing at the one a pending x-update refers to
      s->reason = isnan(pq) ? FL_DIVERGED_NANORINF : FL_DIVERGED_INDEFINITE_MAT;
      return;
    }
    s->cur ^= 1;
    s->alpha_old = s->alpha;
    s->alpha     = s->rz / pq;
    return;
  }
  double       rz;
  const double dp = cg_norms(s, out, rz);
  if (mode == 0) {
    s->rz     = rz;
    s->rz_old = 1.;
    s->beta   = 0.;
    s->it     = 0;
    s->rnorm0 = dp;
    s->ttol   = fmax(s->rtol * dp, s->atol);
  } else {
    s->rz_old    = s->rz;
    s->rz        = rz;
    s->beta      = rz / s->rz_old;
    s->it += 1;
    s->pending_x = mode == 4 ? 0 : 1;
    if (mode == 4) s->x_valid = 1;
  }
  s->dp = dp;
  if (hist && s->it < nhist) hist[s->it] = dp;
  int reason = converged_default(s, dp);
  if (!reason) {
    if (s->it >= s->maxit) reason = FL_DIVERGED_ITS;
    else if (rz < 0.) reason = FL_DIVERGED_INDEFINITE_PC;
  }
  s->reason = reason;
}

// If nblocks > 0 the partials are reduced here (single rank, unfused path); else `sums` already holds the (all-reduced) sums.
__global__ void __launch_bounds__(256) k_cg_fin(int mode, const double *__restrict__ partial, int nblocks, int stride, const double *__restrict__ sums, KspScal *__restrict__ s, double *__restrict__ hist, int nhist)
{
  __shared__ double out[NSLOT], red[NSLOT * 4];
  if (s->reason != 0) return;
  if (nblocks > 0) reduce_partials(partial, nblocks, stride, (mode == 1 || mode == 3) ? 1 : 5, out, red);
  else {
    if (threadIdx.x < NSLOT) out[threadIdx.x] = sums[threadIdx.x];
    __syncthreads();
  }
  if (threadIdx.x == 0) cg_fin_apply(mode, out, s, hist, nhist);
}

// The same scalar update done by the LAST block of the producing kernel (single rank): saves two launches per CG
// iteration.  Hand-off = MI355X_MICROARCH "valid forms": every block publishes its partial sums with agent-scope (sc1,
// write-through) stores, drains them (s_waitcnt vmcnt(0)), then draws a ticket from a device-scope counter; the block
// that draws the last ticket reads every partial with agent-scope (sc1, L1-bypassing) loads.  The sum order is fixed
// (thread t takes blocks t, t+T, ...; then a fixed tree), so the result does not depend on which block is last.
struct FinCtx {
  unsigned *counter;  // zero between launches (the last block resets it)
  double   *hist;
  int       nhist, enabled;
  double   *sums;     // several ranks: the last block only leaves the rank's sums here (all-reduce + k_cg_fin follow)
};
template <int NV, int NTHR>
__device__ __forceinline__ void fused_fin(int mode, const double (&v)[NV] /* thread 0 */, double *__restrict__ partial, int stride, const FinCtx &f, KspScal *__restrict__ s, double *red /* shared [NV * NTHR/64] */, int *flag /* shared */)
{
  const int nblocks = gridDim.x;
  if (threadIdx.x == 0) {
#pragma unroll
    for (int a = 0; a < NV; ++a) __hip_atomic_store(&partial[(int64_t)a * stride + blockIdx.x], v[a], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const unsigned t = __hip_atomic_fetch_add(f.counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    *flag            = (t == (unsigned)(nblocks - 1));
  }
  __syncthreads();
  if (!*flag) return;
  double w[NV];
#pragma unroll
  for (int a = 0; a < NV; ++a) {
    w[a] = 0.;
    for (int b = threadIdx.x; b < nblocks; b += NTHR) w[a] += __hip_atomic_load(&partial[(int64_t)a * stride + b], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  __syncthreads();  // red may still hold the block's own reduction
#pragma unroll
  for (int a = 0; a < NV; ++a) {
    w[a] = wave_sum(w[a]);
    if (lane == 0) red[a * (NTHR / 64) + wv] = w[a];
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    double out[NSLOT];
#pragma unroll
    for (int a = 0; a < NSLOT; ++a) {
      out[a] = 0.;
      if (a < NV)
        for (int q = 0; q < NTHR / 64; ++q) out[a] += red[a * (NTHR / 64) + q];
    }
    if (f.sums) {
#pragma unroll
      for (int a = 0; a < NSLOT; ++a) f.sums[a] = out[a];
    } else cg_fin_apply(mode, out, s, f.hist, f.nhist);
    __hip_atomic_store(f.counter, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
}

// ------------------------------------------------------------------------------------------------ CG: init / B / flush

// work split of the streaming kernels: one wave per 128-cell row segment, grid-stride over segments
struct SegIter {
  int64_t seg, nseg, stride;
  int     nxs;
};

// r_pad = b ; five partial sums.  b unpadded.
template <bool JAC>
__global__ void __launch_bounds__(256) k_cg_init(GridP g, const double *__restrict__ b, double *__restrict__ r, double *__restrict__ x0, double *__restrict__ partial, int stride, int pairs)
{
  __shared__ double red[5 * 4];
  const int         lane = threadIdx.x & 63;
  const int         nxs  = (g.nx + 127) / 128;
  const int64_t     nseg = (int64_t)nxs * g.ny * g.nz;
  double            acc[5] = {0., 0., 0., 0., 0.};
  for (int64_t seg = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6); seg < nseg; seg += (int64_t)gridDim.x * 4) {
    const int     xs = (int)(seg % nxs);
    const int64_t R  = seg / nxs;
    const int     j = (int)(R % g.ny), k = (int)(R / g.ny);
    const double  dyz = g.sc[1][j] + g.sc[2][k];
    const int     i = xs * 128 + 2 * lane;
    double        rv[2] = {0., 0.};
    const int64_t ob = ((int64_t)k * g.ny + j) * g.nx + i, op = pidx(g, i, j, k);
    if (pairs && i + 1 < g.nx) {  // even nx, 16-byte aligned b: one 16-byte access per stream
      const double2 v = *reinterpret_cast<const double2 *>(b + ob);
      rv[0] = v.x;
      rv[1] = v.y;
      *reinterpret_cast<double2 *>(r + op) = v;
      if (x0) *reinterpret_cast<double2 *>(x0 + op) = make_double2(0., 0.);
    } else {
#pragma unroll
      for (int c = 0; c < 2; ++c)
        if (i + c < g.nx) {
          rv[c]      = b[ob + c];
          r[op + c]  = rv[c];
          if (x0) x0[op + c] = 0.;
        }
    }
#pragma unroll
    for (int c = 0; c < 2; ++c)
      if (i + c < g.nx) {
        const double z0 = JAC ? rv[c] / (g.sc[0][i + c] + dyz) : rv[c];
        acc[0] += rv[c] * z0;
        acc[1] += z0 * z0;
        acc[2] += z0;
        acc[3] += rv[c];
        acc[4] += rv[c] * rv[c];
      }
  }
  block_sum<5>(acc, red);
  if (threadIdx.x == 0)
#pragma unroll
    for (int a = 0; a < 5; ++a) partial[(int64_t)a * stride + blockIdx.x] = acc[a];
}

// r -= alpha q ; partial sums of the new r.  24 B/cell.  Same 128 x (4*RY) x zc tiling as k_cg_A (no integer division in
// the loop); loads are unconditional on clamped, always-valid addresses so that the compiler can count them (a load
// inside a divergent branch costs an s_waitcnt vmcnt(0)); only the stores and the sums are masked.
#ifdef FL_KBENCH_VARIANTS  // variant 2 of the CG pair: the r-update that reads a stored q
template <int RY, bool JAC, int NT>
__global__ void __launch_bounds__(256) k_cg_B(GridP g, const double *__restrict__ q, double *__restrict__ r, KspScal *__restrict__ s, double *__restrict__ partial, int stride, int nchunk, int zc, int tiles_x, FinCtx fin)
{
  __shared__ double red[5 * 4];
  __shared__ int    flag;
  if (s->reason != 0) return;
  const double alpha = s->alpha;
  const int    b = blockIdx.x, chunk = b % nchunk, tile = b / nchunk;
  const int    i0 = (tile % tiles_x) * 128, j0 = (tile / tiles_x) * (4 * RY);
  const int    k0 = chunk * zc, k1 = min(k0 + zc, g.nz);
  const int    lane = threadIdx.x & 63, w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int    i = i0 + 2 * lane, il = min(i, g.nx & ~1);
  const bool   own0 = i < g.nx, own1 = i + 1 < g.nx;
  const double xc0 = g.sc[0][min(i, g.nx)], xc1 = g.sc[0][min(i + 1, g.nx)];
  int64_t      ro[RY];
  bool         rown[RY];
  double       yc[RY];
#pragma unroll
  for (int m = 0; m < RY; ++m) {
    const int j = j0 + w * RY + m;
    rown[m]     = j < g.ny;
    ro[m]       = g.off0 + (int64_t)min(j, g.ny) * g.sx;  // wave-uniform; the lane adds il
    yc[m]       = g.sc[1][min(j, g.ny)];
  }
  double acc[5] = {0., 0., 0., 0., 0.};
  struct Raw {
    double2 q[RY], r[RY];
    double  zc;
  };
  // loads of plane k (clamped into the chunk: the trip past the end re-reads a cached plane instead of branching)
  auto load = [&](int k, Raw &R) {
    const int     kc = min(k, k1 - 1);
    const int64_t pl = (int64_t)kc * g.sxy;
#pragma unroll
    for (int m = 0; m < RY; ++m) {
      R.q[m] = ld2<NT>(q + ro[m] + pl + il);
      R.r[m] = ld2<NT>(r + ro[m] + pl + il);
    }
    R.zc = g.sc[2][kc];
  };
  // one plane: prefetch k+1 into N, work on C.  Called with (A,B) then (B,A): the two register sets ping-pong, so no
  // register copy ever has to wait for a load in flight.
  auto step = [&](int k, const Raw &C, Raw &N) {
    load(k + 1, N);
    const int64_t pc = (int64_t)k * g.sxy;
#pragma unroll
    for (int m = 0; m < RY; ++m) {
      double2 rn;
      rn.x = fma(-alpha, C.q[m].x, C.r[m].x);  // the same rounding as k_pack_faces_rq
      rn.y = fma(-alpha, C.q[m].y, C.r[m].y);
      const double dyz = yc[m] + C.zc;
      const double z0 = JAC ? rn.x / (xc0 + dyz) : rn.x;
      const double z1 = JAC ? rn.y / (xc1 + dyz) : rn.y;
      if (rown[m]) {
        if (own1) st2<NT>(r + ro[m] + pc + il, rn);
        else if (own0) r[ro[m] + pc + il] = rn.x;
      }
      const double m0 = (rown[m] && own0) ? 1. : 0., m1 = (rown[m] && own1) ? 1. : 0.;
      const double r0 = m0 * rn.x, r1 = m1 * rn.y, zz0 = m0 * z0, zz1 = m1 * z1;
      acc[0] += r0 * zz0 + r1 * zz1;
      acc[1] += zz0 * zz0 + zz1 * zz1;
      acc[2] += zz0 + zz1;
      acc[3] += r0 + r1;
      acc[4] += r0 * r0 + r1 * r1;
    }
  };
  if (k0 < k1) {
    Raw A, B;
    load(k0, A);
    for (int k = k0; k < k1; k += 2) {
      step(k, A, B);
      if (k + 1 < k1) step(k + 1, B, A);
    }
  }
  block_sum<5>(acc, red);
  if (fin.enabled) fused_fin<5, 256>(2, acc, partial, stride, fin, s, red, &flag);
  else if (threadIdx.x == 0)
#pragma unroll
    for (int a = 0; a < 5; ++a) partial[(int64_t)a * stride + blockIdx.x] = acc[a];
}
#endif  // FL_KBENCH_VARIANTS

// the x-update still owed when the iteration stops, fused with the copy into the caller's (unpadded) array:
//   xout = x + alpha p   (p = the current direction; plain copy when nothing is pending).  The padded x is not updated: the
// next solve starts from its own zeroed copy.
__global__ void __launch_bounds__(256) k_cg_finish(GridP g, const double *__restrict__ P0, const double *__restrict__ P1, const double *__restrict__ x, double *__restrict__ xout, const KspScal *__restrict__ s, int pairs)
{
  const double *p     = s->cur ? P1 : P0;
  const double  alpha = s->pending_x ? s->alpha : 0.;
  const bool    upd   = s->pending_x != 0;
  const bool    xv    = s->x_valid != 0;  // false: x has never been written in this solve (it stands for 0)
  const int     lane  = threadIdx.x & 63;
  const int     nxs   = (g.nx + 127) / 128;
  const int64_t nseg  = (int64_t)nxs * g.ny * g.nz;
  for (int64_t seg = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6); seg < nseg; seg += (int64_t)gridDim.x * 4) {
    const int     xs = (int)(seg % nxs);
    const int64_t R  = seg / nxs;
    const int     j = (int)(R % g.ny), k = (int)(R / g.ny);
    const int     i = xs * 128 + 2 * lane;
    const int64_t ob = ((int64_t)k * g.ny + j) * g.nx + i, op = pidx(g, i, j, k);
    if (pairs && i + 1 < g.nx) {
      double2 v = xv ? *reinterpret_cast<const double2 *>(x + op) : make_double2(0., 0.);
      if (upd) {
        const double2 pv = *reinterpret_cast<const double2 *>(p + op);
        v.x += alpha * pv.x;
        v.y += alpha * pv.y;
      }
      *reinterpret_cast<double2 *>(xout + ob) = v;
    } else {
#pragma unroll
      for (int c = 0; c < 2; ++c)
        if (i + c < g.nx) {
          const double xv0 = xv ? x[op + c] : 0.;
          xout[ob + c]     = upd ? xv0 + alpha * p[op + c] : xv0;
        }
    }
  }
}

// ------------------------------------------------------------------------------------------------ CG: the fused stencil kernel

// RY rows per wave, NW waves per block (tile 128 x NW*RY), PF prefetch mode, NT: 0 plain, 1 non-temporal stores,
// 2 non-temporal stores and tile loads (halo loads stay plain: they are meant to hit in L2)
// SQ: store q (k_cg_B reads it back) and apply the x-update deferred from the iteration before; !SQ: neither -- q is formed again and
// x is updated by k_cg_Bq, so this kernel reads r, p and writes p' (24 B/cell).
template <int RY, int NW, bool JAC, int PF, int NT, bool SQ>
__device__ __forceinline__ void cg_A_body(const GridP &g, const double *__restrict__ r, double *__restrict__ P0, double *__restrict__ P1, double *__restrict__ q, double *__restrict__ x, KspScal *__restrict__ s,
                                                      double *__restrict__ partial, int nchunk, int zc, int tiles_x, int tiles, int remap, FinCtx fin)
{
  using T               = TileA<RY, NW>;
  constexpr int TX = T::TX, TY = T::TY, LX = T::LX, LY = T::LY;
  constexpr int NTL = NT >= 2, NTS = NT >= 1;
  // three staged p' planes: kk (being written), kk-1 (stencil centre + in-plane neighbours), kk-2 (z-low neighbour).
  // Keeping the two older planes in LDS instead of registers frees 32 VGPRs for the load prefetch.
  __shared__ __attribute__((aligned(16))) double lds[3][LY][LX];
  __shared__ double                              red[NW];
  __shared__ int                                 flag;
  if (s->reason != 0) return;

  const int     cur        = s->cur;
  const double *pold       = cur ? P1 : P0;
  double       *pnew       = cur ? P0 : P1;
  const double  beta       = s->beta;
  const double  zs         = s->zshift;
  const double  alpha_prev = s->alpha;  // 0 on the first iteration (and p_old = 0): the deferred x-update is a no-op then

  const bool qbnd = !SQ && (remap & 2) != 0;  // store q on the block's boundary layers only (PlanA::qb)
  const int  b    = (remap & 1) ? xcd_remap(blockIdx.x, gridDim.x) : blockIdx.x;
  const int chunk = b / tiles, tile = b % tiles;  // chunk-major: consecutive logical blocks are neighbouring tiles
  const int i0 = (tile % tiles_x) * TX, j0 = (tile / tiles_x) * TY;
  const int k0 = chunk * zc, k1 = min(k0 + zc, g.nz);  // plan_tiles guarantees k0 < k1 for every block
  const int tid = threadIdx.x, lane = tid & 63;
  const int w  = __builtin_amdgcn_readfirstlane(tid >> 6);  // wave-uniform: rows, row offsets and y-coefficients live in SGPRs
  const int i  = i0 + 2 * lane;  // first of this lane's two cells
  const int jb = j0 + w * RY;    // first of this wave's RY rows

  // Every load below is unconditional and goes to a clamped, always-valid address: a load inside a divergent branch
  // makes hipcc wait vmcnt(0) at its use, which would also drain the next plane's prefetch.  Masks apply to stores only.
  const bool own0 = i < g.nx, own1 = i + 1 < g.nx;
  const bool gh0 = i == g.nx, gh1 = i + 1 == g.nx;  // high ghost column inside the tile
  const int  il  = min(i, g.nx & ~1);               // clamped pair start (16-B aligned, inside the padded row)
  const int  ic0 = min(i, g.nx), ic1 = min(i + 1, g.nx);
  const double xl0 = g.sl[0][ic0], xc0 = g.sc[0][ic0], xh0 = g.sh[0][ic0];
  const double xl1 = g.sl[0][ic1], xc1 = g.sc[0][ic1], xh1 = g.sh[0][ic1];
  int64_t rob[RY];  // wave-uniform offset of (0, clamped row, plane 0); the lane adds il
  bool    rown[RY], rgh[RY];
  double  yl[RY], yc[RY], yh[RY];
#pragma unroll
  for (int m = 0; m < RY; ++m) {
    const int j  = jb + m, jc = min(j, g.ny);
    rown[m]      = j < g.ny;
    rgh[m]       = j == g.ny;
    rob[m]       = g.off0 + (int64_t)jc * g.sx;
    yl[m]        = g.sl[1][jc];
    yc[m]        = g.sc[1][jc];
    yh[m]        = g.sh[1][jc];
  }
#define RO(m) (rob[m] + il)

  // halo cells of this thread -----------------------------------------------------------------------------------
  // A: rows jj = -1 (tid < 128) / TY (128 <= tid < 256), column ii = tid & 127
  // B: columns ii = -1 / TX, jj = (tid - HB0) >> 1 for HB0 <= tid < HB0 + 2*TY; a 512-thread block gives A and B to
  //    different threads (HB0 = 256), a 256-thread block gives some threads one of each
  constexpr int HB0 = NW > 4 ? 256 : 0;
  const int  tb  = tid - HB0;
  const int  hAi = i0 + (tid & 127), hAj = j0 + (tid < 128 ? -1 : TY);
  const bool hAok = tid < 256 && hAi < g.nx && hAj <= g.ny;
  const bool hAgh = hAok && (hAj == -1 || hAj == g.ny);
  const int  hBi = i0 + ((tb & 1) ? TX : -1), hBj = j0 + (tb >> 1);
  const bool hBok = tb >= 0 && tb < 2 * TY && hBj < g.ny && hBi <= g.nx;
  const bool hBgh = hBok && (hBi == -1 || hBi == g.nx);
  const int64_t tbase = g.off0 + (int64_t)j0 * g.sx + i0;  // uniform; halo cells are 32-bit offsets from it
  const int     hAo   = hAok ? (hAj - j0) * g.sx + (hAi - i0) : 0;
  const int     hBo   = hBok ? (hBj - j0) * g.sx + (hBi - i0) : 0;
  const double  hAdxy = hAok ? g.sc[0][hAi] + g.sc[1][hAj] : 1.;
  const double  hBdxy = hBok ? g.sc[0][hBi] + g.sc[1][hBj] : 1.;
  const int hAr = hAok ? (tid < 128 ? 0 : TY + 1) : 0, hAc = hAok ? (tid & 127) + 2 : 0;  // (0,0) is a dead corner slot
  const int hBr = hBok ? (tb >> 1) + 1 : 0, hBc = hBok ? ((tb & 1) ? TX + 2 : 1) : 0;

  double2 pnext[RY];
  double  dot = 0.;
  double  zlc = 0., zcc = 0., zhc = 0.;  // z-row of plane kk-1 (the plane whose q is formed)

  // raw inputs of one plane, fetched one trip ahead
  struct Raw {
    double2 r[RY], p[RY], x[RY];
    double  hrA, hpA, hrB, hpB;
    double  zl, zc, zh;  // z-row of that plane: prefetched with it so that no load sits between prefetch and use
  };
  auto load = [&](int kn_, Raw &R) {
    // clamped to k1 (the trip past the end re-reads a cached plane instead of branching)
    const int     kn = min(kn_, k1);
    const int64_t pl = (int64_t)kn * g.sxy;
#pragma unroll
    for (int m = 0; m < RY; ++m) {
      R.r[m] = ld2<NTL>(r + RO(m) + pl);
      R.p[m] = ld2<NTL>(pold + RO(m) + pl);
    }
    if (SQ) {
      // x only exists on owned planes: the two extra trips re-read a cached plane instead of branching
      const int64_t px = (int64_t)min(max(kn_, k0), k1 - 1) * g.sxy;
#pragma unroll
      for (int m = 0; m < RY; ++m) R.x[m] = ld2<NTL>(x + RO(m) + px);
    }
    R.hrA = r[tbase + pl + hAo];
    R.hpA = pold[tbase + pl + hAo];
    R.hrB = r[tbase + pl + hBo];
    R.hpB = pold[tbase + pl + hBo];
    R.zl  = g.sl[2][kn];
    R.zc  = g.sc[2][kn];
    R.zh  = g.sh[2][kn];
  };

  // one plane: prefetch kk+1 into N, work on C.  Called alternately with (A,B) and (B,A): the two register sets
  // ping-pong, so no register copy ever has to wait for a load in flight.
  // PF == 1: called alternately with (A,B) / (B,A), prefetch at the top.  PF == 0: C and N are the same set; the
  // next plane is fetched as soon as p' and the x-update have consumed the current one (fewer VGPRs, shorter flight).
  // deferred x-update of plane kk:  x += alpha_prev * p_old
  auto xupdate = [&](int kk, int64_t pl, bool pown, const Raw &C, const double2 *xv) {
    if (!SQ || !pown) return;
#pragma unroll
    for (int m = 0; m < RY; ++m) {
      double2 xn;
      xn.x = xv[m].x + alpha_prev * C.p[m].x;
      xn.y = xv[m].y + alpha_prev * C.p[m].y;
      if (rown[m]) {
        if (own1) st2<NTS>(x + RO(m) + pl, xn);
        else if (own0) x[RO(m) + pl] = xn.x;
      }
    }
  };

  auto step = [&](int kk, Raw &C, Raw &N) {
    if (PF == 1) load(kk + 1, N);
    const double nzl = C.zl, nzc = C.zc, nzh = C.zh;

    const bool    pown = kk >= k0 && kk < k1;    // plane owned by this chunk: its p', x are stored here
    const bool    pgh  = kk == -1 || kk == g.nz;  // z-ghost plane: p' of the owned columns is stored too
    const double  dz   = C.zc;
    const int     buf  = (kk + 3) % 3;  // kk >= -1
    const int64_t pl   = (int64_t)kk * g.sxy;

    // p' of plane kk ------------------------------------------------------------------------------------------------
#pragma unroll
    for (int m = 0; m < RY; ++m) {
      const double dy = yc[m] + dz;
      const double z0 = JAC ? C.r[m].x / (xc0 + dy) : C.r[m].x;
      const double z1 = JAC ? C.r[m].y / (xc1 + dy) : C.r[m].y;
      double2      pn;
      pn.x = (z0 - zs) + beta * C.p[m].x;
      pn.y = (z1 - zs) + beta * C.p[m].y;
      pnext[m] = pn;
      // owned cells always; on owned planes also the high ghost row / column that lives inside the tile
      const bool st0 = rown[m] ? ((own0 && (pown || pgh)) || (gh0 && pown)) : (rgh[m] && own0 && pown);
      const bool st1 = rown[m] ? ((own1 && (pown || pgh)) || (gh1 && pown)) : (rgh[m] && own1 && pown);
      if (st0 && st1) st2<NTS>(pnew + RO(m) + pl, pn);
      else if (st0) pnew[RO(m) + pl] = pn.x;
      else if (st1) pnew[RO(m) + pl + 1] = pn.y;
    }
    const double hnA = ((JAC ? C.hrA / (hAdxy + dz) : C.hrA) - zs) + beta * C.hpA;
    const double hnB = ((JAC ? C.hrB / (hBdxy + dz) : C.hrB) - zs) + beta * C.hpB;
    if (pown) {
      if (hAgh) pnew[tbase + pl + hAo] = hnA;
      if (hBgh) pnew[tbase + pl + hBo] = hnB;
    }
    if (PF == 0) {
      xupdate(kk, pl, pown, C, C.x);
      load(kk + 1, N);  // N aliases C: every value of the old plane has been consumed
    }

    // q of plane kc = kk-1: centre, in-plane neighbours from lds[kc%3] (staged one trip ago), z-low from lds[(kc-1)%3] ------
    const int kc = kk - 1;
    if (kc >= k0) {
      const int     bc = (kc + 3) % 3, bp = (kc + 2) % 3;
      const int64_t pc = (int64_t)kc * g.sxy;
      const int     lc = 2 * lane + 2;
#pragma unroll
      for (int m = 0; m < RY; ++m) {
        const int     lr = w * RY + m + 1;
        const double2 cen   = *reinterpret_cast<const double2 *>(&lds[bc][lr][lc]);
        const double2 south = *reinterpret_cast<const double2 *>(&lds[bc][lr - 1][lc]);
        const double2 north = *reinterpret_cast<const double2 *>(&lds[bc][lr + 1][lc]);
        const double2 below = *reinterpret_cast<const double2 *>(&lds[bp][lr][lc]);
        const double  west = lds[bc][lr][lc - 1], east = lds[bc][lr][lc + 2];
        const double  dyc = yc[m] + zcc;
        double2       qq;
        qq.x = st7(xc0 + dyc, cen.x, xl0, west, xh0, cen.y, yl[m], south.x, yh[m], north.x, zlc, below.x, zhc, pnext[m].x);
        qq.y = st7(xc1 + dyc, cen.y, xl1, cen.x, xh1, east, yl[m], south.y, yh[m], north.y, zlc, below.y, zhc, pnext[m].y);
        if (rown[m]) {
          if (own1) {
            if (SQ) st2<NTS>(q + RO(m) + pc, qq);
            dot += cen.x * qq.x + cen.y * qq.y;
          } else if (own0) {
            if (SQ) q[RO(m) + pc] = qq.x;
            dot += cen.x * qq.x;
          }
          if (qbnd) {
            // the first / last plane and row of the block entirely, of the other rows the first and the last cell
            const int j = jb + m;
            if (kc == 0 || kc == g.nz - 1 || j == 0 || j == g.ny - 1) {
              if (own1) st2<0>(q + RO(m) + pc, qq);
              else if (own0) q[RO(m) + pc] = qq.x;
            } else {
              if (own0 && (i == 0 || i == g.nx - 1)) q[RO(m) + pc] = qq.x;
              if (own1 && i + 1 == g.nx - 1) q[RO(m) + pc + 1] = qq.y;
            }
          }
        }
      }
    }

    if (PF == 1) xupdate(kk, pl, pown, C, C.x);

    // stage plane kk for the next trip ------------------------------------------------------------------------------------
#pragma unroll
    for (int m = 0; m < RY; ++m) *reinterpret_cast<double2 *>(&lds[buf][w * RY + m + 1][2 * lane + 2]) = pnext[m];
    lds[buf][hAr][hAc] = hnA;
    lds[buf][hBr][hBc] = hnB;
    __syncthreads();
    zlc = nzl;
    zcc = nzc;
    zhc = nzh;
  };

  if (PF == 1) {
    Raw A, B;
    load(k0 - 1, A);
    for (int kk = k0 - 1; kk <= k1; kk += 2) {
      step(kk, A, B);
      if (kk + 1 <= k1) step(kk + 1, B, A);
    }
  } else {
    Raw A;
    load(k0 - 1, A);
    for (int kk = k0 - 1; kk <= k1; ++kk) step(kk, A, A);
  }

#undef RO
  dot = wave_sum(dot);
  if (lane == 0) red[w] = dot;
  __syncthreads();
  double tot[1] = {0.};
  if (tid == 0) {
#pragma unroll
    for (int a = 0; a < NW; ++a) tot[0] += red[a];
  }
  if (fin.enabled) fused_fin<1, 64 * NW>(SQ ? 1 : 3, tot, partial, 0, fin, s, red, &flag);
  else if (tid == 0) partial[blockIdx.x] = tot[0];
}

#define FL_CG_A_ARGS GridP g, const double *__restrict__ r, double *__restrict__ P0, double *__restrict__ P1, double *__restrict__ q, double *__restrict__ x, KspScal *__restrict__ s, double *__restrict__ partial, int nchunk, int zc, int tiles_x, int tiles, int remap, FinCtx fin
template <int RY, int NW, bool JAC, int PF, int NT, bool SQ>
__global__ void __launch_bounds__(64 * NW, FL_CGA_WPE) k_cg_A(FL_CG_A_ARGS)
{
  cg_A_body<RY, NW, JAC, PF, NT, SQ>(g, r, P0, P1, q, x, s, partial, nchunk, zc, tiles_x, tiles, remap, fin);
}
// The same code under another name: launched only by fl_poisson_tune_placement, so that profiles keep the probe
// launches (half of them on deliberately rejected placements) apart from the solver's own launches.
template <int RY, int NW, bool JAC, int PF, int NT, bool SQ>
__global__ void __launch_bounds__(64 * NW, 2) k_cg_A_probe(FL_CG_A_ARGS)
{
  cg_A_body<RY, NW, JAC, PF, NT, SQ>(g, r, P0, P1, q, x, s, partial, nchunk, zc, tiles_x, tiles, remap, fin);
}
#undef FL_CG_A_ARGS

// r -= alpha q with q = S p' FORMED AGAIN from the direction k_cg_A has just written, instead of being read back: k_cg_A<SQ = false>
// then never writes q.  Per cell this kernel reads p' (plus the tile's one-cell ring) and r and writes r -- 24 B + ring -- where the
// q store of k_cg_A and the q load of k_cg_B moved 16 B: an iteration moves 64 B/cell instead of 72.  Same tile walk as k_cg_A
// (128 x NW*RY tile marching through a z chunk, three p' planes in LDS, raw planes fetched one trip ahead into a second register
// set, unconditional loads on clamped addresses, masked stores); q comes out of st7 exactly as in k_cg_A.  The ghost layer of p'
// is complete: k_cg_A stores p' on every star-ghost cell it forms (rows / columns / planes -1 and n).  Sums as in k_cg_B.
// The x-update lives here too (XM): the direction is in registers anyway.  XM == 2 on odd iterations: x += alpha_old p_old + alpha p'
// (p_old = the other direction buffer, which k_cg_A overwrites only in the NEXT iteration), XM == 0 on even ones: nothing -- x is read
// and written every second iteration only (12 instead of 16 B/cell/iteration).  XM == 3: the first odd iteration of a solve, as XM == 2
// with x = 0 not read (the padded x is not zeroed by k_cg_init then).  XM == 1: x += alpha p' every iteration (A/B runs).
// The two fma of XM == 2 are the two separate updates in the same order: same x bit for bit.
template <int RY, int NW, bool JAC, int NT, int XM>
__device__ __forceinline__ void cg_Bq_body(const GridP &g, const double *__restrict__ P0, const double *__restrict__ P1, double *__restrict__ r, double *__restrict__ x, KspScal *__restrict__ s,
                                           double *__restrict__ partial, int stride, int nchunk, int zc, int tiles_x, int tiles, int remap, FinCtx fin)
{
  using T               = TileA<RY, NW>;
  constexpr int TX = T::TX, TY = T::TY, LX = T::LX, LY = T::LY;
  constexpr int NTL = NT >= 2, NTS = NT >= 1;
  __shared__ __attribute__((aligned(16))) double lds[3][LY][LX];
  __shared__ double                              red[5 * NW];
  __shared__ int                                 flag;
  if (s->reason != 0) return;
  const double *p     = s->cur ? P1 : P0;  // the scalar step after k_cg_A has flipped cur: this is the direction it wrote
  const double *pprev = s->cur ? P0 : P1;
  const double  alpha = s->alpha, alpha_old = s->alpha_old;

  const int b     = remap ? xcd_remap(blockIdx.x, gridDim.x) : blockIdx.x;
  const int chunk = b / tiles, tile = b % tiles;
  const int i0 = (tile % tiles_x) * TX, j0 = (tile / tiles_x) * TY;
  const int k0 = chunk * zc, k1 = min(k0 + zc, g.nz);
  const int tid = threadIdx.x, lane = tid & 63;
  const int w  = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int i  = i0 + 2 * lane;
  const int jb = j0 + w * RY;

  const bool   own0 = i < g.nx, own1 = i + 1 < g.nx;
  const int    il  = min(i, g.nx & ~1);
  const int    ic0 = min(i, g.nx), ic1 = min(i + 1, g.nx);
  const double xl0 = g.sl[0][ic0], xc0 = g.sc[0][ic0], xh0 = g.sh[0][ic0];
  const double xl1 = g.sl[0][ic1], xc1 = g.sc[0][ic1], xh1 = g.sh[0][ic1];
  int64_t rob[RY];
  bool    rown[RY];
  double  yl[RY], yc[RY], yh[RY];
#pragma unroll
  for (int m = 0; m < RY; ++m) {
    const int j = jb + m, jc = min(j, g.ny);
    rown[m]     = j < g.ny;
    rob[m]      = g.off0 + (int64_t)jc * g.sx;
    yl[m]       = g.sl[1][jc];
    yc[m]       = g.sc[1][jc];
    yh[m]       = g.sh[1][jc];
  }
#define RO(m) (rob[m] + il)
  // ring cells of this thread: as in k_cg_A (A: rows -1 / TY, B: columns -1 / TX)
  constexpr int HB0 = NW > 4 ? 256 : 0;
  const int     tb  = tid - HB0;
  const int     hAi = i0 + (tid & 127), hAj = j0 + (tid < 128 ? -1 : TY);
  const bool    hAok = tid < 256 && hAi < g.nx && hAj <= g.ny;
  const int     hBi = i0 + ((tb & 1) ? TX : -1), hBj = j0 + (tb >> 1);
  const bool    hBok = tb >= 0 && tb < 2 * TY && hBj < g.ny && hBi <= g.nx;
  const int64_t tbase = g.off0 + (int64_t)j0 * g.sx + i0;
  const int     hAo   = hAok ? (hAj - j0) * g.sx + (hAi - i0) : 0;
  const int     hBo   = hBok ? (hBj - j0) * g.sx + (hBi - i0) : 0;
  const int     hAr = hAok ? (tid < 128 ? 0 : TY + 1) : 0, hAc = hAok ? (tid & 127) + 2 : 0;  // (0,0) is a dead corner slot
  const int     hBr = hBok ? (tb >> 1) + 1 : 0, hBc = hBok ? ((tb & 1) ? TX + 2 : 1) : 0;

  double acc[5] = {0., 0., 0., 0., 0.};
  double zlc = 0., zcc = 0., zhc = 0.;  // z-row of plane kk-1 (the plane whose q is formed)
  struct Raw {
    double2 p[RY], r[RY];  // p' of plane kn, r of plane kn - 1
    double2 x[(XM == 1 || XM == 2) ? RY : 1], pp[XM >= 2 ? RY : 1];  // x (and the direction before) of plane kn - 1
    double  hpA, hpB;
    double  zl, zc, zh;
  };
  auto load = [&](int kn_, Raw &R) {
    const int     kn = min(kn_, k1);
    const int64_t pl = (int64_t)kn * g.sxy, pr = (int64_t)min(max(kn_ - 1, k0), k1 - 1) * g.sxy;
#pragma unroll
    for (int m = 0; m < RY; ++m) {
      R.p[m] = ld2<NTL>(p + RO(m) + pl);
      R.r[m] = ld2<NTL>(r + RO(m) + pr);
      if (XM == 1 || XM == 2) R.x[m] = ld2<NTL>(x + RO(m) + pr);
      if (XM >= 2) R.pp[m] = ld2<NTL>(pprev + RO(m) + pr);
    }
    R.hpA = p[tbase + pl + hAo];
    R.hpB = p[tbase + pl + hBo];
    R.zl  = g.sl[2][kn];
    R.zc  = g.sc[2][kn];
    R.zh  = g.sh[2][kn];
  };
  auto step = [&](int kk, Raw &C, Raw &N) {
    load(kk + 1, N);
    const double nzl = C.zl, nzc = C.zc, nzh = C.zh;
    const int    buf = (kk + 3) % 3;
    const int    kc  = kk - 1;
    if (kc >= k0) {
      const int     bc = (kc + 3) % 3, bp = (kc + 2) % 3;
      const int64_t pc = (int64_t)kc * g.sxy;
      const int     lc = 2 * lane + 2;
#pragma unroll
      for (int m = 0; m < RY; ++m) {
        const int     lr = w * RY + m + 1;
        const double2 cen   = *reinterpret_cast<const double2 *>(&lds[bc][lr][lc]);
        const double2 south = *reinterpret_cast<const double2 *>(&lds[bc][lr - 1][lc]);
        const double2 north = *reinterpret_cast<const double2 *>(&lds[bc][lr + 1][lc]);
        const double2 below = *reinterpret_cast<const double2 *>(&lds[bp][lr][lc]);
        const double  west = lds[bc][lr][lc - 1], east = lds[bc][lr][lc + 2];
        const double  dyc = yc[m] + zcc;
        const double  q0 = st7(xc0 + dyc, cen.x, xl0, west, xh0, cen.y, yl[m], south.x, yh[m], north.x, zlc, below.x, zhc, C.p[m].x);
        const double  q1 = st7(xc1 + dyc, cen.y, xl1, cen.x, xh1, east, yl[m], south.y, yh[m], north.y, zlc, below.y, zhc, C.p[m].y);
        double2       rn;
        rn.x = fma(-alpha, q0, C.r[m].x);  // the same rounding as k_pack_faces_rq
        rn.y = fma(-alpha, q1, C.r[m].y);
        const double z0 = JAC ? rn.x / (xc0 + dyc) : rn.x;
        const double z1 = JAC ? rn.y / (xc1 + dyc) : rn.y;
        if (rown[m]) {
          if (own1) st2<NTS>(r + RO(m) + pc, rn);
          else if (own0) r[RO(m) + pc] = rn.x;
        }
        if (XM) {
          double2 xn = XM == 3 ? make_double2(0., 0.) : C.x[m];  // XM == 3: the first pair of updates of a solve, x = 0 is not read
          if (XM >= 2) {
            xn.x = fma(alpha_old, C.pp[m].x, xn.x);
            xn.y = fma(alpha_old, C.pp[m].y, xn.y);
          }
          xn.x = fma(alpha, cen.x, xn.x);
          xn.y = fma(alpha, cen.y, xn.y);
          if (rown[m]) {
            if (own1) st2<NTS>(x + RO(m) + pc, xn);
            else if (own0) x[RO(m) + pc] = xn.x;
          }
        }
        // selects, not 0/1 factors: outside the block q is inf * 0 (the ghost diagonal is +inf)
        const bool   o0 = rown[m] && own0, o1 = rown[m] && own1;
        const double r0 = o0 ? rn.x : 0., r1 = o1 ? rn.y : 0., zz0 = o0 ? z0 : 0., zz1 = o1 ? z1 : 0.;
        acc[0] += r0 * zz0 + r1 * zz1;
        acc[1] += zz0 * zz0 + zz1 * zz1;
        acc[2] += zz0 + zz1;
        acc[3] += r0 + r1;
        acc[4] += r0 * r0 + r1 * r1;
      }
    }
#pragma unroll
    for (int m = 0; m < RY; ++m) *reinterpret_cast<double2 *>(&lds[buf][w * RY + m + 1][2 * lane + 2]) = C.p[m];
    lds[buf][hAr][hAc] = C.hpA;
    lds[buf][hBr][hBc] = C.hpB;
    __syncthreads();
    zlc = nzl;
    zcc = nzc;
    zhc = nzh;
  };
  {
    Raw A, B;
    load(k0 - 1, A);
    for (int kk = k0 - 1; kk <= k1; kk += 2) {
      step(kk, A, B);
      if (kk + 1 <= k1) step(kk + 1, B, A);
    }
  }
#undef RO
#pragma unroll
  for (int a = 0; a < 5; ++a) {
    acc[a] = wave_sum(acc[a]);
    if (lane == 0) red[a * NW + w] = acc[a];
  }
  __syncthreads();
  double tot[5] = {0., 0., 0., 0., 0.};
  if (tid == 0) {
#pragma unroll
    for (int a = 0; a < 5; ++a)
#pragma unroll
      for (int q = 0; q < NW; ++q) tot[a] += red[a * NW + q];
  }
  if (fin.enabled) fused_fin<5, 64 * NW>(XM ? 4 : 2, tot, partial, stride, fin, s, red, &flag);
  else if (tid == 0)
#pragma unroll
    for (int a = 0; a < 5; ++a) partial[(int64_t)a * stride + blockIdx.x] = tot[a];
}
#define FL_CG_BQ_ARGS GridP g, const double *__restrict__ P0, const double *__restrict__ P1, double *__restrict__ r, double *__restrict__ x, KspScal *__restrict__ s, double *__restrict__ partial, int stride, int nchunk, int zc, int tiles_x, int tiles, int remap, FinCtx fin
template <int RY, int NW, bool JAC, int NT, int XM>
__global__ void __launch_bounds__(64 * NW, FL_CGB_WPE) k_cg_Bq(FL_CG_BQ_ARGS)
{
  cg_Bq_body<RY, NW, JAC, NT, XM>(g, P0, P1, r, x, s, partial, stride, nchunk, zc, tiles_x, tiles, remap, fin);
}
// the placement probe's launches under their own name (see k_cg_A_probe)
template <int RY, int NW, bool JAC, int NT, int XM>
__global__ void __launch_bounds__(64 * NW, 2) k_cg_Bq_probe(FL_CG_BQ_ARGS)
{
  cg_Bq_body<RY, NW, JAC, NT, XM>(g, P0, P1, r, x, s, partial, stride, nchunk, zc, tiles_x, tiles, remap, fin);
}
#undef FL_CG_BQ_ARGS

// ------------------------------------------------------------------------------------------------ unfused CG pieces (variant 1)

// p = (r/diag - mean) + beta p on the owned cells
#ifdef FL_KBENCH_VARIANTS  // variant 1 of the CG solver: one kernel per BLAS-1 / SpMV step
template <bool JAC>
__global__ void k_cg_pupdate(GridP g, const double *__restrict__ r, double *__restrict__ P0, double *__restrict__ P1, const KspScal *__restrict__ s)
{
  if (s->reason != 0) return;
  const int i = blockIdx.x * 64 + threadIdx.x, j = blockIdx.y * 4 + threadIdx.y, k = blockIdx.z;
  if (i >= g.nx || j >= g.ny) return;
  const double *pold = s->cur ? P1 : P0;
  double       *pnew = s->cur ? P0 : P1;
  const int64_t o    = pidx(g, i, j, k);
  const double  z    = JAC ? r[o] / (g.sc[0][i] + g.sc[1][j] + g.sc[2][k]) : r[o];
  pnew[o]            = (z - s->zshift) + s->beta * pold[o];
}

// q = S p' (p' = the buffer k_cg_pupdate wrote), partial p'.q, deferred x update as in k_cg_A
__global__ void __launch_bounds__(256) k_cg_apply_dot(GridP g, const double *__restrict__ P0, const double *__restrict__ P1, double *__restrict__ q, double *__restrict__ x, const KspScal *__restrict__ s, double *__restrict__ partial)
{
  __shared__ double red[4];
  if (s->reason != 0) return;
  const int     i = blockIdx.x * 64 + threadIdx.x, j = blockIdx.y * 4 + threadIdx.y, k = blockIdx.z;
  const double *pold = s->cur ? P1 : P0;
  const double *pnew = s->cur ? P0 : P1;
  double        acc[1] = {0.};
  if (i < g.nx && j < g.ny) {
    const int64_t o = pidx(g, i, j, k);
    const double  v = stencil7(g, pnew, i, j, k);
    q[o]            = v;
    acc[0]          = pnew[o] * v;
    if (s->pending_x) x[o] += s->alpha * pold[o];
  }
  // blockDim = (64,4): linear thread id matches block_sum's expectations
  double v1[1] = {acc[0]};
  {
    const int lin = threadIdx.y * 64 + threadIdx.x;
    double    t   = wave_sum(v1[0]);
    if ((lin & 63) == 0) red[lin >> 6] = t;
    __syncthreads();
    if (lin == 0) partial[((int64_t)blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x] = (red[0] + red[1]) + (red[2] + red[3]);
  }
}
#endif  // FL_KBENCH_VARIANTS

// ------------------------------------------------------------------------------------------------ bandwidth reference
// NR input streams, NW output streams, 16 B per lane per access, grid-stride: the realistic HBM ceiling for a kernel with
// this read:write mix on this box (profiles/ and DESIGN.md quote it next to the solver kernels).
template <int NR, int NW>
__global__ void __launch_bounds__(256) k_stream_ref(int64_t n2, const double2 *__restrict__ a, const double2 *__restrict__ b, const double2 *__restrict__ c, double2 *__restrict__ o0, double2 *__restrict__ o1, double2 *__restrict__ o2)
{
  const int64_t step = (int64_t)gridDim.x * blockDim.x;
  for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < n2; t += 2 * step) {
    const int64_t t2 = min(t + step, n2 - 1);
    double2       v = a[t], w = a[t2];
    if (NR > 1) { const double2 u = b[t], u2 = b[t2]; v.x += u.x; v.y += u.y; w.x += u2.x; w.y += u2.y; }
    if (NR > 2) { const double2 u = c[t], u2 = c[t2]; v.x += u.x; v.y += u.y; w.x += u2.x; w.y += u2.y; }
    o0[t] = v;
    if (t + step < n2) o0[t2] = w;
    if (NW > 1) { o1[t] = v; if (t + step < n2) o1[t2] = w; }
    if (NW > 2) { o2[t] = v; if (t + step < n2) o2[t2] = w; }
  }
}

// parametric variant: U accesses of 16 B per lane per stream in flight, optional non-temporal hints
template <int NR, int NW, int U, bool NT>
__global__ void __launch_bounds__(256) k_stream_par(int64_t n2, const double2 *__restrict__ a, const double2 *__restrict__ b, const double2 *__restrict__ c, double2 *__restrict__ o0, double2 *__restrict__ o1, double2 *__restrict__ o2)
{
  const int64_t tile = 256 * U;
  for (int64_t base = (int64_t)blockIdx.x * tile; base < n2; base += (int64_t)gridDim.x * tile) {
    double2 v[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int64_t t = min(base + u * 256 + threadIdx.x, n2 - 1);
      if (NT) {
        v[u].x = __builtin_nontemporal_load(&a[t].x);
        v[u].y = __builtin_nontemporal_load(&a[t].y);
      } else v[u] = a[t];
    }
    if (NR > 1) {
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int64_t t = min(base + u * 256 + threadIdx.x, n2 - 1);
        double2       w;
        if (NT) { w.x = __builtin_nontemporal_load(&b[t].x); w.y = __builtin_nontemporal_load(&b[t].y); } else w = b[t];
        v[u].x += w.x; v[u].y += w.y;
      }
    }
    if (NR > 2) {
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int64_t t = min(base + u * 256 + threadIdx.x, n2 - 1);
        double2       w;
        if (NT) { w.x = __builtin_nontemporal_load(&c[t].x); w.y = __builtin_nontemporal_load(&c[t].y); } else w = c[t];
        v[u].x += w.x; v[u].y += w.y;
      }
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int64_t t = base + u * 256 + threadIdx.x;
      if (t < n2) {
        if (NT) {
          __builtin_nontemporal_store(v[u].x, &o0[t].x); __builtin_nontemporal_store(v[u].y, &o0[t].y);
          if (NW > 1) { __builtin_nontemporal_store(v[u].x, &o1[t].x); __builtin_nontemporal_store(v[u].y, &o1[t].y); }
          if (NW > 2) { __builtin_nontemporal_store(v[u].x, &o2[t].x); __builtin_nontemporal_store(v[u].y, &o2[t].y); }
        } else {
          o0[t] = v[u];
          if (NW > 1) o1[t] = v[u];
          if (NW > 2) o2[t] = v[u];
        }
      }
    }
  }
}

// ------------------------------------------------------------------------------------------------ launch helpers

static inline dim3 grid3(int nx, int ny, int nz) { return dim3((nx + 63) / 64, (ny + 3) / 4, nz); }
static inline dim3 blk3() { return dim3(64, 4, 1); }

}  // namespace fl

// ================================================================================================ host-callable launchers
// (kept in this translation unit so that only hipcc sees <<< >>>)

namespace fl {

void launch_pad_copy(hipStream_t st, const GridP &g, const double *src, double *dst) { hipLaunchKernelGGL(k_pad_copy, grid3(g.nx, g.ny, g.nz), blk3(), 0, st, g, src, dst); }
void launch_unpad_copy(hipStream_t st, const GridP &g, const double *src, double *dst, const double *shift) { hipLaunchKernelGGL(k_unpad_copy, grid3(g.nx, g.ny, g.nz), blk3(), 0, st, g, src, dst, shift); }
void launch_wrap(hipStream_t st, const GridP &g, double *v, int axis, int nvec, int64_t vstride)
{
  const int na = axis == 0 ? g.ny : g.nx, nb = axis == 2 ? g.ny : g.nz;
  hipLaunchKernelGGL(k_wrap_ghosts, grid3(na, nb, nvec), blk3(), 0, st, g, v, axis, vstride);
}
void launch_face_ext(hipStream_t st, const GridP &g, double *v, double *buf, int axis, int side, int ea, int eb, int mode)
{
  const int na = (axis == 0 ? g.ny : g.nx) + 2 * ea, nb = (axis == 2 ? g.ny : g.nz) + 2 * eb;
  hipLaunchKernelGGL(k_face_ext, grid3(na, nb, 1), blk3(), 0, st, g, v, buf, axis, side, ea, eb, mode);
}
void launch_face_ext_deep(hipStream_t st, const GridP &g, double *v, double *buf, int axis, int side, int ea, int eb, int dp, int mode)
{
  const int na = (axis == 0 ? g.ny : g.nx) + 2 * ea, nb = (axis == 2 ? g.ny : g.nz) + 2 * eb;
  hipLaunchKernelGGL(k_face_ext_deep, grid3(na, nb, 1), blk3(), 0, st, g, v, buf, axis, side, ea, eb, dp, mode);
}
void launch_pack(hipStream_t st, const GridP &g, const double *v, double *buf, int axis, int side)
{
  const int na = axis == 0 ? g.ny : g.nx, nb = axis == 2 ? g.ny : g.nz;
  hipLaunchKernelGGL(k_pack_face, grid3(na, nb, 1), blk3(), 0, st, g, v, buf, axis, side);
}
void launch_unpack(hipStream_t st, const GridP &g, double *v, const double *buf, int axis, int side)
{
  const int na = axis == 0 ? g.ny : g.nx, nb = axis == 2 ? g.ny : g.nz;
  hipLaunchKernelGGL(k_unpack_face, grid3(na, nb, 1), blk3(), 0, st, g, v, buf, axis, side);
}
// bufs[b] may be NULL; one launch covers the largest plane in x/y and the six boundaries in z
void launch_pack_faces(hipStream_t st, const GridP &g, const double *v, double *const bufs[6])
{
  FaceBufs fb;
  for (int b = 0; b < 6; ++b) fb.buf[b] = bufs[b];
  const int na = std::max(g.nx, g.ny), nb = std::max(g.ny, g.nz);
  dim3      grid((na + 63) / 64, (nb + 3) / 4, 6);
  hipLaunchKernelGGL(k_pack_faces, grid, dim3(64, 4), 0, st, g, v, fb);
}
void launch_pack_faces_rq(hipStream_t st, const GridP &g, const double *r, const double *q, const KspScal *s, double *const bufs[6])
{
  FaceBufs fb;
  for (int b = 0; b < 6; ++b) fb.buf[b] = bufs[b];
  const int na = std::max(g.nx, g.ny), nb = std::max(g.ny, g.nz);
  dim3      grid((na + 63) / 64, (nb + 3) / 4, 6);
  hipLaunchKernelGGL(k_pack_faces_rq, grid, dim3(64, 4), 0, st, g, r, q, s, fb);
}
void launch_pack_faces_sr(hipStream_t st, const GridP &g, const double *r, const double *sb, const double *W, const KspScal *s, double *const bufs[6])
{
  FaceBufs fb;
  for (int b = 0; b < 6; ++b) fb.buf[b] = bufs[b];
  const int na = std::max(g.nx, g.ny), nb = std::max(g.ny, g.nz);
  dim3      grid((na + 63) / 64, (nb + 3) / 4, 6);
  hipLaunchKernelGGL(k_pack_faces_sr, grid, dim3(64, 4), 0, st, g, r, sb, W, s, fb);
}
void launch_unpack_faces(hipStream_t st, const GridP &g, double *v, double *const bufs[6])
{
  FaceBufs fb;
  for (int b = 0; b < 6; ++b) fb.buf[b] = bufs[b];
  const int na = std::max(g.nx, g.ny), nb = std::max(g.ny, g.nz);
  dim3      grid((na + 63) / 64, (nb + 3) / 4, 6);
  hipLaunchKernelGGL(k_unpack_faces, grid, dim3(64, 4), 0, st, g, v, fb);
}
void launch_apply(hipStream_t st, const GridP &g, const double *xpad, double *y, int ypad) { hipLaunchKernelGGL(k_apply, grid3(g.nx, g.ny, g.nz), blk3(), 0, st, g, xpad, y, ypad); }
void launch_diagonal(hipStream_t st, const GridP &g, double *d) { hipLaunchKernelGGL(k_diagonal, grid3(g.nx, g.ny, g.nz), blk3(), 0, st, g, d); }
void launch_rhs(hipStream_t st, const GridP &g, const double *Vx, const double *Vy, const double *Vz, const double *hix, const double *hiy, const double *hiz, const double *contrhs, double *b)
{
  hipLaunchKernelGGL(k_rhs, grid3(g.nx, g.ny, g.nz), blk3(), 0, st, g, Vx, Vy, Vz, hix, hiy, hiz, contrhs, b);
}
void launch_face_plane0(hipStream_t st, const GridP &g, const double *V, double *buf, int axis)
{
  const int na = axis == 0 ? g.ny : g.nx, nb = axis == 2 ? g.ny : g.nz;
  hipLaunchKernelGGL(k_face_plane0, grid3(na, nb, 1), blk3(), 0, st, g, V, buf, axis);
}
void launch_project_faces(hipStream_t st, const GridP &g, const double *p, double *V, int axis)
{
#ifdef FL_KBENCH_VARIANTS
  const int lx = axis == 0 ? g.fx : g.nx, ly = axis == 1 ? g.fy : g.ny, lz = axis == 2 ? g.fz : g.nz;
  if (lz > 0) hipLaunchKernelGGL(k_project_faces, grid3(lx, ly, lz), blk3(), 0, st, g, p, V, axis);
#endif
}
void launch_project_all(hipStream_t st, const GridP &g, const double *p, double *const v[3], double *const V[3])
{
  ProjOut o;
  for (int d = 0; d < 3; ++d) {
    o.v[d] = v[d];
    o.V[d] = V[d];
  }
  const int     nseg = (g.nx + 127) / 128;
  const int64_t items = (int64_t)nseg * g.ny * g.nz;
  int64_t       nwaves = std::max<int64_t>(nseg, std::min<int64_t>(items, 4 * 8192));
  nwaves = (nwaves + 4 * nseg - 1) / (4 * nseg) * (4 * nseg);  // whole blocks of four waves, every wave keeps its segment
  int pairs = g.nx % 2 == 0;  // 16-byte accesses to the caller's unpadded arrays: even rows and aligned bases
  for (int d = 0; d < 3; ++d)
    if ((v[d] && (reinterpret_cast<uintptr_t>(v[d]) & 15)) || (d > 0 && V[d] && (reinterpret_cast<uintptr_t>(V[d]) & 15))) pairs = 0;
  hipLaunchKernelGGL(k_project_all, dim3((unsigned)(nwaves / 4)), dim3(256), 0, st, g, p, o, pairs);
}
// k_project_six is for all six arrays, even rows, 16-byte aligned bases and (DIRECT) a 16-byte aligned unpadded p
bool project_six_usable(const GridP &g, const double *p_unpadded, double *const v[3], double *const V[3])
{
  if (g.nx % 2 != 0 || (int64_t)g.ny * g.nz >= ((int64_t)1 << 31)) return false;
  for (int d = 0; d < 3; ++d)
    if (!v[d] || !V[d] || (reinterpret_cast<uintptr_t>(v[d]) & 15) || (d > 0 && (reinterpret_cast<uintptr_t>(V[d]) & 15))) return false;
  return !p_unpadded || (reinterpret_cast<uintptr_t>(p_unpadded) & 15) == 0;
}
// direct: p is the caller's unpadded array, per = bit d set where axis d is periodic (single rank); else p is padded with its ghost layers filled
void launch_project_six(hipStream_t st, const GridP &g, const double *p, bool direct, int per, double *const v[3], double *const V[3])
{
  ProjOut o;
  for (int d = 0; d < 3; ++d) {
    o.v[d] = v[d];
    o.V[d] = V[d];
  }
  // experiments (tools/experiments/r04_project.sh): FLUCA_PROJECT_VAR="nt,nxcd,blocks_per_xcd".  512^3 (profiles/r04_project.txt): 1024 - 2048 blocks
  // per XCD 2.69 ms, 256: 2.97, 16384 (a row per wave): 2.87; without the slabs (nxcd 1) 2.76 - 2.86; without the hint 2.78; two rows per pass
  // (244 VGPRs) 2.86; capped at 128 VGPRs 2.85
  struct Var { int nt = 1, nxcd = 8, nbx = 1024; };
  Var var;
  if (const char *e = variant_env("FLUCA_PROJECT_VAR")) std::sscanf(e, "%d,%d,%d", &var.nt, &var.nxcd, &var.nbx);
  const int nseg = (g.nx + 127) / 128;
  const int nxcd = (var.nxcd == 1 || g.ny < 8) ? 1 : 8;
  // blocks of four waves per XCD: a multiple of nseg (a wave keeps its segment), no more than the rows of a slab need
  const int64_t items = (int64_t)nseg * ((g.ny + nxcd - 1) / nxcd) * g.nz;
  int64_t       nbx = std::max<int64_t>(1, std::min<int64_t>((items + 3) / 4, (int64_t)std::max(var.nbx, 1) * (8 / nxcd)));
  nbx = (nbx + nseg - 1) / nseg * nseg;
  const dim3 gr((unsigned)(nbx * nxcd)), bl(256);
  if (direct) {
    if (var.nt) hipLaunchKernelGGL((k_project_six<true, 1, 1>), gr, bl, 0, st, g, p, o, per, nxcd);
    else hipLaunchKernelGGL((k_project_six<true, 0, 1>), gr, bl, 0, st, g, p, o, per, nxcd);
  } else {
    if (var.nt) hipLaunchKernelGGL((k_project_six<false, 1, 1>), gr, bl, 0, st, g, p, o, per, nxcd);
    else hipLaunchKernelGGL((k_project_six<false, 0, 1>), gr, bl, 0, st, g, p, o, per, nxcd);
  }
}
void launch_project_cells(hipStream_t st, const GridP &g, const double *p, double *v, int axis)
{
#ifdef FL_KBENCH_VARIANTS
  hipLaunchKernelGGL(k_project_cells, grid3(g.nx, g.ny, g.nz), blk3(), 0, st, g, p, v, axis);
#endif
}
void launch_gst_bc(hipStream_t st, const GridP &g, const double *pb, double *V, int axis, int side, double coeff, int add)
{
  const int na = axis == 0 ? g.ny : g.nx, nb = axis == 2 ? g.ny : g.nz;
  hipLaunchKernelGGL(k_gst_bc, grid3(na, nb, 1), blk3(), 0, st, g, pb, V, axis, side, coeff, add);
}
void launch_bc_add_cells(hipStream_t st, const GridP &g, const double *plane, double *cells, int axis, int side, double coeff)
{
  const int na = axis == 0 ? g.ny : g.nx, nb = axis == 2 ? g.ny : g.nz;
  hipLaunchKernelGGL(k_bc_add_cells, grid3(na, nb, 1), blk3(), 0, st, g, plane, cells, axis, side, coeff);
}
void launch_pressure_update(hipStream_t st, int64_t n, int first, const double *dp, const double *p0, double *phalf, double *p)
{
  const int nb = (int)std::min<int64_t>((n + 255) / 256, 4096);
  hipLaunchKernelGGL(k_pressure_update, dim3(nb), dim3(256), 0, st, n, first, dp, p0, phalf, p);
}
void launch_reduce(hipStream_t st, const double *partial, int nblocks, int stride, int nslot, double *sums) { hipLaunchKernelGGL(k_reduce, dim3(1), dim3(256), 0, st, partial, nblocks, stride, nslot, sums); }
void launch_cg_fin(hipStream_t st, int mode, const double *partial, int nblocks, int stride, const double *sums, KspScal *s, double *hist, int nhist)
{
  hipLaunchKernelGGL(k_cg_fin, dim3(1), dim3(256), 0, st, mode, partial, nblocks, stride, sums, s, hist, nhist);
}

int stream_blocks(const GridP &g)
{
  const int64_t nseg = (int64_t)((g.nx + 127) / 128) * g.ny * g.nz;
  return (int)std::max<int64_t>(1, std::min<int64_t>((nseg + 7) / 8, 2048));
}

// 16-byte accesses to a caller's unpadded array need even rows and an aligned base
static int unpadded_pairs(const GridP &g, const void *a) { return (g.nx % 2 == 0 && (reinterpret_cast<uintptr_t>(a) & 15) == 0) ? 1 : 0; }

void launch_cg_init(hipStream_t st, const GridP &g, bool jac, const double *b, double *r, double *x0, double *partial, int stride, int nblocks)
{
  const int pairs = unpadded_pairs(g, b);
  if (jac) hipLaunchKernelGGL(k_cg_init<true>, dim3(nblocks), dim3(256), 0, st, g, b, r, x0, partial, stride, pairs);
  else hipLaunchKernelGGL(k_cg_init<false>, dim3(nblocks), dim3(256), 0, st, g, b, r, x0, partial, stride, pairs);
}
void launch_cg_finish(hipStream_t st, const GridP &g, const double *P0, const double *P1, const double *x, double *xout, const KspScal *s, int nblocks)
{
  hipLaunchKernelGGL(k_cg_finish, dim3(nblocks), dim3(256), 0, st, g, P0, P1, x, xout, s, unpadded_pairs(g, xout));
}

// tiling of k_cg_A: returns the number of blocks
struct PlanA {
  int ry, nw, tiles_x, tiles_y, nchunk, zc, nblocks, pf, nt, remap, probe;
  int sq;  // 1: k_cg_A stores q (k_cg_B reads it back); 0: q is formed again by k_cg_Bq (the default of the solver)
  int qb;  // sq == 0 only: k_cg_A still stores q on the six boundary layers of the block (the overlapped halo exchange packs r - alpha q there)
};
// tiling of k_cg_A / k_cg_B (K_B always runs 4-wave blocks: it reuses ry, nchunk, zc with nw = 4)
PlanA plan_tiles(const GridP &g, int ry, int nw, int nchunk_force, int target_blocks, int min_zc = 8)
{
  PlanA p;
  p.ry = ry;
  p.nw = nw;
  p.pf = 0;
  p.nt = 0;
  p.remap   = 1;
  p.probe   = 0;
  p.sq      = 0;
  p.qb      = 0;
  p.tiles_x = (g.nx + 127) / 128;
  p.tiles_y = (g.ny + nw * ry - 1) / (nw * ry);
  const int tiles = p.tiles_x * p.tiles_y;
  int       nchunk = nchunk_force > 0 ? nchunk_force : std::max(1, (target_blocks + tiles / 2) / tiles);
  if (nchunk_force <= 0) nchunk = std::min(nchunk, std::max(1, g.nz / min_zc));  // min_zc = 8 keeps the 2-plane chunk prologue <= 25 %
  nchunk    = std::max(1, std::min(nchunk, g.nz));
  p.zc      = (g.nz + nchunk - 1) / nchunk;
  p.nchunk  = (g.nz + p.zc - 1) / p.zc;
  p.nblocks = tiles * p.nchunk;
  return p;
}
// Defaults from the tools/kbench.py sweeps on MI355X at 512^3 (profiles/r01_kbench*.txt, three different boxes): 128 x 16
// tiles of 8 waves x 2 rows, two ping-pong prefetch sets, non-temporal tile loads and stores, XCD-contiguous
// chunk-major block order.  Grids too small to give every CU a block with that shape (the reference's own 64 x 64 x 32
// cavity makes 16) are latency-bound, not bandwidth-bound: they get smaller tiles and short z chunks instead, the chunk
// prologue no longer matters (tools/experiments/small_grid_plan.py: k_cg_A 23.4 -> 6.9 us, k_cg_B 9.6 -> 4.7 us there).
constexpr int MIN_BLOCKS = 256;  // one per CU
// Plans from the sweeps of tools/experiments/sweep256.py at 128^3, 256^3 and 512^3 (profiles/r02_plan_sweep.txt): both kernels are
// fastest with about one block per CU (k_cg_A) or one to two (k_cg_B) and as few z chunks as that allows -- every chunk pays a
// two-plane prologue in k_cg_A and a ramp in both.  512^3: k_cg_A 128 x 16 tiles x 2 chunks (1094 us; 4 chunks 1106, 128 x 8 tiles
// 1123), k_cg_B 128 x 16 x 2 chunks (543 us; the 1024 blocks of round 1: 585).  256^3: k_cg_A 128 x 8 x 4 chunks (151 us; the
// round-1 plan of 128 x 16 x 16 chunks: 164), k_cg_B 128 x 4 x 4 chunks (70 us; round 1: 128 x 16 x 32 chunks, 108).
PlanA plan_cg_A(const GridP &g, int ry_force, int nchunk_force)
{
  const int target_env = FL_VARIANT(cga_target, 0);
  const int target = target_env > 0 ? target_env : 256;
  const int ry = ry_force > 0 ? ry_force : (g.ny >= 8 ? 2 : 1);
  // 128 x 16 tiles (8 waves) when they alone nearly fill the chip, 128 x 8 (4 waves) below that
  const int tiles16 = ((g.nx + 127) / 128) * ((g.ny + 8 * ry - 1) / (8 * ry));
  const int nw      = (ry == 2 && g.ny >= 32 && tiles16 >= 128) ? 8 : 4;
  PlanA     p       = plan_tiles(g, ry, nw, nchunk_force, target);
  if (ry_force <= 0 && nchunk_force <= 0 && p.nblocks < MIN_BLOCKS) {
    p = plan_tiles(g, ry, 4, 0, 512, 2);
    if (p.nblocks < MIN_BLOCKS && ry == 2) p = plan_tiles(g, 1, 4, 0, 512, 2);
  }
  // 256^3 .. 512^3 (round 4, profiles/r04_cg256.txt, r04_cg_plans.txt): an 8-wave block holds 147 - 186 VGPRs, so a CU runs ONE of them; the pair is
  // fastest when every CU has exactly one -- as many z chunks of 128 x 16 tiles as fit into 256 blocks, never a second round (384^3: 216 blocks
  // 0.633 ms per iteration against 0.844 with the 288 four-wave blocks of round 2's rule; 256^3: 0.188 - 0.198 against 0.204; 512^3 unchanged)
  if (ry_force <= 0 && nchunk_force <= 0 && target_env <= 0 && ry == 2 && g.ny >= 32 && (int64_t)g.nx * g.ny * g.nz >= ((int64_t)1 << 24) && tiles16 < 128) {
    const int nchunk = std::max(1, std::min(256 / tiles16, g.nz / 8));
    if (tiles16 * nchunk >= 192) p = plan_tiles(g, 2, 8, nchunk, 0);
  }
  p.pf         = 1;
  p.nt         = 2;
  // experiments (tools/experiments/r04_cg256.sh): FLUCA_CG_PLAN="ry,nw,nchunk" replaces the tiling of k_cg_A / k_cg_Bq on every grid
  struct Force { int ry = 0, nw = 0, nchunk = 0; };
  Force force;
  if (const char *e = variant_env("FLUCA_CG_PLAN")) std::sscanf(e, "%d,%d,%d", &force.ry, &force.nw, &force.nchunk);
  if (ry_force <= 0 && nchunk_force <= 0 && (force.ry == 1 || force.ry == 2) && (force.nw == 4 || (force.nw == 8 && force.ry == 2)) && g.ny >= 8) {
    p    = plan_tiles(g, force.ry, force.nw, std::max(force.nchunk, 1), 0);
    p.pf = 1;
    p.nt = 2;
  }
  return p;
}
PlanA plan_cg_B(const GridP &g)
{
  const int64_t cells = (int64_t)g.nx * g.ny * g.nz;
  const bool    big   = cells >= ((int64_t)1 << 26);  // 512^3: few fat blocks; below: one row per wave, two blocks per CU
  int           ry = big ? (g.ny >= 32 ? 4 : (g.ny >= 8 ? 2 : 1)) : 1;
  const int     target = big ? 256 : 512;
  PlanA         p  = plan_tiles(g, ry, 4, 0, target);
  while (p.nblocks < MIN_BLOCKS) {
    p = plan_tiles(g, ry, 4, 0, target, 2);
    if (p.nblocks >= MIN_BLOCKS || ry == 1) break;
    ry /= 2;
  }
  p.nt         = 1;
  return p;
}

template <int RY, int NW, int PF, int NT, bool SQ>
static void launch_cg_A_q(hipStream_t st, const GridP &g, bool jac, const PlanA &p, const double *r, double *P0, double *P1, double *q, double *x, KspScal *s, double *partial, const FinCtx &fin)
{
  const int  tiles = p.tiles_x * p.tiles_y;
  const int  rq    = (p.remap ? 1 : 0) | (p.qb ? 2 : 0);  // bit 0: XCD-contiguous block order, bit 1: q on the boundary layers
  const dim3 gr(p.nblocks), bl(64 * NW);
  if (p.probe) {
    if (jac) hipLaunchKernelGGL((k_cg_A_probe<RY, NW, true, PF, NT, SQ>), gr, bl, 0, st, g, r, P0, P1, q, x, s, partial, p.nchunk, p.zc, p.tiles_x, tiles, rq, fin);
    else hipLaunchKernelGGL((k_cg_A_probe<RY, NW, false, PF, NT, SQ>), gr, bl, 0, st, g, r, P0, P1, q, x, s, partial, p.nchunk, p.zc, p.tiles_x, tiles, rq, fin);
  } else {
    if (jac) hipLaunchKernelGGL((k_cg_A<RY, NW, true, PF, NT, SQ>), gr, bl, 0, st, g, r, P0, P1, q, x, s, partial, p.nchunk, p.zc, p.tiles_x, tiles, rq, fin);
    else hipLaunchKernelGGL((k_cg_A<RY, NW, false, PF, NT, SQ>), gr, bl, 0, st, g, r, P0, P1, q, x, s, partial, p.nchunk, p.zc, p.tiles_x, tiles, rq, fin);
  }
}
template <int RY, int NW, int PF, int NT>
static void launch_cg_A_t(hipStream_t st, const GridP &g, bool jac, const PlanA &p, const double *r, double *P0, double *P1, double *q, double *x, KspScal *s, double *partial, const FinCtx &fin)
{
#ifdef FL_KBENCH_VARIANTS
  if (p.sq) {  // variant 2: q stored
    launch_cg_A_q<RY, NW, PF, NT, true>(st, g, jac, p, r, P0, P1, q, x, s, partial, fin);
    return;
  }
#endif
  launch_cg_A_q<RY, NW, PF, NT, false>(st, g, jac, p, r, P0, P1, q, x, s, partial, fin);
}
template <int RY, int NW>
static void launch_cg_A_v(hipStream_t st, const GridP &g, bool jac, const PlanA &p, const double *r, double *P0, double *P1, double *q, double *x, KspScal *s, double *partial, const FinCtx &fin)
{
#ifdef FL_KBENCH_VARIANTS  // the whole sweep space of tools/kbench.py (build with -DFL_KBENCH_VARIANTS)
  switch (p.pf * 10 + p.nt) {
  case 0: launch_cg_A_t<RY, NW, 0, 0>(st, g, jac, p, r, P0, P1, q, x, s, partial, fin); return;
  case 1: launch_cg_A_t<RY, NW, 0, 1>(st, g, jac, p, r, P0, P1, q, x, s, partial, fin); return;
  case 2: launch_cg_A_t<RY, NW, 0, 2>(st, g, jac, p, r, P0, P1, q, x, s, partial, fin); return;
  case 10: launch_cg_A_t<RY, NW, 1, 0>(st, g, jac, p, r, P0, P1, q, x, s, partial, fin); return;
  case 11: launch_cg_A_t<RY, NW, 1, 1>(st, g, jac, p, r, P0, P1, q, x, s, partial, fin); return;
  default: break;
  }
#endif
  launch_cg_A_t<RY, NW, 1, 2>(st, g, jac, p, r, P0, P1, q, x, s, partial, fin);  // the shipped variant
}
void launch_cg_A(hipStream_t st, const GridP &g, bool jac, const PlanA &p, const double *r, double *P0, double *P1, double *q, double *x, KspScal *s, double *partial, unsigned *counter, double *hist, int nhist, double *sums)
{
  FinCtx fin;
  fin.sums    = sums;
  fin.counter = counter;
  fin.hist    = hist;
  fin.nhist   = nhist;
  fin.enabled = counter != nullptr;
  switch (p.ry * 10 + p.nw) {
#ifdef FL_KBENCH_VARIANTS
  case 48: launch_cg_A_v<4, 8>(st, g, jac, p, r, P0, P1, q, x, s, partial, fin); break;
  case 44: launch_cg_A_v<4, 4>(st, g, jac, p, r, P0, P1, q, x, s, partial, fin); break;
#endif
  case 28: launch_cg_A_v<2, 8>(st, g, jac, p, r, P0, P1, q, x, s, partial, fin); break;
  case 24: launch_cg_A_v<2, 4>(st, g, jac, p, r, P0, P1, q, x, s, partial, fin); break;
  default: launch_cg_A_v<1, 4>(st, g, jac, p, r, P0, P1, q, x, s, partial, fin); break;
  }
}

#ifdef FL_KBENCH_VARIANTS  // variant 2 of the CG pair (q stored, k_cg_B reads it back)
template <int RY>
static void launch_cg_B_ry(hipStream_t st, const GridP &g, bool jac, const PlanA &p, const double *q, double *r, KspScal *s, double *partial, int stride, const FinCtx &fin)
{
  const dim3 gr(p.nblocks), bl(256);
  if (p.nt) {
    if (jac) hipLaunchKernelGGL((k_cg_B<RY, true, 1>), gr, bl, 0, st, g, q, r, s, partial, stride, p.nchunk, p.zc, p.tiles_x, fin);
    else hipLaunchKernelGGL((k_cg_B<RY, false, 1>), gr, bl, 0, st, g, q, r, s, partial, stride, p.nchunk, p.zc, p.tiles_x, fin);
  } else {
    if (jac) hipLaunchKernelGGL((k_cg_B<RY, true, 0>), gr, bl, 0, st, g, q, r, s, partial, stride, p.nchunk, p.zc, p.tiles_x, fin);
    else hipLaunchKernelGGL((k_cg_B<RY, false, 0>), gr, bl, 0, st, g, q, r, s, partial, stride, p.nchunk, p.zc, p.tiles_x, fin);
  }
}
#endif  // FL_KBENCH_VARIANTS
void launch_cg_B(hipStream_t st, const GridP &g, bool jac, const PlanA &p, const double *q, double *r, KspScal *s, double *partial, int stride, unsigned *counter, double *hist, int nhist, double *sums)
{
#ifdef FL_KBENCH_VARIANTS
  FinCtx fin;
  fin.sums    = sums;
  fin.counter = counter;
  fin.hist    = hist;
  fin.nhist   = nhist;
  fin.enabled = counter != nullptr;
  switch (p.ry) {
  case 4: launch_cg_B_ry<4>(st, g, jac, p, q, r, s, partial, stride, fin); break;
  case 2: launch_cg_B_ry<2>(st, g, jac, p, q, r, s, partial, stride, fin); break;
  default: launch_cg_B_ry<1>(st, g, jac, p, q, r, s, partial, stride, fin); break;
  }
#endif
}

// k_cg_Bq on the tiling of k_cg_A (plan_cg_A).  xmode: 0 no x-update, 1 x += alpha p', 2 the two updates owed on odd iterations
template <int RY, int NW, int XM>
static void launch_cg_Bq_x(hipStream_t st, const GridP &g, bool jac, const PlanA &p, const double *P0, const double *P1, double *r, double *x, KspScal *s, double *partial, int stride, const FinCtx &fin)
{
  const int  tiles = p.tiles_x * p.tiles_y;
  const dim3 gr(p.nblocks), bl(64 * NW);
  if (p.probe) hipLaunchKernelGGL((k_cg_Bq_probe<RY, NW, true, 2, XM>), gr, bl, 0, st, g, P0, P1, r, x, s, partial, stride, p.nchunk, p.zc, p.tiles_x, tiles, p.remap, fin);
  else if (jac) hipLaunchKernelGGL((k_cg_Bq<RY, NW, true, 2, XM>), gr, bl, 0, st, g, P0, P1, r, x, s, partial, stride, p.nchunk, p.zc, p.tiles_x, tiles, p.remap, fin);
  else hipLaunchKernelGGL((k_cg_Bq<RY, NW, false, 2, XM>), gr, bl, 0, st, g, P0, P1, r, x, s, partial, stride, p.nchunk, p.zc, p.tiles_x, tiles, p.remap, fin);
}
template <int RY, int NW>
static void launch_cg_Bq_t(hipStream_t st, const GridP &g, bool jac, const PlanA &p, int xmode, const double *P0, const double *P1, double *r, double *x, KspScal *s, double *partial, int stride, const FinCtx &fin)
{
  if (xmode == 3) launch_cg_Bq_x<RY, NW, 3>(st, g, jac, p, P0, P1, r, x, s, partial, stride, fin);
  else if (xmode == 2) launch_cg_Bq_x<RY, NW, 2>(st, g, jac, p, P0, P1, r, x, s, partial, stride, fin);
  else if (xmode == 1) launch_cg_Bq_x<RY, NW, 1>(st, g, jac, p, P0, P1, r, x, s, partial, stride, fin);
  else launch_cg_Bq_x<RY, NW, 0>(st, g, jac, p, P0, P1, r, x, s, partial, stride, fin);
}
void launch_cg_Bq(hipStream_t st, const GridP &g, bool jac, const PlanA &p, int xmode, const double *P0, const double *P1, double *r, double *x, KspScal *s, double *partial, int stride, unsigned *counter, double *hist,
                  int nhist, double *sums)
{
  FinCtx fin;
  fin.sums    = sums;
  fin.counter = counter;
  fin.hist    = hist;
  fin.nhist   = nhist;
  fin.enabled = counter != nullptr;
  switch (p.ry * 10 + p.nw) {
  case 28: launch_cg_Bq_t<2, 8>(st, g, jac, p, xmode, P0, P1, r, x, s, partial, stride, fin); break;
  case 24: launch_cg_Bq_t<2, 4>(st, g, jac, p, xmode, P0, P1, r, x, s, partial, stride, fin); break;
  default: launch_cg_Bq_t<1, 4>(st, g, jac, p, xmode, P0, P1, r, x, s, partial, stride, fin); break;
  }
}

void launch_stream_ref(hipStream_t st, int nr, int nw, int64_t n2, const double *a, const double *b, const double *c, double *o0, double *o1, double *o2)
{
  const dim3 gr(2048), bl(256);
  const double2 *A = (const double2 *)a, *B = (const double2 *)b, *C = (const double2 *)c;
  double2       *O0 = (double2 *)o0, *O1 = (double2 *)o1, *O2 = (double2 *)o2;
  switch (nr * 10 + nw) {
  case 11: hipLaunchKernelGGL((k_stream_ref<1, 1>), gr, bl, 0, st, n2, A, B, C, O0, O1, O2); break;
  case 21: hipLaunchKernelGGL((k_stream_ref<2, 1>), gr, bl, 0, st, n2, A, B, C, O0, O1, O2); break;
  case 22: hipLaunchKernelGGL((k_stream_ref<2, 2>), gr, bl, 0, st, n2, A, B, C, O0, O1, O2); break;
  case 32: hipLaunchKernelGGL((k_stream_ref<3, 2>), gr, bl, 0, st, n2, A, B, C, O0, O1, O2); break;
  default: hipLaunchKernelGGL((k_stream_ref<3, 3>), gr, bl, 0, st, n2, A, B, C, O0, O1, O2); break;
  }
}

template <int NR, int NW>
static void stream_par_t(hipStream_t st, int u, int nt, int nblocks, int64_t n2, const double2 *A, const double2 *B, const double2 *C, double2 *O0, double2 *O1, double2 *O2)
{
  const dim3 gr(nblocks), bl(256);
  switch (u * 10 + nt) {
  case 10: hipLaunchKernelGGL((k_stream_par<NR, NW, 1, false>), gr, bl, 0, st, n2, A, B, C, O0, O1, O2); break;
  case 11: hipLaunchKernelGGL((k_stream_par<NR, NW, 1, true>), gr, bl, 0, st, n2, A, B, C, O0, O1, O2); break;
  case 20: hipLaunchKernelGGL((k_stream_par<NR, NW, 2, false>), gr, bl, 0, st, n2, A, B, C, O0, O1, O2); break;
  case 21: hipLaunchKernelGGL((k_stream_par<NR, NW, 2, true>), gr, bl, 0, st, n2, A, B, C, O0, O1, O2); break;
  case 40: hipLaunchKernelGGL((k_stream_par<NR, NW, 4, false>), gr, bl, 0, st, n2, A, B, C, O0, O1, O2); break;
  case 41: hipLaunchKernelGGL((k_stream_par<NR, NW, 4, true>), gr, bl, 0, st, n2, A, B, C, O0, O1, O2); break;
  case 80: hipLaunchKernelGGL((k_stream_par<NR, NW, 8, false>), gr, bl, 0, st, n2, A, B, C, O0, O1, O2); break;
  default: hipLaunchKernelGGL((k_stream_par<NR, NW, 8, true>), gr, bl, 0, st, n2, A, B, C, O0, O1, O2); break;
  }
}
void launch_stream_par(hipStream_t st, int nr, int nw, int u, int nt, int nblocks, int64_t n2, const double *a, const double *b, const double *c, double *o0, double *o1, double *o2)
{
  const double2 *A = (const double2 *)a, *B = (const double2 *)b, *C = (const double2 *)c;
  double2       *O0 = (double2 *)o0, *O1 = (double2 *)o1, *O2 = (double2 *)o2;
  if (nr == 1 && nw == 1) stream_par_t<1, 1>(st, u, nt, nblocks, n2, A, B, C, O0, O1, O2);
  else if (nr == 2 && nw == 1) stream_par_t<2, 1>(st, u, nt, nblocks, n2, A, B, C, O0, O1, O2);
  else stream_par_t<3, 3>(st, u, nt, nblocks, n2, A, B, C, O0, O1, O2);
}

void launch_cg_pupdate(hipStream_t st, const GridP &g, bool jac, const double *r, double *P0, double *P1, const KspScal *s)
{
#ifdef FL_KBENCH_VARIANTS
  if (jac) hipLaunchKernelGGL(k_cg_pupdate<true>, grid3(g.nx, g.ny, g.nz), blk3(), 0, st, g, r, P0, P1, s);
  else hipLaunchKernelGGL(k_cg_pupdate<false>, grid3(g.nx, g.ny, g.nz), blk3(), 0, st, g, r, P0, P1, s);
#endif
}
int  apply_dot_blocks(const GridP &g) { return ((g.nx + 63) / 64) * ((g.ny + 3) / 4) * g.nz; }
void launch_cg_apply_dot(hipStream_t st, const GridP &g, const double *P0, const double *P1, double *q, double *x, const KspScal *s, double *partial)
{
#ifdef FL_KBENCH_VARIANTS
  hipLaunchKernelGGL(k_cg_apply_dot, grid3(g.nx, g.ny, g.nz), blk3(), 0, st, g, P0, P1, q, x, s, partial);
#endif
}

}  // namespace fl
