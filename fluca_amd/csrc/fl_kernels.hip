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
#include "fl_internal.h"

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
__global__ void k_wrap_ghosts(GridP g, double *__restrict__ v, int axis)
{
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

// boundary face plane of V (axis, side) = coeff * pb     (INSERT_VALUES)
__global__ void k_gst_bc(GridP g, const double *__restrict__ pb, double *__restrict__ V, int axis, int side, double coeff)
{
  const int a = blockIdx.x * 64 + threadIdx.x, b = blockIdx.y * 4 + threadIdx.y;
  int       na, nb;
  if (axis == 0) { na = g.ny; nb = g.nz; }
  else if (axis == 1) { na = g.nx; nb = g.nz; }
  else { na = g.nx; nb = g.ny; }
  if (a >= na || b >= nb) return;
  const int f = side ? (axis == 0 ? g.fx : (axis == 1 ? g.fy : g.fz)) - 1 : 0;
  int64_t   p = axis == 0 ? ((int64_t)b * g.ny + a) * g.fx + f : (axis == 1 ? ((int64_t)b * g.fy + f) * g.nx + a : ((int64_t)f * g.ny + b) * g.nx + a);
  V[p]        = coeff * pb[(int64_t)b * na + a];
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
  if (threadIdx.x < NSLOT) sums[threadIdx.x] = threadIdx.x < nslot ? out[threadIdx.x] : 0.;
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

// mode 0: after k_cg_init.  mode 1: after k_cg_A (alpha).  mode 2: after k_cg_B (beta, convergence).
// If nblocks > 0 the partials are reduced here (single rank); else `sums` already holds the (all-reduced) sums.
__global__ void __launch_bounds__(256) k_cg_fin(int mode, const double *__restrict__ partial, int nblocks, int stride, const double *__restrict__ sums, KspScal *__restrict__ s, double *__restrict__ hist, int nhist)
{
  __shared__ double out[NSLOT], red[NSLOT * 4];
  if (s->reason != 0) return;
  if (nblocks > 0) reduce_partials(partial, nblocks, stride, mode == 1 ? 1 : 5, out, red);
  else {
    if (threadIdx.x < NSLOT) out[threadIdx.x] = sums[threadIdx.x];
    __syncthreads();
  }
  if (threadIdx.x != 0) return;
  if (mode == 1) {
    s->pending_x = 0;
    s->cur ^= 1;
    const double pq = out[0];
    s->pq           = pq;
    if (!(pq > 0.)) {
      s->reason = isnan(pq) ? FL_DIVERGED_NANORINF : FL_DIVERGED_INDEFINITE_MAT;
      return;
    }
    s->alpha = s->rz / pq;
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
    s->pending_x = 1;
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

// ------------------------------------------------------------------------------------------------ CG: init / B / flush

// work split of the streaming kernels: one wave per 128-cell row segment, grid-stride over segments
struct SegIter {
  int64_t seg, nseg, stride;
  int     nxs;
};

// r_pad = b ; five partial sums.  b unpadded.
template <bool JAC>
__global__ void __launch_bounds__(256) k_cg_init(GridP g, const double *__restrict__ b, double *__restrict__ r, double *__restrict__ partial, int stride)
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
#pragma unroll
    for (int c = 0; c < 2; ++c) {
      const int i = xs * 128 + 2 * lane + c;
      if (i < g.nx) {
        const double rv = b[((int64_t)k * g.ny + j) * g.nx + i];
        r[pidx(g, i, j, k)] = rv;
        const double z0 = JAC ? rv / (g.sc[0][i] + dyz) : rv;
        acc[0] += rv * z0;
        acc[1] += z0 * z0;
        acc[2] += z0;
        acc[3] += rv;
        acc[4] += rv * rv;
      }
    }
  }
  block_sum<5>(acc, red);
  if (threadIdx.x == 0)
#pragma unroll
    for (int a = 0; a < 5; ++a) partial[(int64_t)a * stride + blockIdx.x] = acc[a];
}

// r -= alpha q ; partial sums of the new r.  24 B/cell.
template <bool JAC>
__global__ void __launch_bounds__(256) k_cg_B(GridP g, const double *__restrict__ q, double *__restrict__ r, const KspScal *__restrict__ s, double *__restrict__ partial, int stride)
{
  __shared__ double red[5 * 4];
  if (s->reason != 0) return;
  const double  alpha = s->alpha;
  const int     lane  = threadIdx.x & 63;
  const int     nxs   = (g.nx + 127) / 128;
  const int64_t nseg  = (int64_t)nxs * g.ny * g.nz;
  const int64_t step  = (int64_t)gridDim.x * 4;
  double        acc[5] = {0., 0., 0., 0., 0.};
  for (int64_t seg0 = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6); seg0 < nseg; seg0 += 2 * step) {
    // two independent segments per trip: more bytes in flight
    double2 qv[2], rv[2];
    int64_t off[2];
    int     ii[2], jj[2], kk[2];
    bool    ok[2];
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const int64_t seg = seg0 + u * step;
      ok[u]             = seg < nseg;
      const int64_t sg  = ok[u] ? seg : 0;
      const int     xs  = (int)(sg % nxs);
      const int64_t R   = sg / nxs;
      jj[u]             = (int)(R % g.ny);
      kk[u]             = (int)(R / g.ny);
      ii[u]             = xs * 128 + 2 * lane;
      ok[u]             = ok[u] && ii[u] < g.nx;
      off[u]            = pidx(g, ii[u], jj[u], kk[u]);
      if (ok[u]) {
        qv[u] = *reinterpret_cast<const double2 *>(q + off[u]);
        rv[u] = *reinterpret_cast<const double2 *>(r + off[u]);
      }
    }
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      if (!ok[u]) continue;
      const double dyz = g.sc[1][jj[u]] + g.sc[2][kk[u]];
      double2      rn;
      rn.x = rv[u].x - alpha * qv[u].x;
      rn.y = rv[u].y - alpha * qv[u].y;
      if (ii[u] + 1 < g.nx) {
        *reinterpret_cast<double2 *>(r + off[u]) = rn;
        const double z1 = JAC ? rn.y / (g.sc[0][ii[u] + 1] + dyz) : rn.y;
        acc[0] += rn.y * z1;
        acc[1] += z1 * z1;
        acc[2] += z1;
        acc[3] += rn.y;
        acc[4] += rn.y * rn.y;
      } else {
        r[off[u]] = rn.x;  // odd nx: the pair's second entry is the ghost column, leave it alone
      }
      const double z0 = JAC ? rn.x / (g.sc[0][ii[u]] + dyz) : rn.x;
      acc[0] += rn.x * z0;
      acc[1] += z0 * z0;
      acc[2] += z0;
      acc[3] += rn.x;
      acc[4] += rn.x * rn.x;
    }
  }
  block_sum<5>(acc, red);
  if (threadIdx.x == 0)
#pragma unroll
    for (int a = 0; a < 5; ++a) partial[(int64_t)a * stride + blockIdx.x] = acc[a];
}

// the x-update still owed when the iteration stops:  x += alpha p   (p = the current direction)
__global__ void __launch_bounds__(256) k_cg_flush(GridP g, const double *__restrict__ P0, const double *__restrict__ P1, double *__restrict__ x, const KspScal *__restrict__ s)
{
  if (!s->pending_x) return;
  const double *p     = s->cur ? P1 : P0;
  const double  alpha = s->alpha;
  const int     lane  = threadIdx.x & 63;
  const int     nxs   = (g.nx + 127) / 128;
  const int64_t nseg  = (int64_t)nxs * g.ny * g.nz;
  for (int64_t seg = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6); seg < nseg; seg += (int64_t)gridDim.x * 4) {
    const int     xs = (int)(seg % nxs);
    const int64_t R  = seg / nxs;
    const int     j = (int)(R % g.ny), k = (int)(R / g.ny);
#pragma unroll
    for (int c = 0; c < 2; ++c) {
      const int i = xs * 128 + 2 * lane + c;
      if (i < g.nx) {
        const int64_t o = pidx(g, i, j, k);
        x[o] += alpha * p[o];
      }
    }
  }
}

// ------------------------------------------------------------------------------------------------ CG: the fused stencil kernel

template <int RY>
struct TileA {
  static constexpr int TX = 128, TY = 4 * RY, LX = TX + 4, LY = TY + 2;
};

template <int RY, bool JAC>
__global__ void __launch_bounds__(256, 2) k_cg_A(GridP g, const double *__restrict__ r, double *__restrict__ P0, double *__restrict__ P1, double *__restrict__ q, double *__restrict__ x, const KspScal *__restrict__ s,
                                                  double *__restrict__ partial, int nchunk, int zc, int tiles_x)
{
  using T               = TileA<RY>;
  constexpr int TX = T::TX, TY = T::TY, LX = T::LX, LY = T::LY;
  __shared__ __attribute__((aligned(16))) double lds[2][LY][LX];
  __shared__ double                              red[4];
  if (s->reason != 0) return;

  const int     cur        = s->cur;
  const double *pold       = cur ? P1 : P0;
  double       *pnew       = cur ? P0 : P1;
  const double  beta       = s->beta;
  const double  zs         = s->zshift;
  const double  alpha_prev = s->alpha;
  const bool    pend       = s->pending_x != 0;

  const int b     = blockIdx.x;
  const int chunk = b % nchunk, tile = b / nchunk;
  const int i0 = (tile % tiles_x) * TX, j0 = (tile / tiles_x) * TY;
  const int k0 = chunk * zc, k1 = min(k0 + zc, g.nz);
  if (k0 >= k1) return;
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int i  = i0 + 2 * lane;  // first of this lane's two cells
  const int jb = j0 + w * RY;    // first of this wave's RY rows

  // column state ------------------------------------------------------------------------------------------------
  const bool act  = i <= g.nx;      // pair holds at least one active (owned or high-ghost) cell -> load it
  const bool own0 = i < g.nx, own1 = i + 1 < g.nx;
  const bool gh0 = i == g.nx, gh1 = i + 1 == g.nx;  // high ghost column inside the tile
  const int  ic0 = min(i, g.nx), ic1 = min(i + 1, g.nx);
  const double xl0 = g.sl[0][ic0], xc0 = g.sc[0][ic0], xh0 = g.sh[0][ic0];
  const double xl1 = g.sl[0][ic1], xc1 = g.sc[0][ic1], xh1 = g.sh[0][ic1];

  // halo cells of this thread -----------------------------------------------------------------------------------
  // A: rows jj = -1 (tid < 128) / TY (tid >= 128), column ii = tid & 127
  const int  hAi = i0 + (tid & 127), hAj = j0 + (tid < 128 ? -1 : TY);
  const bool hAok = hAi < g.nx && hAj <= g.ny;
  const bool hAgh = hAok && (hAj == -1 || hAj == g.ny);
  // B: columns ii = -1 / TX for tid < 2*TY: jj = tid >> 1
  const int  hBi = i0 + ((tid & 1) ? TX : -1), hBj = j0 + (tid >> 1);
  const bool hBok = tid < 2 * TY && hBj < g.ny && hBi <= g.nx;
  const bool hBgh = hBok && (hBi == -1 || hBi == g.nx);
  const double hAdxy = hAok ? g.sc[0][hAi] + g.sc[1][hAj] : 1.;
  const double hBdxy = hBok ? g.sc[0][hBi] + g.sc[1][hBj] : 1.;
  const int hAr = (tid < 128 ? 0 : TY + 1), hAc = (tid & 127) + 2;
  const int hBr = (tid >> 1) + 1, hBc = (tid & 1) ? TX + 2 : 1;

  double2 pprev[RY], pcur[RY], pnext[RY];
  double2 ra[RY], pa[RY], xa[RY];  // raw values of the plane being processed
  double  hrA = 0., hpA = 0., hrB = 0., hpB = 0.;
  double  dot = 0.;
#pragma unroll
  for (int m = 0; m < RY; ++m) pprev[m] = pcur[m] = pnext[m] = ra[m] = pa[m] = xa[m] = make_double2(0., 0.);

  // plane loader (issued one plane ahead of its use)
  auto load_plane = [&](int kk, double2 (&rr)[RY], double2 (&pp)[RY], double2 (&xx)[RY], double &r_a, double &p_a, double &r_b, double &p_b) {
    const bool pown = kk >= k0 && kk < k1;
#pragma unroll
    for (int m = 0; m < RY; ++m) {
      const int j = jb + m;
      if (act && j <= g.ny) {
        const int64_t o = pidx(g, i, j, kk);
        rr[m]           = *reinterpret_cast<const double2 *>(r + o);
        pp[m]           = *reinterpret_cast<const double2 *>(pold + o);
        if (pend && pown && j < g.ny) xx[m] = *reinterpret_cast<const double2 *>(x + o);
      }
    }
    if (pown) {
      if (hAok) {
        const int64_t o = pidx(g, hAi, hAj, kk);
        r_a             = r[o];
        p_a             = pold[o];
      }
      if (hBok) {
        const int64_t o = pidx(g, hBi, hBj, kk);
        r_b             = r[o];
        p_b             = pold[o];
      }
    }
  };

  load_plane(k0 - 1, ra, pa, xa, hrA, hpA, hrB, hpB);

  for (int kk = k0 - 1; kk <= k1; ++kk) {
    double2 rb[RY], pb[RY], xb[RY];
    double  nrA = 0., npA = 0., nrB = 0., npB = 0.;
#pragma unroll
    for (int m = 0; m < RY; ++m) rb[m] = pb[m] = xb[m] = make_double2(0., 0.);
    if (kk + 1 <= k1) load_plane(kk + 1, rb, pb, xb, nrA, npA, nrB, npB);

    const bool   pown = kk >= k0 && kk < k1;          // plane owned by this chunk: its p', x are stored here
    const bool   pgh  = kk == -1 || kk == g.nz;        // z-ghost plane: p' of the owned columns is stored too
    const double dz   = g.sc[2][kk];
    const int    buf  = kk & 1;

    // p' of plane kk ------------------------------------------------------------------------------------------------
#pragma unroll
    for (int m = 0; m < RY; ++m) {
      const int j = jb + m;
      double2   pn = make_double2(0., 0.);
      if (act && j <= g.ny) {
        const double dy = g.sc[1][j] + dz;
        const double z0 = JAC ? ra[m].x / (xc0 + dy) : ra[m].x;
        const double z1 = JAC ? ra[m].y / (xc1 + dy) : ra[m].y;
        pn.x            = (z0 - zs) + beta * pa[m].x;
        pn.y            = (z1 - zs) + beta * pa[m].y;
        const int64_t o = pidx(g, i, j, kk);
        const bool    rown = j < g.ny;
        if (pown || (pgh && rown)) {
          // store the owned cells, and (owned planes only) the high ghost row / column living inside the tile
          const bool st0 = rown ? (own0 || (gh0 && pown)) : (own0 && pown);
          const bool st1 = rown ? (own1 || (gh1 && pown)) : (own1 && pown);
          if (st0 && st1) *reinterpret_cast<double2 *>(pnew + o) = pn;
          else if (st0) pnew[o] = pn.x;
          else if (st1) pnew[o + 1] = pn.y;
        }
        if (pend && pown && rown) {
          double2 xn;
          xn.x = xa[m].x + alpha_prev * pa[m].x;
          xn.y = xa[m].y + alpha_prev * pa[m].y;
          if (own1) *reinterpret_cast<double2 *>(x + o) = xn;
          else if (own0) x[o] = xn.x;
        }
      }
      pnext[m] = pn;
    }
    double hnA = 0., hnB = 0.;
    if (pown) {
      if (hAok) {
        const double z = JAC ? hrA / (hAdxy + dz) : hrA;
        hnA            = (z - zs) + beta * hpA;
        if (hAgh) pnew[pidx(g, hAi, hAj, kk)] = hnA;
      }
      if (hBok) {
        const double z = JAC ? hrB / (hBdxy + dz) : hrB;
        hnB            = (z - zs) + beta * hpB;
        if (hBgh) pnew[pidx(g, hBi, hBj, kk)] = hnB;
      }
    }

    // q of plane kc = kk-1 (its in-plane neighbours were staged in lds[kc&1] one trip ago) -------------------------------
    const int kc = kk - 1;
    if (kc >= k0) {
      const double zl = g.sl[2][kc], zh = g.sh[2][kc], dzc = g.sc[2][kc];
      const int    bc = kc & 1;
#pragma unroll
      for (int m = 0; m < RY; ++m) {
        const int j = jb + m;
        if (own0 && j < g.ny) {
          const int    lr = w * RY + m + 1, lc = 2 * lane + 2;
          const double west = lds[bc][lr][lc - 1], east = lds[bc][lr][lc + 2];
          double2      south, north;
          if (m > 0) south = pcur[m - 1];
          else south = *reinterpret_cast<const double2 *>(&lds[bc][lr - 1][lc]);
          if (m < RY - 1) north = pcur[m + 1];
          else north = *reinterpret_cast<const double2 *>(&lds[bc][lr + 1][lc]);
          const double yl = g.sl[1][j], yh = g.sh[1][j], dyc = g.sc[1][j] + dzc;
          double2      qq;
          qq.x = (xc0 + dyc) * pcur[m].x + xl0 * west + xh0 * pcur[m].y + yl * south.x + yh * north.x + zl * pprev[m].x + zh * pnext[m].x;
          qq.y = (xc1 + dyc) * pcur[m].y + xl1 * pcur[m].x + xh1 * east + yl * south.y + yh * north.y + zl * pprev[m].y + zh * pnext[m].y;
          const int64_t o = pidx(g, i, j, kc);
          dot += pcur[m].x * qq.x;
          if (own1) {
            *reinterpret_cast<double2 *>(q + o) = qq;
            dot += pcur[m].y * qq.y;
          } else {
            q[o] = qq.x;
          }
        }
      }
    }

    // stage plane kk for the next trip ------------------------------------------------------------------------------------
    if (pown) {
#pragma unroll
      for (int m = 0; m < RY; ++m) *reinterpret_cast<double2 *>(&lds[buf][w * RY + m + 1][2 * lane + 2]) = pnext[m];
      if (hAok) lds[buf][hAr][hAc] = hnA;
      if (hBok) lds[buf][hBr][hBc] = hnB;
    }
    __syncthreads();

#pragma unroll
    for (int m = 0; m < RY; ++m) {
      pprev[m] = pcur[m];
      pcur[m]  = pnext[m];
      ra[m]    = rb[m];
      pa[m]    = pb[m];
      xa[m]    = xb[m];
    }
    hrA = nrA; hpA = npA; hrB = nrB; hpB = npB;
  }

  dot = wave_sum(dot);
  if (lane == 0) red[w] = dot;
  __syncthreads();
  if (tid == 0) partial[b] = (red[0] + red[1]) + (red[2] + red[3]);
}

// ------------------------------------------------------------------------------------------------ unfused CG pieces (variant 1)

// p = (r/diag - mean) + beta p on the owned cells
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

// ------------------------------------------------------------------------------------------------ launch helpers

static inline dim3 grid3(int nx, int ny, int nz) { return dim3((nx + 63) / 64, (ny + 3) / 4, nz); }
static inline dim3 blk3() { return dim3(64, 4, 1); }

}  // namespace fl

// ================================================================================================ host-callable launchers
// (kept in this translation unit so that only hipcc sees <<< >>>)

namespace fl {

void launch_pad_copy(hipStream_t st, const GridP &g, const double *src, double *dst) { hipLaunchKernelGGL(k_pad_copy, grid3(g.nx, g.ny, g.nz), blk3(), 0, st, g, src, dst); }
void launch_unpad_copy(hipStream_t st, const GridP &g, const double *src, double *dst, const double *shift) { hipLaunchKernelGGL(k_unpad_copy, grid3(g.nx, g.ny, g.nz), blk3(), 0, st, g, src, dst, shift); }
void launch_wrap(hipStream_t st, const GridP &g, double *v, int axis)
{
  const int na = axis == 0 ? g.ny : g.nx, nb = axis == 2 ? g.ny : g.nz;
  hipLaunchKernelGGL(k_wrap_ghosts, grid3(na, nb, 1), blk3(), 0, st, g, v, axis);
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
  const int lx = axis == 0 ? g.fx : g.nx, ly = axis == 1 ? g.fy : g.ny, lz = axis == 2 ? g.fz : g.nz;
  if (lz > 0) hipLaunchKernelGGL(k_project_faces, grid3(lx, ly, lz), blk3(), 0, st, g, p, V, axis);
}
void launch_project_cells(hipStream_t st, const GridP &g, const double *p, double *v, int axis) { hipLaunchKernelGGL(k_project_cells, grid3(g.nx, g.ny, g.nz), blk3(), 0, st, g, p, v, axis); }
void launch_gst_bc(hipStream_t st, const GridP &g, const double *pb, double *V, int axis, int side, double coeff)
{
  const int na = axis == 0 ? g.ny : g.nx, nb = axis == 2 ? g.ny : g.nz;
  hipLaunchKernelGGL(k_gst_bc, grid3(na, nb, 1), blk3(), 0, st, g, pb, V, axis, side, coeff);
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

void launch_cg_init(hipStream_t st, const GridP &g, bool jac, const double *b, double *r, double *partial, int stride, int nblocks)
{
  if (jac) hipLaunchKernelGGL(k_cg_init<true>, dim3(nblocks), dim3(256), 0, st, g, b, r, partial, stride);
  else hipLaunchKernelGGL(k_cg_init<false>, dim3(nblocks), dim3(256), 0, st, g, b, r, partial, stride);
}
void launch_cg_B(hipStream_t st, const GridP &g, bool jac, const double *q, double *r, const KspScal *s, double *partial, int stride, int nblocks)
{
  if (jac) hipLaunchKernelGGL(k_cg_B<true>, dim3(nblocks), dim3(256), 0, st, g, q, r, s, partial, stride);
  else hipLaunchKernelGGL(k_cg_B<false>, dim3(nblocks), dim3(256), 0, st, g, q, r, s, partial, stride);
}
void launch_cg_flush(hipStream_t st, const GridP &g, const double *P0, const double *P1, double *x, const KspScal *s, int nblocks) { hipLaunchKernelGGL(k_cg_flush, dim3(nblocks), dim3(256), 0, st, g, P0, P1, x, s); }

// tiling of k_cg_A: returns the number of blocks
struct PlanA {
  int ry, tiles_x, tiles_y, nchunk, zc, nblocks;
};
PlanA plan_cg_A(const GridP &g, int ry_force, int nchunk_force)
{
  PlanA p;
  p.ry = ry_force > 0 ? ry_force : (g.ny >= 256 ? 4 : (g.ny >= 64 ? 2 : 1));
  p.tiles_x = (g.nx + 127) / 128;
  p.tiles_y = (g.ny + 4 * p.ry - 1) / (4 * p.ry);
  const int tiles = p.tiles_x * p.tiles_y;
  int       nchunk = nchunk_force > 0 ? nchunk_force : std::max(1, (512 + tiles / 2) / tiles);  // ~2 resident blocks per CU
  nchunk           = std::min(nchunk, std::max(1, g.nz / 8));                                    // keep the 2-plane chunk prologue <= 25 %
  nchunk           = std::max(1, std::min(nchunk, g.nz));
  p.zc             = (g.nz + nchunk - 1) / nchunk;
  p.nchunk         = (g.nz + p.zc - 1) / p.zc;
  p.nblocks        = tiles * p.nchunk;
  return p;
}

template <int RY>
static void launch_cg_A_ry(hipStream_t st, const GridP &g, bool jac, const PlanA &p, const double *r, double *P0, double *P1, double *q, double *x, const KspScal *s, double *partial)
{
  if (jac) hipLaunchKernelGGL((k_cg_A<RY, true>), dim3(p.nblocks), dim3(256), 0, st, g, r, P0, P1, q, x, s, partial, p.nchunk, p.zc, p.tiles_x);
  else hipLaunchKernelGGL((k_cg_A<RY, false>), dim3(p.nblocks), dim3(256), 0, st, g, r, P0, P1, q, x, s, partial, p.nchunk, p.zc, p.tiles_x);
}
void launch_cg_A(hipStream_t st, const GridP &g, bool jac, const PlanA &p, const double *r, double *P0, double *P1, double *q, double *x, const KspScal *s, double *partial)
{
  switch (p.ry) {
  case 4: launch_cg_A_ry<4>(st, g, jac, p, r, P0, P1, q, x, s, partial); break;
  case 2: launch_cg_A_ry<2>(st, g, jac, p, r, P0, P1, q, x, s, partial); break;
  default: launch_cg_A_ry<1>(st, g, jac, p, r, P0, P1, q, x, s, partial); break;
  }
}

void launch_cg_pupdate(hipStream_t st, const GridP &g, bool jac, const double *r, double *P0, double *P1, const KspScal *s)
{
  if (jac) hipLaunchKernelGGL(k_cg_pupdate<true>, grid3(g.nx, g.ny, g.nz), blk3(), 0, st, g, r, P0, P1, s);
  else hipLaunchKernelGGL(k_cg_pupdate<false>, grid3(g.nx, g.ny, g.nz), blk3(), 0, st, g, r, P0, P1, s);
}
int  apply_dot_blocks(const GridP &g) { return ((g.nx + 63) / 64) * ((g.ny + 3) / 4) * g.nz; }
void launch_cg_apply_dot(hipStream_t st, const GridP &g, const double *P0, const double *P1, double *q, double *x, const KspScal *s, double *partial)
{
  hipLaunchKernelGGL(k_cg_apply_dot, grid3(g.nx, g.ny, g.nz), blk3(), 0, st, g, P0, P1, q, x, s, partial);
}

}  // namespace fl
