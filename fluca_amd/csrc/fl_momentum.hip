// fl_momentum.hip -- the momentum block of the Jacobian, matrix-free:  A = I + dt C - (mu dt / 2 rho) L
//
//   NSFormJacobian_CNLinear_Cart3d_Internal    cnlinearcart3d.c:2930-2941
//   ComputeVelocityLaplacianOperator_Private   cnlinearcart3d.c:425-632      L: one 1-D second-derivative row per axis
//   ComputeConvectionOperator_Private          cnlinearcart3d.c:873-1294     (C v)_c = 1/2 d/dx_d (v_c V0_d + v0interp_c v_d)
//   KSPSolve(abf->kspA, momrhs, vstar)         abfpc.c:72                    -ns_abf_momentum_ksp_type bcgs -pc_type jacobi
//
// The reference assembles A as an AIJ matrix every time step (MatZeroEntries + 24 MatSetValuesStencil per row).  Here
// nothing is assembled: the rows are products of 1-D tables (fl_coeff.cpp: build_axis_momentum) with the face fields V0
// and v0interp, evaluated on the fly.  Velocity vectors are three padded cell arrays (component-major), the twelve face
// fields are kept as padded arrays whose entry (i,j,k) is the LOW face of cell (i,j,k) along the field's axis, so the high
// face is the next entry along that axis (ghost layer = the last face / the periodic image / the neighbour rank's first
// face).  Algorithmic HBM traffic of one application: 3 reads + 3 writes of v, 12 face reads = 144 B per cell.
#include <new>

#include "fl_handle.h"
#include "fl_device.h"

namespace fl {

struct MomP {
  const double *tab[3];  // MOM_NTAB x len, slot-major, LOCAL block of each axis
  int           len[3];
  double        cI, cC, cL;
};

__device__ __forceinline__ double uniform_d(double v)
{
  // v is wave-uniform: keep it in scalar registers
  const int lo = __builtin_amdgcn_readfirstlane(__double2loint(v)), hi = __builtin_amdgcn_readfirstlane(__double2hiint(v));
  return __hiloint2double(hi, lo);
}

struct MTile {
  int  ic, jc, i, j, k0, k1;
  bool own;
};
__device__ __forceinline__ MTile mom_tile(const GridP &g, int tiles_x, int nchunk, int zc)
{
  MTile     t;
  const int b = blockIdx.x, chunk = b % nchunk, tile = b / nchunk;
  t.i   = (tile % tiles_x) * 64 + (threadIdx.x & 63);
  t.j   = (tile / tiles_x) * 4 + (threadIdx.x >> 6);
  t.own = t.i < g.nx && t.j < g.ny;
  t.ic  = min(t.i, g.nx - 1);
  t.jc  = min(t.j, g.ny - 1);
  t.k0  = chunk * zc;
  t.k1  = min(t.k0 + zc, g.nz);
  return t;
}

// y = [1/diag] A x   (x padded with valid ghosts; y padded, or unpadded component-major when OUT == 1).
// OUT == 2 writes diag(A) instead (padded).  DOT: partial slots 0 sum y, 1 y.o (o padded, may be NULL), 2 x.y, 3 y.y.
template <bool DOT, bool JAC, int OUT>
__global__ void __launch_bounds__(256) k_mom_apply(GridP g, MomP m, const double *__restrict__ x, double *__restrict__ y, const double *__restrict__ F, int64_t cs, const double *__restrict__ o,
                                                   const KspScal *__restrict__ s, double *__restrict__ partial, int pstride, int tiles_x, int nchunk, int zc)
{
  __shared__ double red[4 * 4];
  if (s && s->reason != 0) return;
  const MTile t = mom_tile(g, tiles_x, nchunk, zc);
  double      tx[MOM_NTAB], ty[MOM_NTAB];
#pragma unroll
  for (int a = 0; a < MOM_NTAB; ++a) {
    tx[a] = m.tab[0][(int64_t)a * m.len[0] + t.ic];
    ty[a] = uniform_d(m.tab[1][(int64_t)a * m.len[1] + t.jc]);
  }
  const int64_t sx = g.sx, sxy = g.sxy;
  const int64_t fox = t.ic == 0 ? 2 : -2, foy = (t.jc == 0 ? 2 : -2) * sx;
  const int64_t ncell = (int64_t)g.nx * g.ny * g.nz;
  double        acc[4] = {0., 0., 0., 0.};
  for (int k = t.k0; k < t.k1; ++k) {
    double tz[MOM_NTAB];
#pragma unroll
    for (int a = 0; a < MOM_NTAB; ++a) tz[a] = m.tab[2][(int64_t)a * m.len[2] + k];
    const int64_t idx = g.off0 + (int64_t)k * sxy + (int64_t)t.jc * sx + t.ic;
    const int64_t foz = (k == 0 ? 2 : -2) * sxy;
    double        u[3][3][3], uf[3][3], vl[3], vh[3], wl[3][3], wh[3][3];
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      const double *X = x + c * cs + idx;
      const double  uc = X[0];
      u[c][0][0] = X[-1];
      u[c][0][1] = uc;
      u[c][0][2] = X[1];
      u[c][1][0] = X[-sx];
      u[c][1][1] = uc;
      u[c][1][2] = X[sx];
      u[c][2][0] = X[-sxy];
      u[c][2][1] = uc;
      u[c][2][2] = X[sxy];
      uf[c][0]   = X[fox];
      uf[c][1]   = X[foy];
      uf[c][2]   = X[foz];
    }
    const int64_t str[3] = {1, sx, sxy};
#pragma unroll
    for (int d = 0; d < 3; ++d) {
      const double *V = F + d * cs + idx;
      vl[d] = V[0];
      vh[d] = V[str[d]];
#pragma unroll
      for (int c = 0; c < 3; ++c) {
        const double *W = F + (3 + c * 3 + d) * cs + idx;
        wl[c][d] = W[0];
        wh[c][d] = W[str[d]];
      }
    }
    // face interpolants of the face-normal component along each axis ("normal" rule)
    double Glo[3], Ghi[3];
#pragma unroll
    for (int d = 0; d < 3; ++d) {
      const double *tb = d == 0 ? tx : (d == 1 ? ty : tz);
      Glo[d] = tb[11] * u[d][d][0] + tb[12] * u[d][d][1] + tb[13] * u[d][d][2];
      Ghi[d] = tb[17] * u[d][d][0] + tb[18] * u[d][d][1] + tb[19] * u[d][d][2];
    }
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      double conv = 0., lap = 0., dgc = 0., dgl = 0.;
#pragma unroll
      for (int d = 0; d < 3; ++d) {
        const double *tb = d == 0 ? tx : (d == 1 ? ty : tz);
        const int     r = c == d ? 1 : 0;
        double        Ilo, Ihi;
        if (r) {
          Ilo = Glo[d];
          Ihi = Ghi[d];
        } else {
          Ilo = tb[8] * u[c][d][0] + tb[9] * u[c][d][1] + tb[10] * u[c][d][2];
          Ihi = tb[14] * u[c][d][0] + tb[15] * u[c][d][1] + tb[16] * u[c][d][2];
        }
        conv += vl[d] * Ilo + vh[d] * Ihi;              // first term:  v_c V0_d
        conv += wl[c][d] * Glo[d] + wh[c][d] * Ghi[d];  // second term: v0interp_c v_d
        lap += tb[r * 4 + 0] * u[c][d][0] + tb[r * 4 + 1] * u[c][d][1] + tb[r * 4 + 2] * u[c][d][2] + tb[r * 4 + 3] * uf[c][d];
        if (JAC || OUT == 2) {
          dgc += vl[d] * tb[8 + r * 3 + 1] + vh[d] * tb[14 + r * 3 + 1];
          if (r) dgc += wl[c][d] * tb[12] + wh[c][d] * tb[18];
          dgl += tb[r * 4 + 1];
        }
      }
      const double uc = u[c][0][1];
      double       yv = m.cI * uc + m.cC * conv + m.cL * lap;
      if (JAC || OUT == 2) {
        const double dg = m.cI + m.cC * dgc + m.cL * dgl;
        if (OUT == 2) yv = dg;
        else yv = yv / dg;
      }
      if (t.own) {
        if (OUT == 1) y[c * ncell + ((int64_t)k * g.ny + t.j) * g.nx + t.i] = yv;
        else y[c * cs + idx] = yv;
        if (DOT) {
          acc[0] += yv;
          if (o) acc[1] += yv * o[c * cs + idx];
          acc[2] += uc * yv;
          acc[3] += yv * yv;
        }
      }
    }
  }
  if (DOT) {
    block_sum<4>(acc, red);
    if (threadIdx.x == 0)
#pragma unroll
      for (int a = 0; a < 4; ++a) partial[(int64_t)a * pstride + blockIdx.x] = acc[a];
  }
}

// BiCGStab vector updates on three-component padded vectors (interior cells only).
// OP 0: P = R - (omega_old beta) V + beta P
// OP 1: S = R - alpha V
// OP 2: X += alpha P + omega S ; R = S - omega T          slots: 0 R.R  1 R.RP  2 sum R
// OP 3: R = RP = b / diag (b unpadded, component-major)    slots: 0 sum R  1 R.R        (dg NULL: no preconditioner)
template <int OP>
__global__ void __launch_bounds__(256) k_mom_pw(GridP g, int64_t cs, const double *__restrict__ a0, const double *__restrict__ a1, const double *__restrict__ a2, const double *__restrict__ a3, double *__restrict__ w0,
                                                double *__restrict__ w1, const KspScal *__restrict__ s, double *__restrict__ partial, int pstride, int tiles_x, int nchunk, int zc)
{
  __shared__ double red[3 * 4];
  if (OP != 3 && s->reason != 0) return;
  const MTile   t = mom_tile(g, tiles_x, nchunk, zc);
  const double  alpha = s->alpha, omega = s->omega, beta = s->beta, ob = s->omega_old * s->beta;
  const int64_t ncell = (int64_t)g.nx * g.ny * g.nz;
  double        acc[3] = {0., 0., 0.};
  if (t.own)
    for (int k = t.k0; k < t.k1; ++k) {
      const int64_t idx = g.off0 + (int64_t)k * g.sxy + (int64_t)t.j * g.sx + t.i;
#pragma unroll
      for (int c = 0; c < 3; ++c) {
        const int64_t q = c * cs + idx;
        if (OP == 0) {
          w0[q] = a0[q] - ob * a1[q] + beta * w0[q];
        } else if (OP == 1) {
          w0[q] = a0[q] - alpha * a1[q];
        } else if (OP == 2) {
          const double S = a1[q];
          const double rn = S - omega * a2[q];
          w0[q] += alpha * a0[q] + omega * S;
          w1[q] = rn;
          acc[0] += rn * rn;
          acc[1] += rn * a3[q];
          acc[2] += rn;
        } else {
          const double b = a0[c * ncell + ((int64_t)k * g.ny + t.j) * g.nx + t.i];
          const double r = a1 ? b / a1[q] : b;
          w0[q] = r;
          w1[q] = r;
          acc[0] += r;
          acc[1] += r * r;
        }
      }
    }
  if (OP == 2 || OP == 3) {
    block_sum<3>(acc, red);
    if (threadIdx.x == 0)
#pragma unroll
      for (int a = 0; a < 3; ++a) partial[(int64_t)a * pstride + blockIdx.x] = acc[a];
  }
}

// src: x-fastest array of ex*ey*ez entries (ex <= nx+1, ...) -> padded array, entry (i,j,k) at off0 + k sxy + j sx + i
__global__ void __launch_bounds__(256) k_pad_copy_ext(GridP g, const double *__restrict__ src, double *__restrict__ dst, int ex, int ey, int ez)
{
  const int64_t n = (int64_t)ex * ey * ez;
  for (int64_t q = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; q < n; q += (int64_t)gridDim.x * blockDim.x) {
    const int     i = (int)(q % ex);
    const int64_t r = q / ex;
    const int     j = (int)(r % ey), k = (int)(r / ey);
    dst[g.off0 + (int64_t)k * g.sxy + (int64_t)j * g.sx + i] = src[q];
  }
}

}  // namespace fl

using namespace fl;

// ------------------------------------------------------------------------------------------------ handle

struct fl_momentum {
  fl_poisson *p = nullptr;
  MomP        mp;
  void       *tabs[3] = {nullptr, nullptr, nullptr};
  double     *F = nullptr;   // 12 padded face fields: V0[0..2], v0interp[c*3+d] at 3 + c*3 + d
  double     *dg = nullptr;  // diag(A), 3 padded components (valid after set_state)
  double     *vec[8] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
  bool        have_state = false;
  int         tiles_x = 1, tiles_y = 1, nchunk = 1, zc = 1, nblocks = 1;
};

namespace {

int mom_vec(fl_momentum *m, int slot)
{
  if (m->vec[slot]) return 0;
  return fl_dev_alloc(m->p, (void **)&m->vec[slot], sizeof(double) * 3 * m->p->padlen, true);
}

int mom_ghosts(fl_momentum *m, double *v3)
{
  fl_poisson *h = m->p;
  if (!fl_any_ghost_exchange(h)) return 0;
  for (int c = 0; c < 3; ++c) FL_CHK(fl_fill_ghosts(h, v3 + (size_t)c * h->padlen));
  return 0;
}

template <bool DOT, bool JAC, int OUT>
void mom_apply_t(fl_momentum *m, const double *x, double *y, const double *o, const KspScal *s)
{
  fl_poisson *h = m->p;
  hipLaunchKernelGGL((k_mom_apply<DOT, JAC, OUT>), dim3(m->nblocks), dim3(256), 0, h->stream, h->g, m->mp, x, y, m->F, (int64_t)h->padlen, o, s, h->partial, h->partial_stride, m->tiles_x, m->nchunk, m->zc);
}

template <int OP>
void mom_pw(fl_momentum *m, const double *a0, const double *a1, const double *a2, const double *a3, double *w0, double *w1)
{
  fl_poisson *h = m->p;
  hipLaunchKernelGGL((k_mom_pw<OP>), dim3(m->nblocks), dim3(256), 0, h->stream, h->g, (int64_t)h->padlen, a0, a1, a2, a3, w0, w1, h->scal, h->partial, h->partial_stride, m->tiles_x, m->nchunk, m->zc);
}

int mom_init(fl_momentum *m, fl_poisson *h)
{
  m->p = h;
  const GridP &g = h->g;
  if (g.nx < 2 || g.ny < 2 || g.nz < 2) return FL_ERR_SUP;  // the one-sided wall rows reach two cells inwards
  const int64_t lo[3] = {h->dec.lo[0], h->dec.lo[1], h->dec.lo[2]};
  const int     len[3] = {g.nx, g.ny, g.nz};
  for (int d = 0; d < 3; ++d) {
    std::vector<double> tab, loc;
    FL_CHK(build_axis_momentum(h->ax[d], tab));
    const int64_t n = h->ax[d].n;
    loc.resize((size_t)MOM_NTAB * len[d]);
    for (int a = 0; a < MOM_NTAB; ++a)
      for (int i = 0; i < len[d]; ++i) loc[(size_t)a * len[d] + i] = tab[(size_t)a * n + (size_t)(lo[d] + i)];
    FL_HIP(hipMalloc(&m->tabs[d], sizeof(double) * loc.size()));
    FL_HIP(hipMemcpy(m->tabs[d], loc.data(), sizeof(double) * loc.size(), hipMemcpyHostToDevice));
    m->mp.tab[d] = (const double *)m->tabs[d];
    m->mp.len[d] = len[d];
  }
  m->mp.cI = 1.;
  m->mp.cC = 0.;
  m->mp.cL = 0.;
  m->tiles_x = (g.nx + 63) / 64;
  m->tiles_y = (g.ny + 3) / 4;
  const int tiles = m->tiles_x * m->tiles_y;
  int       nchunk = std::max(1, (2048 + tiles / 2) / tiles);
  nchunk     = std::max(1, std::min(std::min(nchunk, std::max(1, g.nz / 8)), g.nz));
  m->zc      = (g.nz + nchunk - 1) / nchunk;
  m->nchunk  = (g.nz + m->zc - 1) / m->zc;
  m->nblocks = tiles * m->nchunk;
  if (m->nblocks > MAX_PARTIAL_BLOCKS) {
    // fewer, taller chunks: the partial-sum buffers hold MAX_PARTIAL_BLOCKS entries per slot
    m->nchunk  = std::max(1, MAX_PARTIAL_BLOCKS / tiles);
    m->zc      = (g.nz + m->nchunk - 1) / m->nchunk;
    m->nchunk  = (g.nz + m->zc - 1) / m->zc;
    m->nblocks = tiles * m->nchunk;
    if (m->nblocks > MAX_PARTIAL_BLOCKS) return FL_ERR_SUP;
  }
  FL_CHK(fl_ensure_partials(h, m->nblocks));
  FL_CHK(fl_dev_alloc(h, (void **)&m->F, sizeof(double) * 12 * h->padlen, true));
  FL_CHK(fl_dev_alloc(h, (void **)&m->dg, sizeof(double) * 3 * h->padlen, true));
  return fl_momentum_set_coefficients(m, 1., 0., 0.);  // A = I until the first set_state (also fills diag(A))
}

}  // namespace

extern "C" int fl_momentum_create(fl_poisson *grid_from, fl_momentum **out)
{
  if (!grid_from || !out) return FL_ERR_ARG_NULL;
  *out = nullptr;
  FL_HIP(hipSetDevice(grid_from->device));
  fl_momentum *m = new (std::nothrow) fl_momentum;
  if (!m) return FL_ERR_MEM;
  const int rc = mom_init(m, grid_from);
  if (rc != 0) {
    fl_momentum_destroy(m);
    return rc;
  }
  *out = m;
  return FL_SUCCESS;
}

extern "C" int fl_momentum_destroy(fl_momentum *m)
{
  if (!m) return FL_SUCCESS;
  if (m->p) {
    (void)hipSetDevice(m->p->device);
    (void)hipStreamSynchronize(m->p->stream);
  }
  for (void *t : m->tabs)
    if (t) (void)hipFree(t);
  if (m->F) (void)hipFree(m->F);
  if (m->dg) (void)hipFree(m->dg);
  for (double *v : m->vec)
    if (v) (void)hipFree(v);
  delete m;
  return FL_SUCCESS;
}

extern "C" int fl_momentum_set_coefficients(fl_momentum *m, double cI, double cC, double cL)
{
  if (!m) return FL_ERR_ARG_NULL;
  m->mp.cI = cI;
  m->mp.cC = cC;
  m->mp.cL = cL;
  fl_poisson *h = m->p;
  FL_HIP(hipSetDevice(h->device));
  mom_apply_t<false, false, 2>(m, m->F, m->dg, nullptr, nullptr);  // x is not used for the diagonal: any valid padded array
  FL_HIP(hipGetLastError());
  return FL_SUCCESS;
}

extern "C" int fl_momentum_set_state(fl_momentum *m, double dt, double rho, double mu, const double *const V0_dev[3], const double *const v0interp_dev[9])
{
  if (!m || !V0_dev || !v0interp_dev) return FL_ERR_ARG_NULL;
  if (!(rho > 0.)) return FL_ERR_ARG_OUTOFRANGE;
  fl_poisson  *h = m->p;
  const GridP &g = h->g;
  FL_HIP(hipSetDevice(h->device));
  for (int d = 0; d < 3; ++d) {
    if (!V0_dev[d]) return FL_ERR_ARG_NULL;
    for (int c = 0; c < 3; ++c)
      if (!v0interp_dev[c * 3 + d]) return FL_ERR_ARG_NULL;
  }
  const int ext[3][3] = {{g.fx, g.ny, g.nz}, {g.nx, g.fy, g.nz}, {g.nx, g.ny, g.fz}};
  for (int f = 0; f < 12; ++f) {
    const int     d = f < 3 ? f : (f - 3) % 3;
    const double *src = f < 3 ? V0_dev[f] : v0interp_dev[f - 3];
    double       *dst = m->F + (size_t)f * h->padlen;
    const int64_t n = (int64_t)ext[d][0] * ext[d][1] * ext[d][2];
    const int     nb = (int)std::min<int64_t>((n + 255) / 256, 8192);
    hipLaunchKernelGGL(k_pad_copy_ext, dim3(std::max(nb, 1)), dim3(256), 0, h->stream, g, src, dst, ext[d][0], ext[d][1], ext[d][2]);
    // high face of the last owned cell: periodic image or the neighbour's first face (the physical last face came with the copy)
    if (fl_any_ghost_exchange(h)) FL_CHK(fl_fill_ghosts(h, dst));
  }
  m->have_state = true;
  return fl_momentum_set_coefficients(m, 1., dt, -0.5 * mu * dt / rho);  // MatScale(A, dt); MatAXPY(A, -mu dt / 2 rho, L); MatShift(A, 1)
}

extern "C" int fl_momentum_apply(fl_momentum *m, const double *v_dev, double *y_dev)
{
  if (!m || !v_dev || !y_dev) return FL_ERR_ARG_NULL;
  if (!m->have_state && m->mp.cC != 0.) return FL_ERR_ARG_WRONGSTATE;
  fl_poisson *h = m->p;
  FL_HIP(hipSetDevice(h->device));
  FL_CHK(mom_vec(m, 7));
  for (int c = 0; c < 3; ++c) launch_pad_copy(h->stream, h->g, v_dev + (size_t)c * h->ncell, m->vec[7] + (size_t)c * h->padlen);
  FL_CHK(mom_ghosts(m, m->vec[7]));
  mom_apply_t<false, false, 1>(m, m->vec[7], y_dev, nullptr, nullptr);
  FL_HIP(hipGetLastError());
  return FL_SUCCESS;
}

extern "C" int fl_momentum_diagonal(fl_momentum *m, double *d_dev)
{
  if (!m || !d_dev) return FL_ERR_ARG_NULL;
  if (!m->have_state && m->mp.cC != 0.) return FL_ERR_ARG_WRONGSTATE;
  fl_poisson *h = m->p;
  FL_HIP(hipSetDevice(h->device));
  mom_apply_t<false, false, 2>(m, m->F, m->dg, nullptr, nullptr);
  for (int c = 0; c < 3; ++c) launch_unpad_copy(h->stream, h->g, m->dg + (size_t)c * h->padlen, d_dev + (size_t)c * h->ncell, nullptr);
  FL_HIP(hipGetLastError());
  return FL_SUCCESS;
}

// KSPSolve(kspA): left-preconditioned BiCGStab (KSPBCGS), zero initial guess, PCJACOBI or PCNONE
extern "C" int fl_momentum_solve(fl_momentum *m, const double *b_dev, double *x_dev, const fl_ksp_opts *opts, fl_ksp_stats *stats)
{
  if (!m || !b_dev || !x_dev || !opts || !stats) return FL_ERR_ARG_NULL;
  if (!m->have_state && m->mp.cC != 0.) return FL_ERR_ARG_WRONGSTATE;
  if (opts->type != FL_KSP_BCGS) return FL_ERR_SUP;
  if (opts->pc != FL_PC_JACOBI && opts->pc != FL_PC_NONE) return FL_ERR_SUP;
  if (opts->norm_type != FL_NORM_PRECONDITIONED) return FL_ERR_SUP;
  if (opts->maxit < 0) return FL_ERR_ARG_OUTOFRANGE;
  fl_poisson *h = m->p;
  FL_HIP(hipSetDevice(h->device));
  std::memset(stats, 0, sizeof(*stats));
  const bool jac = opts->pc == FL_PC_JACOBI;
  for (int a = 0; a < 7; ++a) FL_CHK(mom_vec(m, a));
  double *R = m->vec[0], *RP = m->vec[1], *P = m->vec[2], *V = m->vec[3], *X = m->vec[4], *S = m->vec[5], *T = m->vec[6];
  const int nhist = opts->maxit + 1;
  FL_CHK(fl_ensure_hist(h, nhist));
  FL_CHK(fl_ensure_partials(h, m->nblocks));
  fl_ksp_opts o = *opts;
  o.remove_nullspace = 0;  // A = I + ... is non-singular
  FL_CHK(fl_ksp_begin(h, &o));
  const size_t bytes = sizeof(double) * 3 * h->padlen;
  for (double *v : {P, V, X}) FL_HIP(hipMemsetAsync(v, 0, bytes, h->stream));
  mom_pw<3>(m, b_dev, jac ? m->dg : nullptr, nullptr, nullptr, R, RP);
  FL_CHK(fl_bcgs_fin_step(h, 0, m->nblocks, 3, nhist));
  const int every = o.check_every > 0 ? o.check_every : 4;
  int       it = 0;
  bool      done = false;
  while (!done) {
    const int stop = std::min(o.maxit, it + every);
    for (; it < stop; ++it) {
      mom_pw<0>(m, R, V, nullptr, nullptr, P, nullptr);
      FL_CHK(mom_ghosts(m, P));
      if (jac) mom_apply_t<true, true, 0>(m, P, V, RP, h->scal);
      else mom_apply_t<true, false, 0>(m, P, V, RP, h->scal);
      FL_CHK(fl_bcgs_fin_step(h, 1, m->nblocks, 4, nhist));
      mom_pw<1>(m, R, V, nullptr, nullptr, S, nullptr);
      FL_CHK(mom_ghosts(m, S));
      if (jac) mom_apply_t<true, true, 0>(m, S, T, nullptr, h->scal);
      else mom_apply_t<true, false, 0>(m, S, T, nullptr, h->scal);
      FL_CHK(fl_bcgs_fin_step(h, 3, m->nblocks, 4, nhist));
      mom_pw<2>(m, P, S, T, RP, X, R);
      FL_CHK(fl_bcgs_fin_step(h, 4, m->nblocks, 3, nhist));
    }
    FL_CHK(fl_poll_scal(h));
    if (h->scal_host->reason != 0 || it >= o.maxit) done = true;
  }
  for (int c = 0; c < 3; ++c) launch_unpad_copy(h->stream, h->g, X + (size_t)c * h->padlen, x_dev + (size_t)c * h->ncell, nullptr);
  return fl_ksp_finish(h, &o, stats);
}
