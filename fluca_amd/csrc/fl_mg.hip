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
//   restriction: volume-weighted average of the children's residuals;  prolongation: piecewise constant (child += parent)
//   cycle      : V(nu, nu):  x = smooth(b); r = b - S x; e = V(R r); x += P e; r = b - S x; x += smooth(r)
//   coarsest   : Jacobi-PCG to rtol 1e-2 (at most 200 iterations)
//   outer      : KSPCG, left preconditioning, preconditioned norm ||z||, KSPConvergedDefault, constant null space
//                removed from every preconditioner output
//
// Round-1 shape: every level is a full fl_poisson handle on the fine handle's stream and the cycle is composed from the
// public entry points (apply, Chebyshev solve) plus three small kernels; scalars of the outer CG live on the host (an
// iteration is ~25 ms at 512^3, the round trips per iteration do not matter).
// Several ranks: every level keeps the fine decomposition (block boundaries coincide with coarse faces), so restriction
// and prolongation stay local; an axis is coarsened only while every rank's share stays even and >= 8 cells; the levels
// borrow the fine handle's communicator for their halo exchanges and reductions.
#include <new>

#include "fl_handle.h"
#include "fl_device.h"

namespace fl {

// coarse(I,J,K) = sum over children of wx wy wz * fine(child); w = child extent / parent extent along each axis
__global__ void __launch_bounds__(256) k_mg_restrict(int nxc, int nyc, int nzc, int rx, int ry, int rz, int nxf, int nyf, const double *__restrict__ wx, const double *__restrict__ wy,
                                                     const double *__restrict__ wz, const double *__restrict__ fine, double *__restrict__ coarse)
{
  const int64_t n = (int64_t)nxc * nyc * nzc;
  for (int64_t q = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; q < n; q += (int64_t)gridDim.x * blockDim.x) {
    const int     I = (int)(q % nxc);
    const int64_t t = q / nxc;
    const int     J = (int)(t % nyc), K = (int)(t / nyc);
    double        s = 0.;
    for (int c = 0; c < rz; ++c)
      for (int b = 0; b < ry; ++b)
        for (int a = 0; a < rx; ++a) {
          const int i = I * rx + a, j = J * ry + b, k = K * rz + c;
          s += wx[i] * wy[j] * wz[k] * fine[((int64_t)k * nyf + j) * nxf + i];
        }
    coarse[q] = s;
  }
}

// fine(child) += coarse(parent)
__global__ void __launch_bounds__(256) k_mg_prolong_add(int nxf, int nyf, int nzf, int rx, int ry, int rz, int nxc, int nyc, const double *__restrict__ coarse, double *__restrict__ fine)
{
  const int64_t n = (int64_t)nxf * nyf * nzf;
  for (int64_t q = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; q < n; q += (int64_t)gridDim.x * blockDim.x) {
    const int     i = (int)(q % nxf);
    const int64_t t = q / nxf;
    const int     j = (int)(t % nyf), k = (int)(t / nyf);
    fine[q] += coarse[((int64_t)(k / rz) * nyc + j / ry) * nxc + i / rx];
  }
}

// y = a x + b z (z may be NULL), and optionally partial[block] = sum y*w (w may be NULL -> sum y)
__global__ void __launch_bounds__(256) k_mg_lincomb_dot(int64_t n, double a, const double *x, double b, const double *z, double *y, const double *w, double *partial)
{
  __shared__ double red[4];
  double            v[1] = {0.};
  // 16 B per lane over the even part (hipMalloc'ed arrays are 16-B aligned), the odd tail by thread 0 of block 0
  const int64_t n2 = n / 2;
  for (int64_t q = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; q < n2; q += (int64_t)gridDim.x * blockDim.x) {
    const double2 xv = reinterpret_cast<const double2 *>(x)[q];
    double2       t  = make_double2(a * xv.x, a * xv.y);
    if (z) {
      const double2 zv = reinterpret_cast<const double2 *>(z)[q];
      t.x += b * zv.x;
      t.y += b * zv.y;
    }
    reinterpret_cast<double2 *>(y)[q] = t;
    if (partial) {
      if (w) {
        const double2 wv = reinterpret_cast<const double2 *>(w)[q];
        v[0] += t.x * wv.x + t.y * wv.y;
      } else v[0] += t.x + t.y;
    }
  }
  if ((n & 1) && blockIdx.x == 0 && threadIdx.x == 0) {
    const int64_t q = n - 1;
    const double  t = a * x[q] + (z ? b * z[q] : 0.);
    y[q]            = t;
    v[0] += w ? t * w[q] : t;
  }
  if (partial) {
    block_sum<1>(v, red);
    if (threadIdx.x == 0) partial[blockIdx.x] = v[0];
  }
}

}  // namespace fl

using namespace fl;

struct MgLevel {
  fl_poisson *h = nullptr;  // level 0: the caller's handle (not owned)
  int         r[3] = {1, 1, 1};  // refinement ratio to the NEXT (coarser) level
  double     *w[3] = {nullptr, nullptr, nullptr};  // restriction weights of this level's cells along each axis
  double     *x = nullptr, *b = nullptr, *res = nullptr, *e = nullptr;  // unpadded cell arrays
};

struct fl_mg {
  std::vector<MgLevel> lv;
  double *r = nullptr, *z = nullptr, *p = nullptr, *q = nullptr;  // outer CG, fine level
};

void fl_mg_destroy(fl_poisson *h);

namespace {

constexpr int MG_DOT_BLOCKS = 4096;

int nblk(int64_t n) { return (int)std::max<int64_t>(1, std::min<int64_t>((n / 2 + 255) / 256, MG_DOT_BLOCKS)); }

// y = a x + b z ; returns (optionally) sum(y * w) or sum(y) on the host
int lincomb_dot(fl_poisson *h, int64_t n, double a, const double *x, double b, const double *z, double *y, const double *w, double *result)
{
  const int nb = nblk(n);
  hipLaunchKernelGGL(k_mg_lincomb_dot, dim3(nb), dim3(256), 0, h->stream, n, a, x, b, z, y, w, result ? h->partial : nullptr);
  if (result) {
    launch_reduce(h->stream, h->partial, nb, h->partial_stride, 1, h->sums);
    if (h->multi) FL_CHK(h->comm.allreduce(h->stream, h->sums, NSLOT));
    FL_HIP(hipMemcpyAsync(result, h->sums, sizeof(double), hipMemcpyDeviceToHost, h->stream));
    FL_HIP(hipStreamSynchronize(h->stream));
  }
  return 0;
}

}  // namespace

// constant shift kernel (y -= m)
namespace fl {
__global__ void __launch_bounds__(256) k_mg_shift(int64_t n, double m, double *y)
{
  for (int64_t q = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; q < n; q += (int64_t)gridDim.x * blockDim.x) y[q] -= m;
}
}  // namespace fl

namespace {

int project_constant(fl_poisson *h, int64_t n, double *y)
{
  double s = 0.;
  FL_CHK(lincomb_dot(h, n, 1., y, 0., nullptr, y, nullptr, &s));
  const double N = (double)h->ax[0].n * (double)h->ax[1].n * (double)h->ax[2].n;  // global cell count
  hipLaunchKernelGGL(k_mg_shift, dim3(nblk(n)), dim3(256), 0, h->stream, n, s / N, y);
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
      any |= r[d] == 2;
    }
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
    }
  }
  for (size_t l = 0; l < mg->lv.size(); ++l) {
    MgLevel &L = mg->lv[l];
    if (l > 0) {
      FL_CHK(alloc_cells(L.h, &L.x));
      FL_CHK(alloc_cells(L.h, &L.b));
    }
    FL_CHK(alloc_cells(L.h, &L.res));
    FL_CHK(alloc_cells(L.h, &L.e));
    FL_CHK(fl_ensure_partials(L.h, MG_DOT_BLOCKS));
  }
  for (double **v : {&mg->r, &mg->z, &mg->p, &mg->q}) FL_CHK(alloc_cells(h, v));
  FL_CHK(fl_ensure_partials(h, MG_DOT_BLOCKS));
  return 0;
}

// x = V-cycle(b) on level l, zero initial guess
int vcycle(fl_mg *mg, size_t l, const double *b, double *x, const fl_ksp_opts *o)
{
  MgLevel      &L = mg->lv[l];
  fl_poisson   *h = L.h;
  const int64_t n = h->ncell;
  fl_ksp_stats  st;
  fl_ksp_opts   so;
  fl_ksp_opts_default(&so);
  so.remove_nullspace = o->remove_nullspace;
  if (l + 1 == mg->lv.size()) {
    so.type  = FL_KSP_CG;
    so.pc    = FL_PC_JACOBI;
    so.rtol  = 1e-2;
    so.maxit = 200;
    return fl_poisson_solve(h, b, x, &so, &st);
  }
  so.type      = FL_KSP_CHEBYSHEV;
  so.pc        = FL_PC_JACOBI;
  so.norm_type = FL_NORM_NONE;
  so.maxit     = o->mg_smooth_its > 0 ? o->mg_smooth_its : 3;
  so.check_every = -1;  // smoother: no convergence test, no host round trip
  MgLevel &C   = mg->lv[l + 1];
  FL_CHK(fl_poisson_solve(h, b, x, &so, &st));                                       // x = smooth(b)
  FL_CHK(fl_residual(h, x, b, L.res));                                               // r = b - S x
  {
    const GridP &gf = h->g, &gc = C.h->g;
    hipLaunchKernelGGL(k_mg_restrict, dim3(nblk(C.h->ncell)), dim3(256), 0, h->stream, gc.nx, gc.ny, gc.nz, L.r[0], L.r[1], L.r[2], gf.nx, gf.ny, L.w[0], L.w[1], L.w[2], L.res, C.b);
  }
  FL_CHK(vcycle(mg, l + 1, C.b, C.x, o));                                            // e_c = V(R r)
  {
    const GridP &gf = h->g, &gc = C.h->g;
    hipLaunchKernelGGL(k_mg_prolong_add, dim3(nblk(n)), dim3(256), 0, h->stream, gf.nx, gf.ny, gf.nz, L.r[0], L.r[1], L.r[2], gc.nx, gc.ny, C.x, x);  // x += P e_c
  }
  FL_CHK(fl_residual(h, x, b, L.res));                                               // r = b - S x
  FL_CHK(fl_poisson_solve(h, L.res, L.e, &so, &st));                                 // e = smooth(r)
  FL_CHK(lincomb_dot(h, n, 1., x, 1., L.e, x, nullptr, nullptr));                    // x += e
  return 0;
}

}  // namespace

void fl_mg_destroy(fl_poisson *h)
{
  fl_mg *mg = h->mg;
  if (!mg) return;
  for (size_t l = 0; l < mg->lv.size(); ++l) {
    MgLevel &L = mg->lv[l];
    for (double *p : {L.x, L.b, L.res, L.e, L.w[0], L.w[1], L.w[2]})
      if (p) (void)hipFree(p);
    if (l > 0 && L.h) fl_poisson_destroy(L.h);
  }
  for (double *p : {mg->r, mg->z, mg->p, mg->q})
    if (p) (void)hipFree(p);
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
  if (h->mg && o->mg_levels > 0 && (int)h->mg->lv.size() != std::min<int>(o->mg_levels, (int)h->mg->lv.size()) ) fl_mg_destroy(h);
  if (!h->mg) FL_CHK(mg_build(h, o->mg_levels));
  fl_mg        *mg = h->mg;
  const int64_t n = h->ncell;
  const bool    ns = o->remove_nullspace != 0;
  const bool    pnorm = o->norm_type == FL_NORM_PRECONDITIONED;
  // the level-0 smoother runs through fl_poisson_solve on this very handle and uses h->ev0 / h->ev1 for its own timing
  hipEvent_t e0, e1;
  FL_HIP(hipEventCreate(&e0));
  FL_HIP(hipEventCreate(&e1));
  FL_HIP(hipEventRecord(e0, h->stream));
  double *r = mg->r, *z = mg->z, *p = mg->p, *q = mg->q;
  double  rz = 0., rz_old = 1., dp = 0., pq = 0., rr = 0.;
  std::vector<double> hist;
  FL_CHK(lincomb_dot(h, n, 1., b, 0., nullptr, r, r, &rr));               // r = b (x = 0)
  FL_CHK(lincomb_dot(h, n, 0., b, 0., nullptr, x, nullptr, nullptr));     // x = 0
  FL_CHK(vcycle(mg, 0, r, z, o));                                         // z = M^-1 r
  if (ns) FL_CHK(project_constant(h, n, z));
  if (pnorm) {
    FL_CHK(lincomb_dot(h, n, 1., z, 0., nullptr, z, z, &dp));             // ||z||^2
    dp = std::sqrt(dp);
  } else dp = std::sqrt(rr);
  FL_CHK(lincomb_dot(h, n, 1., z, 0., nullptr, p, r, &rz));               // p = z ; rz = r.z
  const double rnorm0 = dp, ttol = std::max(o->rtol * dp, o->atol);
  hist.push_back(dp);
  int  it = 0, reason = 0;
  auto converged = [&](double d) {
    if (std::isnan(d) || std::isinf(d)) return (int)FL_DIVERGED_NANORINF;
    if (d <= ttol) return d < o->atol ? (int)FL_CONVERGED_ATOL : (int)FL_CONVERGED_RTOL;
    if (d >= o->dtol * rnorm0) return (int)FL_DIVERGED_DTOL;
    return 0;
  };
  reason = converged(dp);
  if (!reason && o->maxit <= 0) reason = FL_DIVERGED_ITS;
  while (!reason) {
    FL_CHK(fl_poisson_apply(h, p, q));                                     // q = S p
    FL_CHK(lincomb_dot(h, n, 1., q, 0., nullptr, q, p, &pq));              // p.q
    if (!(pq > 0.)) {
      reason = std::isnan(pq) ? FL_DIVERGED_NANORINF : FL_DIVERGED_INDEFINITE_MAT;
      break;
    }
    const double alpha = rz / pq;
    FL_CHK(lincomb_dot(h, n, 1., x, alpha, p, x, nullptr, nullptr));       // x += alpha p
    FL_CHK(lincomb_dot(h, n, 1., r, -alpha, q, r, r, &rr));                // r -= alpha q ; r.r
    FL_CHK(vcycle(mg, 0, r, z, o));                                        // z = M^-1 r
    if (ns) FL_CHK(project_constant(h, n, z));
    rz_old = rz;
    if (pnorm) {
      double zz = 0.;
      FL_CHK(lincomb_dot(h, n, 1., z, 0., nullptr, z, z, &zz));
      dp = std::sqrt(zz);
    } else dp = std::sqrt(rr);
    FL_CHK(lincomb_dot(h, n, 1., z, 0., nullptr, z, r, &rz));              // r.z
    ++it;
    hist.push_back(dp);
    reason = converged(dp);
    if (!reason && it >= o->maxit) reason = FL_DIVERGED_ITS;
    if (!reason && !(rz > 0.)) reason = std::isnan(rz) ? FL_DIVERGED_NANORINF : FL_DIVERGED_INDEFINITE_PC;
    if (reason) break;
    const double beta = rz / rz_old;
    FL_CHK(lincomb_dot(h, n, 1., z, beta, p, p, nullptr, nullptr));        // p = z + beta p
  }
  if (ns) FL_CHK(project_constant(h, n, x));
  FL_HIP(hipEventRecord(e1, h->stream));
  FL_HIP(hipStreamSynchronize(h->stream));
  float ms = 0.f;
  FL_HIP(hipEventElapsedTime(&ms, e0, e1));
  (void)hipEventDestroy(e0);
  (void)hipEventDestroy(e1);
  st->iters   = it;
  st->reason  = reason;
  st->rnorm0  = rnorm0;
  st->rnorm   = dp;
  st->seconds = ms * 1e-3;
  if (o->history && o->nhistory > 0) {
    const int m = std::min<int>(o->nhistory, (int)hist.size());
    std::memcpy(o->history, hist.data(), sizeof(double) * m);
  }
  return 0;
}
