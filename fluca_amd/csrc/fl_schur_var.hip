// fl_schur_var.hip -- the Schur complement of PCABF with schurainv = DIAG / ROWSUM (fluca/src/ns/utils/abfpc/abfpc.c:151-171) in ONE pass.
//
// With a = diag(A) or A 1 (per velocity component, cell-centred) and -R = (-T)(kappa G) + kappa Gst,
//     S p = D ((-T) a^-1 kappa G - (-R)) p = div F ,   F_d(f) = -kappa [ (Gst_d p)(f) + T_d( (a_d^-1 - 1) .* (G_d p) )(f) ] ,
// a 13-point operator (two cells to either side along every axis) whose coefficients change with every time step.  fl_momentum.hip applied it as
// the composition of the kernels that exist -- projection (G and Gst), scaling, face interpolation T, divergence, a sign flip: seven passes,
// ~ 344 B/cell.  Here a cell forms the six face fluxes it needs from p (padded, ghost layers valid) and a^-1 directly: p 8 + a^-1 24 + y 8 = 40 B/cell
// of compulsory traffic.  Every 1-D row comes from the tables the composed kernels use (G: Gs / Gv0..2, Gst: gc0 / ga0 / ga1, T: the FaceT rows of
// kind 0, D: idx), so boundary rows are whatever those tables say; the only index arithmetic of its own is the periodic wrap of a CELL index
// (the cell whose gradient a face interpolates may be the image across the seam: its a^-1 and its G row are the image's).
//
// No LDS: a lane owns one cell of a 64-cell row segment, neighbours along x are the lanes next to it (the same cache lines), neighbours along y and
// z are rows other waves of the same XCD touch at about the same time -- the rows are dealt so that an XCD works through ONE contiguous band of y
// plane after plane, which keeps the five planes of p and three of a_z^-1 it needs inside its 4 MB of L2.  One rank only (the two-deep ring of p
// and the ring of a^-1 are not exchanged): several ranks keep the composition.
#include "fl_device.h"
#include "fl_handle.h"

namespace fl {

// The rows of the four 1-D operators, packed per cell / per face index of an axis (built once per handle from the device tables by k_sv_pack): the
// kernel takes 6 pointers instead of 36 -- with the 36 the compiler spilled 136 scalar registers and every row cost a chain of dependent loads
// (5.8 ms at 512^3, 5.0 with the packed rows) -- and a lane fetches a row in three 16-byte loads.
struct alignas(16) SvCell {
  double gv0, gv1, gv2, idx;  // G row of the cell, 1 / h
  int    gs, pad[3];          // its first column
};
struct alignas(16) SvFace {
  double ga0, ga1, w0, w1;  // Gst row and T row (kind 0) of the face
  int    gc0, c0, pad[2];   // their first columns
};
struct SvGrid {
  int           nx, ny, nz, sx;
  int64_t       sxy, off0;
  double        kappa;
  const SvCell *c[3];
  const SvFace *f[3];
};

__global__ void __launch_bounds__(256) k_sv_pack(GridP g, SchurVarT t, int d, int n, int nf, SvCell *c, SvFace *f)
{
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) {
    SvCell r;
    r.gv0 = g.Gv0[d][i];
    r.gv1 = g.Gv1[d][i];
    r.gv2 = g.Gv2[d][i];
    r.idx = g.idx[d][i];
    r.gs  = g.Gs[d][i];
    r.pad[0] = r.pad[1] = r.pad[2] = 0;
    c[i] = r;
  }
  if (i < nf) {
    SvFace r;
    r.ga0 = g.ga0[d][i];
    r.ga1 = g.ga1[d][i];
    r.w0  = t.w0[d][i];
    r.w1  = t.w1[d][i];
    r.gc0 = g.gc0[d][i];
    r.c0  = t.c0[d][i];
    r.pad[0] = r.pad[1] = 0;
    f[i] = r;
  }
}

// (five scalars by value, not an array or a struct: anything with an address that is picked from with a run-time index ends up in scratch memory)
__device__ __forceinline__ double sv_at(double p0, double p1, double p2, double p3, double p4, int r) { return r <= 0 ? p0 : (r == 1 ? p1 : (r == 2 ? p2 : (r == 3 ? p3 : p4))); }

__device__ __forceinline__ double sv_ldp(const double *__restrict__ p, int64_t pc0, int64_t ps, int m, int n, bool wraps)
{
  if (m < -1) m = wraps ? m + n : -1;
  else if (m > n) m = wraps ? m - n : n;
  return p[pc0 + m * ps];
}

// The contribution of axis D to (S p)(cell) / (-kappa): [ F(a+1) - F(a) ] / h_a with F(f) = (Gst_D p)(f) + T_D( w .* G_D p )(f), on the line through this
// lane's cell a: pc0 / uc0 = padded / unpadded index of the line's cell 0, ps / us the strides.  Everything the two faces need lies in a window of five
// cells of p and three of a^-1, loaded ONCE (the first version fetched every operand of every row again: 9.9 ms at 512^3, slower than the
// composition); the rows pick their operands from the window by their first columns.
// wraps: the axis is periodic inside this block -- face n is face 0 again, and a cell outside 0..n-1 is its image (ghost layers of p hold the
// images of cells -1 and n; -2 and n+1 are fetched from where they live)
// the rows an axis needs at cell a: of the cells a-1, a, a+1 (images across a periodic seam) and of the faces a, a+1
struct SvRows {
  SvCell c0, c1, c2;
  SvFace r0, r1;
  int    q0, q1, q2, f1;
};
__device__ __forceinline__ SvRows sv_rows(const SvCell *__restrict__ ct, const SvFace *__restrict__ ft, int a, int n, bool wraps)
{
  SvRows R;
  int    qw[3];
#pragma unroll
  for (int r = 0; r < 3; ++r) {
    const int q = a - 1 + r;
    qw[r] = q;
    if (q < 0) qw[r] = wraps ? q + n : 0;  // (a physical boundary: the T row carries no weight on a cell that does not exist)
    else if (q >= n) qw[r] = wraps ? q - n : n - 1;
  }
  R.q0 = qw[0]; R.q1 = qw[1]; R.q2 = qw[2];
  R.c0 = ct[qw[0]]; R.c1 = ct[qw[1]]; R.c2 = ct[qw[2]];
  R.f1 = (a + 1 == n && wraps) ? 0 : a + 1;
  R.r0 = ft[a];
  R.r1 = ft[R.f1];
  return R;
}

__device__ __forceinline__ double sv_axis(const SvRows &R, const double *__restrict__ p, const double *__restrict__ ainv, int a, int n, bool wraps, int64_t pc0,
                                          int64_t ps, int64_t uc0, int64_t us)
{
  const double P0 = sv_ldp(p, pc0, ps, a - 2, n, wraps), P1 = sv_ldp(p, pc0, ps, a - 1, n, wraps), P2 = sv_ldp(p, pc0, ps, a, n, wraps), P3 = sv_ldp(p, pc0, ps, a + 1, n, wraps),
               P4 = sv_ldp(p, pc0, ps, a + 2, n, wraps);
  const SvCell &c0 = R.c0, &c1 = R.c1, &c2 = R.c2;
  const SvFace &r0 = R.r0, &r1 = R.r1;
  const int     f1 = R.f1;
  const int     qw[3] = {R.q0, R.q1, R.q2};
  const double  w0 = ainv[uc0 + qw[0] * us] - 1., w1 = ainv[uc0 + qw[1] * us] - 1., w2 = ainv[uc0 + qw[2] * us] - 1.;
  const double  hinv = c1.idx;
  // first columns relative to the window / to the three gradients
  const int e0 = c0.gs - qw[0] + 1, e1 = c1.gs - qw[1] + 2, e2 = c2.gs - qw[2] + 3;   // G rows: 0, 1, 2 away from walls
  const int k0 = r0.gc0 - a + 2, k1 = r1.gc0 - f1 + 3;                                // Gst rows: 1, 2
  const int s0 = r0.c0 - a + 1, s1 = r1.c0 - f1 + 2;                                  // T rows: 0, 1
  double F[2];
  // Away from walls every row starts where a centred row starts: the operands are fixed members of the window and nothing has to be picked at run
  // time (the picks -- 17 five-way selections per axis -- made the kernel VALU-bound: 5.0 ms at 512^3).  Wave-uniform decision: along y and z it is
  // the same for all lanes anyway, along x only the waves that hold a wall cell take the general path.
  const bool centred = e0 == 0 && e1 == 1 && e2 == 2 && k0 == 1 && k1 == 2 && s0 == 0 && s1 == 1;
  if (__builtin_amdgcn_ballot_w64(!centred) == 0) {
    const double G0 = w0 * (c0.gv0 * P0 + c0.gv1 * P1 + c0.gv2 * P2), G1 = w1 * (c1.gv0 * P1 + c1.gv1 * P2 + c1.gv2 * P3), G2 = w2 * (c2.gv0 * P2 + c2.gv1 * P3 + c2.gv2 * P4);
    F[0] = r0.ga0 * P1 + r0.ga1 * P2 + r0.w0 * G0 + r0.w1 * G1;
    F[1] = r1.ga0 * P2 + r1.ga1 * P3 + r1.w0 * G1 + r1.w1 * G2;
  } else {
    const int    a0 = min(max(e0, 0), 2), a1 = min(max(e1, 0), 2), a2 = min(max(e2, 0), 2);
    const double G0 = w0 * (c0.gv0 * sv_at(P0, P1, P2, P3, P4, a0) + c0.gv1 * sv_at(P0, P1, P2, P3, P4, a0 + 1) + c0.gv2 * sv_at(P0, P1, P2, P3, P4, a0 + 2));
    const double G1 = w1 * (c1.gv0 * sv_at(P0, P1, P2, P3, P4, a1) + c1.gv1 * sv_at(P0, P1, P2, P3, P4, a1 + 1) + c1.gv2 * sv_at(P0, P1, P2, P3, P4, a1 + 2));
    const double G2 = w2 * (c2.gv0 * sv_at(P0, P1, P2, P3, P4, a2) + c2.gv1 * sv_at(P0, P1, P2, P3, P4, a2 + 1) + c2.gv2 * sv_at(P0, P1, P2, P3, P4, a2 + 2));
    auto pick = [](double g0, double g1, double g2, int k) { return k <= 0 ? g0 : (k == 1 ? g1 : g2); };
    double s = r0.ga0 * sv_at(P0, P1, P2, P3, P4, k0) + r0.ga1 * sv_at(P0, P1, P2, P3, P4, k0 + 1);
    if (r0.w0 != 0.) s += r0.w0 * pick(G0, G1, G2, s0);
    if (r0.w1 != 0.) s += r0.w1 * pick(G0, G1, G2, s0 + 1);
    F[0] = s;
    s = r1.ga0 * sv_at(P0, P1, P2, P3, P4, k1) + r1.ga1 * sv_at(P0, P1, P2, P3, P4, k1 + 1);
    if (r1.w0 != 0.) s += r1.w0 * pick(G0, G1, G2, s1);
    if (r1.w1 != 0.) s += r1.w1 * pick(G0, G1, G2, s1 + 1);
    F[1] = s;
  }
  return (F[1] - F[0]) * hinv;
}

// y (unpadded) = S p.  Grid: a multiple of 8 blocks; block b works for XCD b % 8 on the y band of that XCD.
#ifndef FL_SV_MINBLOCKS
#define FL_SV_MINBLOCKS 4   // four blocks per CU: 128 blocks per XCD are resident and cover exactly one plane of the XCD's band per loop trip
#endif
__global__ void __launch_bounds__(256, FL_SV_MINBLOCKS) k_schur_var(SvGrid g, int per, const double *__restrict__ p, const double *__restrict__ ainv, double *__restrict__ y)
{
  const int lane = threadIdx.x & 63, w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), nw = blockDim.x >> 6;
  const int xcd = blockIdx.x & 7, lb = blockIdx.x >> 3, nlb = gridDim.x >> 3;
  const int band = (g.ny + 7) / 8, j0 = xcd * band, j1 = min(j0 + band, g.ny);
  if (j0 >= j1) return;
  const int     nseg = (g.nx + 63) / 64, nj = j1 - j0;
  const int64_t nitem = (int64_t)nseg * nj * g.nz, N = (int64_t)g.nx * g.ny * g.nz;
  const bool    wx = per & 1, wy = per & 2, wz = per & 4;
  // a wave's stride is a multiple of the segments per row whenever the launch allows it: the wave then keeps ONE x segment for all its rows, and
  // the x rows of its lanes' cells (15 16-byte loads per lane) are fetched once instead of once per row
  const bool   fixed_seg = ((int64_t)nlb * nw) % nseg == 0;
  const int    seg0 = (int)(((int64_t)lb * nw + w) % nseg), i0 = min(seg0 * 64 + lane, g.nx - 1);
  const SvRows X0 = sv_rows(g.c[0], g.f[0], i0, g.nx, wx);
  for (int64_t it = (int64_t)lb * nw + w; it < nitem; it += (int64_t)nlb * nw) {
    const int seg = (int)(it % nseg), row = (int)(it / nseg);
    const int j = j0 + row % nj, k = row / nj, i = seg * 64 + lane;
    if (i >= g.nx) continue;
    const int64_t prow = g.off0 + (int64_t)k * g.sxy + (int64_t)j * g.sx, urow = ((int64_t)k * g.ny + j) * g.nx;
    double acc = sv_axis(fixed_seg ? X0 : sv_rows(g.c[0], g.f[0], i, g.nx, wx), p, ainv, i, g.nx, wx, prow, 1, urow, 1);
    acc += sv_axis(sv_rows(g.c[1], g.f[1], j, g.ny, wy), p, ainv + N, j, g.ny, wy, g.off0 + (int64_t)k * g.sxy + i, g.sx, (int64_t)k * g.ny * g.nx + i, g.nx);
    acc += sv_axis(sv_rows(g.c[2], g.f[2], k, g.nz, wz), p, ainv + 2 * N, k, g.nz, wz, g.off0 + (int64_t)j * g.sx + i, g.sxy, (int64_t)j * g.nx + i, (int64_t)g.nx * g.ny);
    y[urow + i] = -g.kappa * acc;
  }
}

}  // namespace fl

using namespace fl;

// p_pad: padded, ghost layers filled (fl_fill_ghosts); ainv: 3 * cells, unpadded, component-major; y: cells, unpadded.  One rank.
// t: the T rows (kind 0) of the caller's fl_momentum -- like the other tables a function of the grid and the boundary types only: packed once per handle
#ifndef FL_SV_BLOCKS_PER_XCD
#define FL_SV_BLOCKS_PER_XCD 128
#endif
int fl_schur_var_apply_fused(fl_poisson *h, const SchurVarT &t, const double *ainv, const double *p_pad, double *y)
{
  if (h->multi) return FL_ERR_SUP;
  const GridP &g = h->g;
  const int    n[3] = {g.nx, g.ny, g.nz}, nf[3] = {g.fx, g.fy, g.fz};
  if (!h->sv_pack[0]) {
    for (int d = 0; d < 3; ++d) {
      void *c = nullptr, *f = nullptr;
      FL_HIP(hipMalloc(&c, sizeof(SvCell) * (size_t)std::max(n[d], 1)));
      h->tables.push_back(c);
      FL_HIP(hipMalloc(&f, sizeof(SvFace) * (size_t)std::max(nf[d], 1)));
      h->tables.push_back(f);
      hipLaunchKernelGGL(k_sv_pack, dim3((std::max(n[d], nf[d]) + 255) / 256), dim3(256), 0, h->stream, g, t, d, n[d], nf[d], (SvCell *)c, (SvFace *)f);
      h->sv_pack[2 * d]     = c;
      h->sv_pack[2 * d + 1] = f;
    }
  }
  SvGrid sg;
  sg.nx = g.nx; sg.ny = g.ny; sg.nz = g.nz; sg.sx = g.sx;
  sg.sxy = g.sxy; sg.off0 = g.off0; sg.kappa = g.kappa;
  for (int d = 0; d < 3; ++d) {
    sg.c[d] = (const SvCell *)h->sv_pack[2 * d];
    sg.f[d] = (const SvFace *)h->sv_pack[2 * d + 1];
  }
  int per = 0;
  for (int d = 0; d < 3; ++d) per |= h->wrap_local[d] ? (1 << d) : 0;
  const int64_t items = (int64_t)((g.nx + 63) / 64) * g.ny * g.nz;
  // blocks per XCD: every block must be RESIDENT (four per CU, 32 CUs) -- the blocks of an XCD walk their band in step, and one that starts late walks
  // it again alone, when the planes its neighbours brought in have left the L2 -- and the rows they cover per loop trip should divide the band (64 rows
  // at 512^3), or the waves straddle two planes: 256 blocks 67.7 B/cell fetched, 96 (48 rows per trip) 84.1, 128 63.5, 64 43.7, 32 36.1 (32 compulsory;
  // fewer blocks are slower all the same: profiles/r05_schur_var.txt)
  const int     per_xcd = (int)std::max<int64_t>(1, std::min<int64_t>((items / 8 + 3) / 4, FL_SV_BLOCKS_PER_XCD));
  hipLaunchKernelGGL(k_schur_var, dim3(8 * per_xcd), dim3(256), 0, h->stream, sg, per, p_pad, ainv, y);
  FL_HIP(hipGetLastError());
  return 0;
}
