// fl_ksp.hip -- KSPBCGS and KSPCHEBYSHEV restatements for the Schur complement (the other -ns_abf_schur_ksp_type values a
// Fluca user reaches for: BiCGStab because S is non-symmetric on stretched grids, cnlinearcart3d.c:2348-2361; Chebyshev
// as the Jacobi smoother of BASELINE.json config 3).  Same conventions as the CG path in fl_kernels.hip: padded vectors,
// scalars in device memory (KspScal), fixed-order partial sums, lazy constant-null-space removal.
#include <ctime>
#include "fl_handle.h"
#include "fl_device.h"
#include "fl_stencil.h"

namespace fl {

// uniform base (scalar registers) + 32-bit per-lane byte offset: the addressing mode global_load/store ... v_off, s[base:base+1]
__device__ __forceinline__ double2 LD2(const double *base, unsigned byteoff) { return *reinterpret_cast<const double2 *>(reinterpret_cast<const char *>(base) + byteoff); }
__device__ __forceinline__ void    ST2(double *base, unsigned byteoff, double2 v) { *reinterpret_cast<double2 *>(reinterpret_cast<char *>(base) + byteoff) = v; }
__device__ __forceinline__ void    ST1(double *base, unsigned byteoff, double v) { *reinterpret_cast<double *>(reinterpret_cast<char *>(base) + byteoff) = v; }
// a value every lane holds identically (loaded from one address) moved to scalar registers.  (A scalar load of the same
// address -- constant address space -- measured 15% slower: s_load results return out of order and the wait for them also
// drains the lane-exchange queue.)
__device__ __forceinline__ double uniform(double v)
{
  const int lo = __builtin_amdgcn_readfirstlane(__double2loint(v)), hi = __builtin_amdgcn_readfirstlane(__double2hiint(v));
  return __hiloint2double(hi, lo);
}

// ------------------------------------------------------------------------------------------------ tile walker
// 128 x 4*RY x zc tiles like k_cg_B: lane = pair of x-adjacent cells, wave = RY rows, march in z.

struct Tile {
  int  i, il, k0, k1, lane, w, j0w, i0;
  bool own0, own1;
};
template <int RY>
__device__ __forceinline__ Tile make_tile(const GridP &g, int nchunk, int zc, int tiles_x)
{
  Tile      t;
  const int b = blockIdx.x, chunk = b % nchunk, tile = b / nchunk;
  const int i0 = (tile % tiles_x) * 128, j0 = (tile / tiles_x) * (4 * RY);
  t.k0   = chunk * zc;
  t.k1   = min(t.k0 + zc, g.nz);
  t.lane = threadIdx.x & 63;
  t.w    = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  t.i0   = i0;
  t.i    = i0 + 2 * t.lane;
  t.il   = min(t.i, g.nx & ~1);
  t.own0 = t.i < g.nx;
  t.own1 = t.i + 1 < g.nx;
  t.j0w  = j0 + t.w * RY;
  return t;
}

// ------------------------------------------------------------------------------------------------ generic fused SpMV
// y = M (S x) with M = 1/diag (JAC) or 1, x padded with valid ghosts, y padded.  Optional second output and dots:
//   partial slots: 0 sum y   1 y.o (o may be NULL)   2 x.y   3 y.y
//   unpadded_y: 0 y padded; 1 y unpadded; 2 y = o - S x with o, y unpadded; 3 y = o - S x with o, y padded
// MODE 0: plain.  MODE 1 (Chebyshev step): see k_cheb below (separate kernel).
// (k_cheb's plan -- halo one plane ahead, lane exchange, per-plane barrier -- was tried here too: it removes the 2.9 B/cell of
// excess fetch but the two-stream kernel is latency- rather than bandwidth-limited and BiCGStab got 4-7% slower, so this
// kernel keeps the plain loads.)
template <int RY, bool JAC>
__global__ void __launch_bounds__(256) k_apply_pc(GridP g, const double *__restrict__ x, double *__restrict__ y, const double *__restrict__ o, const KspScal *__restrict__ s, double *__restrict__ partial, int stride, int nchunk,
                                                  int zc, int tiles_x, int unpadded_y)
{
  __shared__ double red[4 * 4];
  if (s && s->reason != 0) return;
  const Tile   t = make_tile<RY>(g, nchunk, zc, tiles_x);
  const double xl0 = g.sl[0][min(t.i, g.nx)], xc0 = g.sc[0][min(t.i, g.nx)], xh0 = g.sh[0][min(t.i, g.nx)];
  const double xl1 = g.sl[0][min(t.i + 1, g.nx)], xc1 = g.sc[0][min(t.i + 1, g.nx)], xh1 = g.sh[0][min(t.i + 1, g.nx)];
  double       acc[4] = {0., 0., 0., 0.};
  if (t.k0 < t.k1) {
    double2 prev[RY], cur[RY], nxt[RY];
#pragma unroll
    for (int m = 0; m < RY; ++m) {
      const int64_t ro = g.off0 + (int64_t)min(t.j0w + m, g.ny) * g.sx + t.il;
      prev[m] = *reinterpret_cast<const double2 *>(x + ro + (int64_t)(t.k0 - 1) * g.sxy);
      cur[m]  = *reinterpret_cast<const double2 *>(x + ro + (int64_t)t.k0 * g.sxy);
    }
    for (int k = t.k0; k < t.k1; ++k) {
      const int64_t pc = (int64_t)k * g.sxy;
      const double  zl = g.sl[2][k], zcc = g.sc[2][k], zh = g.sh[2][k];
      double2       south, north;
      double        west[RY], east[RY];
      {
        const int64_t rs = g.off0 + (int64_t)min(t.j0w - 1, g.ny) * g.sx + t.il + pc;
        const int64_t rn = g.off0 + (int64_t)min(t.j0w + RY, g.ny + 0) * g.sx + t.il + pc;
        south = *reinterpret_cast<const double2 *>(x + rs);
        north = *reinterpret_cast<const double2 *>(x + rn);
      }
#pragma unroll
      for (int m = 0; m < RY; ++m) {
        const int64_t ro = g.off0 + (int64_t)min(t.j0w + m, g.ny) * g.sx + t.il;
        nxt[m]  = *reinterpret_cast<const double2 *>(x + ro + pc + g.sxy);
        west[m] = x[ro + pc - 1];
        east[m] = x[ro + pc + 2];
      }
#pragma unroll
      for (int m = 0; m < RY; ++m) {
        const int     j  = t.j0w + m, jc = min(j, g.ny);
        const double  yl = g.sl[1][jc], ycc = g.sc[1][jc], yh = g.sh[1][jc];
        const double2 so = m > 0 ? cur[m - 1] : south, no = m < RY - 1 ? cur[m + 1] : north;
        const double  dyz = ycc + zcc;
        double2       v;
        v.x = (xc0 + dyz) * cur[m].x + xl0 * west[m] + xh0 * cur[m].y + yl * so.x + yh * no.x + zl * prev[m].x + zh * nxt[m].x;
        v.y = (xc1 + dyz) * cur[m].y + xl1 * cur[m].x + xh1 * east[m] + yl * so.y + yh * no.y + zl * prev[m].y + zh * nxt[m].y;
        if (JAC) {
          v.x /= (xc0 + dyz);
          v.y /= (xc1 + dyz);
        }
        if (j < g.ny) {
          const int64_t ro = g.off0 + (int64_t)j * g.sx + t.il + pc;
          double2       ov = make_double2(0., 0.);
          if (o && unpadded_y != 2) ov = *reinterpret_cast<const double2 *>(o + ro);
          const bool unpadded_y_is_residual = unpadded_y == 3;
          if (unpadded_y == 1 || unpadded_y == 2) {
            const int64_t u = ((int64_t)k * g.ny + j) * g.nx + t.i;
            if (unpadded_y == 2) {  // residual: y = o - S x with o an UNPADDED right-hand side
              if (t.own0) v.x = o[u] - v.x;
              if (t.own1) v.y = o[u + 1] - v.y;
            }
            if (t.own0) y[u] = v.x;
            if (t.own1) y[u + 1] = v.y;
          } else {
            if (unpadded_y_is_residual) {  // y = o - S x, everything padded (multigrid residual)
              v.x = ov.x - v.x;
              v.y = ov.y - v.y;
            }
            if (t.own1) *reinterpret_cast<double2 *>(y + ro) = v;
            else if (t.own0) y[ro] = v.x;
          }
          if (t.own0) {
            acc[0] += v.x;
            acc[1] += v.x * ov.x;
            acc[2] += cur[m].x * v.x;
            acc[3] += v.x * v.x;
          }
          if (t.own1) {
            acc[0] += v.y;
            acc[1] += v.y * ov.y;
            acc[2] += cur[m].y * v.y;
            acc[3] += v.y * v.y;
          }
        }
      }
#pragma unroll
      for (int m = 0; m < RY; ++m) {
        prev[m] = cur[m];
        cur[m]  = nxt[m];
      }
    }
  }
  if (partial) {
    block_sum<4>(acc, red);
    if (threadIdx.x == 0)
#pragma unroll
      for (int a = 0; a < 4; ++a) partial[(int64_t)a * stride + blockIdx.x] = acc[a];
  }
}

// ------------------------------------------------------------------------------------------------ pointwise kernels
// OP 0 (BiCGStab P):  P = R - (omega_old*beta) (V0 - vshift) + beta P
// OP 1 (BiCGStab S):  S0 = R - alpha V0                       sums: 0 sum S0
// OP 2 (BiCGStab X,R): X += alpha P + omega (S0 - sshift) ; R = (S0 - sshift) - omega (T0 - tshift)
//                      sums: 0 R.R  1 R.RP  2 sum R
// OP 3 (init):        R = M b (unpadded b) ; RP = R (after the shift is known: see OP 4)   sums: 0 sum R0  1 R0.R0
// OP 4:               R -= shift ; RP = R
template <int RY, int OP, bool JAC>
__global__ void __launch_bounds__(256) k_bcgs_pw(GridP g, const double *__restrict__ a0, const double *__restrict__ a1, const double *__restrict__ a2, const double *__restrict__ a3, double *__restrict__ w0, double *__restrict__ w1,
                                                 const KspScal *__restrict__ s, double *__restrict__ partial, int stride, int nchunk, int zc, int tiles_x)
{
  __shared__ double red[3 * 4];
  if (OP != 3 && OP != 4 && s->reason != 0) return;
  const Tile   t = make_tile<RY>(g, nchunk, zc, tiles_x);
  const double xc0 = g.sc[0][min(t.i, g.nx)], xc1 = g.sc[0][min(t.i + 1, g.nx)];
  double       acc[3] = {0., 0., 0.};
  const double alpha = s->alpha, omega = s->omega, beta = s->beta, ob = s->omega_old * s->beta, vsh = s->vshift, ssh = s->rshift, tsh = s->tshift, sh0 = s->zshift;
  for (int k = t.k0; k < t.k1; ++k) {
#pragma unroll
    for (int m = 0; m < RY; ++m) {
      const int j = t.j0w + m;
      if (j >= g.ny || !t.own0) continue;
      const int64_t ro = g.off0 + (int64_t)j * g.sx + t.il + (int64_t)k * g.sxy;
      const bool    two = t.own1;
      auto L = [&](const double *p) { return two ? *reinterpret_cast<const double2 *>(p + ro) : make_double2(p[ro], 0.); };
      auto W = [&](double *p, double2 v) {
        if (two) *reinterpret_cast<double2 *>(p + ro) = v;
        else p[ro] = v.x;
      };
      if (OP == 0) {
        const double2 R = L(a0), V = L(a1), P = L(w0);
        double2       o;
        o.x = R.x - ob * (V.x - vsh) + beta * P.x;
        o.y = R.y - ob * (V.y - vsh) + beta * P.y;
        W(w0, o);
      } else if (OP == 1) {
        const double2 R = L(a0), V = L(a1);
        double2       o;
        o.x = R.x - alpha * V.x;
        o.y = R.y - alpha * V.y;
        W(w0, o);
        acc[0] += o.x + (two ? o.y : 0.);
      } else if (OP == 2) {
        const double2 P = L(a0), S0 = L(a1), T0 = L(a2), RP = L(a3), X = L(w0);
        double2       xs, rn;
        const double  s0 = S0.x - ssh, s1 = S0.y - ssh;
        xs.x = X.x + alpha * P.x + omega * s0;
        xs.y = X.y + alpha * P.y + omega * s1;
        rn.x = s0 - omega * (T0.x - tsh);
        rn.y = s1 - omega * (T0.y - tsh);
        W(w0, xs);
        W(w1, rn);
        acc[0] += rn.x * rn.x + (two ? rn.y * rn.y : 0.);
        acc[1] += rn.x * RP.x + (two ? rn.y * RP.y : 0.);
        acc[2] += rn.x + (two ? rn.y : 0.);
      } else if (OP == 3) {
        const int64_t u = ((int64_t)k * g.ny + j) * g.nx + t.i;
        const double  dyz = g.sc[1][j] + g.sc[2][k];
        double2       o;
        o.x = JAC ? a0[u] / (xc0 + dyz) : a0[u];
        o.y = two ? (JAC ? a0[u + 1] / (xc1 + dyz) : a0[u + 1]) : 0.;
        W(w0, o);
        acc[0] += o.x + o.y;
        acc[1] += o.x * o.x + o.y * o.y;
      } else {
        double2 R = L(w0);
        R.x -= sh0;
        R.y -= sh0;
        W(w0, R);
        W(w1, R);
      }
    }
  }
  if (OP == 1 || OP == 2 || OP == 3) {
    block_sum<3>(acc, red);
    if (threadIdx.x == 0)
#pragma unroll
      for (int a = 0; a < 3; ++a) partial[(int64_t)a * stride + blockIdx.x] = acc[a];
  }
}

// ------------------------------------------------------------------------------------------------ BiCGStab without stored products
// V0 = M S P and T0 = M S S0 are never written to memory: a 7-point row costs nothing next to its traffic, so every kernel that needs
// a product stages the vector it belongs to (tile + one-cell ring through LDS, the walk of k_cg_A / k_cg_Bq) and forms it again.
//   MODE 5  P' = R - (omega_old beta)(V0 - vshift) + beta P,  V0 from the OLD P (staged), P' into the other buffer   reads P,R   writes P'   24 B/cell
//   MODE 1  V0 from P' (staged): sums  0 sum V0, 1 V0.RP                                                          reads P',RP             16
//   MODE 2  S0 = R - alpha V0, V0 from P' again; sum 0 sum S0                                                     reads P',R  writes S0   24
//   MODE 3  T0 from S0 (staged): sums  0 sum T0, 2 S0.T0, 3 T0.T0                                                  reads S0                 8
//   MODE 4  X += alpha P' + omega (S0 - sshift); R = (S0 - sshift) - omega (T0 - tshift), T0 from S0 again;
//           sums 0 R.R, 1 R.RP, 2 sum R                                                               reads S0,P',X,RP  writes X,R         48
// = 120 B/cell/iteration against 152 with V0 and T0 stored (k_apply_pc + k_bcgs_pw).  Scalars, lazy null-space shifts and the
// convergence test are k_bcgs_fin's, unchanged.
// MODE 11 only: where the restricted residual goes (row / plane stride and offset of cell (0,0,0) of the coarse padded array)
struct StAux {
  int64_t csx = 0, csxy = 0, coff = 0;
};
template <int MODE>
struct BcgsIo {
  static constexpr int NE = (MODE == 4 || MODE == 10) ? 3 : ((MODE == 3 || MODE == 7 || MODE == 9) ? 0 : (MODE == 6 ? 2 : 1));  // extra per-cell input streams (plane of the product)
  static constexpr int NACC = MODE == 9 ? 7 : 4;                                                                                   // partial-sum slots
  static constexpr bool ZST = MODE == 9 || MODE == 10;  // the staged vector is z = M r formed while staging r (single-reduction CG)
};
// KSPCG with -ksp_cg_single_reduction (the option PETSc offers on the reference's sub-KSP as -ns_abf_schur_ksp_cg_single_reduction,
// prefix built at abfpc.c:206): every inner product of an iteration in ONE reduction.  With z = M r (staged while r is staged) and
// S = A z formed by the walk:
//   MODE 9   sums 0 z.r  1 z.S  2 z.z  3 sum z  4 sum r  5 sum S  6 r.r                                   reads r                  8 B/cell
//   MODE 10  p = (z - zshift) + b p;  W = S + b W (= A p by recurrence);  x += a p;  r' = r - a W into the OTHER r buffer (a tile reads
//            its neighbours' old r while they write the new one)                          reads r,p,W,x  writes r',p,W,x         64
// k_cgsr_fin turns the sums into b = beta'/beta, p.Ap = delta - b^2 (p.Ap)_old, a = beta'/p.Ap, the lazy null-space shift and the
// convergence test.  72 B/cell/iteration against 60 of the two-reduction pair (k_cg_A + k_cg_Bq): the price of one all-reduce and one
// scalar kernel less per iteration, worth it where those weigh as much as the kernels (several ranks, small blocks).
//   MODE 7  y = S x (padded), sums 0 sum y, 2 x.y, 3 y.y          MODE 8  r = o - S x (padded), no sums          (multigrid cycle, JAC = false)
//   MODE 11 the residual of MODE 8, restricted on the way (every axis halved): a lane's pair, the wave's two rows and two consecutive planes
//           are the eight children of one coarse cell; their weighted sum (k_mg_restrict's products in its order) goes to the coarse array
//           w0, the fine residual is never written.  e1 / e2 / w1: the restriction weights along x / y / z.       reads x,o  writes 1/8     17
//   MODE 6  (Chebyshev, one step: KSPCHEBYSHEV + PCJACOBI, the recurrence of k_cheb)  z = M (b - S x), x staged; d = rho d + c z (in place);
//           x' = x + d into the other x buffer; sums 0 sum z, 1 z.z, 2 r.r                              reads x,b,d  writes x',d          40
template <int RY, int NW, bool JAC, int MODE>
__device__ __forceinline__ void st_body(const GridP &g, const double *__restrict__ stg, const double *e0, const double *e1, const double *e2, double *w0, double *w1 /* MODE 4: e1 == w0 (X); MODE 6: e1 == w1 (d) */,
                                        const KspScal *__restrict__ s, double *__restrict__ partial, int stride, int nchunk, int zc, int tiles_x, int tiles, int remap, StAux ax = StAux())
{
  static_assert(MODE != 11 || RY == 2, "the restricted residual pairs the two rows of a wave");
  using T               = TileA<RY, NW>;
  constexpr int TX = T::TX, TY = T::TY, LX = T::LX, LY = T::LY;
  constexpr int NE = BcgsIo<MODE>::NE, NACC = BcgsIo<MODE>::NACC;
  constexpr bool ZST = BcgsIo<MODE>::ZST;
  __shared__ __attribute__((aligned(16))) double lds[3][LY][LX];
  __shared__ double                              red[NACC * NW];
  double alpha = 0., omega = 0., beta = 0., ob = 0., vsh = 0., ssh = 0., tsh = 0., crho = 0., cc = 0., zsh = 0.;
  if (MODE < 7 || MODE == 9 || MODE == 10) {  // MODE 7 / 8 / 11 (plain products for the multigrid cycle) run without a scalar block
    if (s->reason != 0) return;
    alpha = s->alpha; omega = s->omega; beta = s->beta; ob = s->omega_old * s->beta; vsh = s->vshift; ssh = s->rshift; tsh = s->tshift;
    crho = s->cheb_rho; cc = s->cheb_c; zsh = s->zshift;
  }

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
  // ring cells of this thread: rows -1 / TY (A), columns -1 / TX (B), as in k_cg_A
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
  // ZST: x + y part of the diagonal at this thread's ring cells (the ring of z = r / diag is formed from the ring of r)
  const double  hAdxy = (ZST && JAC && hAok) ? g.sc[0][hAi] + g.sc[1][hAj] : 1.;
  const double  hBdxy = (ZST && JAC && hBok) ? g.sc[0][hBi] + g.sc[1][hBj] : 1.;

  double  acc[NACC];
#pragma unroll
  for (int a = 0; a < NACC; ++a) acc[a] = 0.;
  double2 rprev[ZST ? RY : 1];  // ZST: the raw r of the plane whose product is formed (the staged copy is r / diag)
  double  racc = 0.;            // MODE 11: the coarse cell's sum so far
  const double rwx0 = MODE == 11 ? e1[min(i, g.nx - 1)] : 0., rwx1 = MODE == 11 ? e1[min(i + 1, g.nx - 1)] : 0.;
  double zlc = 0., zcc = 0., zhc = 0.;  // z-row of plane kk-1 (the plane whose product is formed)
  struct Raw {
    double2 v[RY];                 // staged vector, plane kn
    double2 e[NE ? NE : 1][RY];    // per-cell inputs of plane kn - 1
    double  hA, hB;
    double  zl, zc, zh;
  };
  auto load = [&](int kn_, Raw &R) {
    const int     kn = min(kn_, k1);
    const int64_t pl = (int64_t)kn * g.sxy, pr = (int64_t)min(max(kn_ - 1, k0), k1 - 1) * g.sxy;
#pragma unroll
    for (int m = 0; m < RY; ++m) {
      R.v[m] = ld2<1>(stg + RO(m) + pl);
      if (NE >= 1) R.e[0][m] = ld2<1>(e0 + RO(m) + pr);
      if (NE >= 2) {
        // MODE 6, first step of a recurrence (rho = 0: d is not looked at): its 8 B/cell are not fetched -- the post-smoother of every multigrid cycle
        // starts with such a step
        if (MODE == 6 && crho == 0.) R.e[1][m] = make_double2(0., 0.);
        else R.e[1][m] = ld2<1>(e1 + RO(m) + pr);
      }
      if (NE >= 3) R.e[NE >= 3 ? 2 : 0][m] = ld2<1>(e2 + RO(m) + pr);
    }
    R.hA = stg[tbase + pl + hAo];
    R.hB = stg[tbase + pl + hBo];
    R.zl = g.sl[2][kn];
    R.zc = g.sc[2][kn];
    R.zh = g.sh[2][kn];
  };
  auto step = [&](int kk, Raw &C, Raw &N) {
    load(kk + 1, N);
    const double nzl = C.zl, nzc = C.zc, nzh = C.zh;
    const int    buf = (kk + 3) % 3;
    const int    kc  = kk - 1;
    // what this step stages (and hands the product of plane kc as its z-high neighbour): the vector itself, or z = M r (ZST) -- every
    // cell divided by its own diagonal (a wall ghost has diagonal +inf: z = 0 there)
    double2 sv[RY];
#pragma unroll
    for (int m = 0; m < RY; ++m) {
      sv[m] = C.v[m];
      if (ZST && JAC) {
        sv[m].x /= (xc0 + yc[m] + C.zc);
        sv[m].y /= (xc1 + yc[m] + C.zc);
      }
    }
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
        double2       y;  // the product M S (staged vector) of this pair of cells
        y.x = st7(xc0 + dyc, cen.x, xl0, west, xh0, cen.y, yl[m], south.x, yh[m], north.x, zlc, below.x, zhc, sv[m].x);
        y.y = st7(xc1 + dyc, cen.y, xl1, cen.x, xh1, east, yl[m], south.y, yh[m], north.y, zlc, below.y, zhc, sv[m].y);
        if (JAC && MODE != 6 && !ZST) {
          y.x /= (xc0 + dyc);
          y.y /= (xc1 + dyc);
        }
        // selects, not 0/1 factors: outside the block the product is inf * 0 (the ghost diagonal is +inf)
        const bool o0 = rown[m] && own0, o1 = rown[m] && own1;
        auto       put = [&](double *dst, double2 v) {
          if (o1) st2<1>(dst + RO(m) + pc, v);
          else if (o0) dst[RO(m) + pc] = v.x;
        };
        if (MODE == 1) {
          const double2 rp = C.e[0][m];
          acc[0] += (o0 ? y.x : 0.) + (o1 ? y.y : 0.);
          acc[1] += (o0 ? y.x * rp.x : 0.) + (o1 ? y.y * rp.y : 0.);
        } else if (MODE == 2) {
          const double2 R = C.e[0][m];
          double2       o;
          o.x = R.x - alpha * y.x;
          o.y = R.y - alpha * y.y;
          put(w0, o);
          acc[0] += (o0 ? o.x : 0.) + (o1 ? o.y : 0.);
        } else if (MODE == 3) {
          acc[0] += (o0 ? y.x : 0.) + (o1 ? y.y : 0.);
          acc[2] += (o0 ? cen.x * y.x : 0.) + (o1 ? cen.y * y.y : 0.);
          acc[3] += (o0 ? y.x * y.x : 0.) + (o1 ? y.y * y.y : 0.);
        } else if (MODE == 4) {
          const double2 P = C.e[0][m], X = C.e[1][m], RP = C.e[NE >= 3 ? 2 : 0][m];
          const double  s0 = cen.x - ssh, s1 = cen.y - ssh;
          double2       xs, rn;
          xs.x = X.x + alpha * P.x + omega * s0;
          xs.y = X.y + alpha * P.y + omega * s1;
          rn.x = s0 - omega * (y.x - tsh);
          rn.y = s1 - omega * (y.y - tsh);
          put(w0, xs);
          put(w1, rn);
          acc[0] += (o0 ? rn.x * rn.x : 0.) + (o1 ? rn.y * rn.y : 0.);
          acc[1] += (o0 ? rn.x * RP.x : 0.) + (o1 ? rn.y * RP.y : 0.);
          acc[2] += (o0 ? rn.x : 0.) + (o1 ? rn.y : 0.);
        } else if (MODE == 9) {
          const double2 rr = rprev[ZST ? m : 0];
          if (w0) {  // several ranks: S = A z of the block's six boundary layers is kept (in the r buffer that is dead until MODE 10 refills it) for the pack of
                     // the overlapped exchange, which forms r - a (S + b W) there before MODE 10 does (k_pack_faces_sr) -- what PlanA::qb is to the pair
            const int jrow = jb + m;
            if (kc == 0 || kc == g.nz - 1 || jrow == 0 || jrow == g.ny - 1) put(w0, y);
            else {
              if (o0 && (i == 0 || i == g.nx - 1)) w0[rob[m] + pc + il] = y.x;
              if (o1 && i + 1 == g.nx - 1) w0[rob[m] + pc + il + 1] = y.y;
            }
          }
          acc[0] += (o0 ? cen.x * rr.x : 0.) + (o1 ? cen.y * rr.y : 0.);
          acc[1] += (o0 ? cen.x * y.x : 0.) + (o1 ? cen.y * y.y : 0.);
          acc[2] += (o0 ? cen.x * cen.x : 0.) + (o1 ? cen.y * cen.y : 0.);
          acc[3] += (o0 ? cen.x : 0.) + (o1 ? cen.y : 0.);
          acc[4] += (o0 ? rr.x : 0.) + (o1 ? rr.y : 0.);
          acc[5] += (o0 ? y.x : 0.) + (o1 ? y.y : 0.);
          acc[NACC > 6 ? 6 : 0] += (o0 ? rr.x * rr.x : 0.) + (o1 ? rr.y * rr.y : 0.);
        } else if (MODE == 10) {
          const double2 rr = rprev[ZST ? m : 0], P = C.e[0][m], Wv = C.e[NE >= 2 ? 1 : 0][m], X = C.e[NE >= 3 ? 2 : 0][m];
          double2       pn, wn, xn, rn;
          pn.x = (cen.x - zsh) + beta * P.x;
          pn.y = (cen.y - zsh) + beta * P.y;
          wn.x = fma(beta, Wv.x, y.x);  // explicit: k_pack_faces_sr forms the same two fma for the neighbour's ghost, bit for bit
          wn.y = fma(beta, Wv.y, y.y);
          xn.x = fma(alpha, pn.x, X.x);
          xn.y = fma(alpha, pn.y, X.y);
          rn.x = fma(-alpha, wn.x, rr.x);
          rn.y = fma(-alpha, wn.y, rr.y);
          put(const_cast<double *>(e0), pn);
          put(const_cast<double *>(e1), wn);
          put(const_cast<double *>(e2), xn);
          put(w0, rn);
        } else if (MODE == 7) {
          put(w0, y);
          acc[0] += (o0 ? y.x : 0.) + (o1 ? y.y : 0.);
          acc[2] += (o0 ? cen.x * y.x : 0.) + (o1 ? cen.y * y.y : 0.);
          acc[3] += (o0 ? y.x * y.x : 0.) + (o1 ? y.y * y.y : 0.);
        } else if (MODE == 8) {
          const double2 ov = C.e[0][m];
          put(w0, make_double2(ov.x - y.x, ov.y - y.y));
        } else if (MODE == 11) {
          const double2 ov = C.e[0][m];
          const double  wyz = e2[min(jb + m, g.ny - 1)] * reinterpret_cast<const double *>(w1)[kc];
          racc += wyz * (rwx0 * (ov.x - y.x) + rwx1 * (ov.y - y.y));
          if (m == RY - 1 && (kc & 1)) {
            if (o1) w0[ax.coff + (int64_t)(kc >> 1) * ax.csxy + (int64_t)(jb >> 1) * ax.csx + (i >> 1)] = racc;
            racc = 0.;
          }
        } else if (MODE == 6) {
          const double2 bv = C.e[0][m], dv = C.e[NE >= 2 ? 1 : 0][m];
          const double  r0 = bv.x - y.x, r1 = bv.y - y.y;  // y = S x here (not preconditioned)
          const double  z0 = JAC ? r0 / (xc0 + dyc) : r0, z1 = JAC ? r1 / (xc1 + dyc) : r1;
          double2       dn, xo;
          dn.x = (crho != 0. ? crho * dv.x : 0.) + cc * z0;  // first step: d is not looked at (it may hold anything)
          dn.y = (crho != 0. ? crho * dv.y : 0.) + cc * z1;
          xo.x = cen.x + dn.x;
          xo.y = cen.y + dn.y;
          put(w1, dn);
          put(w0, xo);
          acc[0] += (o0 ? z0 : 0.) + (o1 ? z1 : 0.);
          acc[1] += (o0 ? z0 * z0 : 0.) + (o1 ? z1 * z1 : 0.);
          acc[2] += (o0 ? r0 * r0 : 0.) + (o1 ? r1 * r1 : 0.);
        } else {  // MODE 5
          const double2 R = C.e[0][m];
          double2       o;
          o.x = R.x - ob * (y.x - vsh) + beta * cen.x;
          o.y = R.y - ob * (y.y - vsh) + beta * cen.y;
          put(w0, o);
        }
      }
    }
#pragma unroll
    for (int m = 0; m < RY; ++m) {
      if (ZST) rprev[m] = C.v[m];
      *reinterpret_cast<double2 *>(&lds[buf][w * RY + m + 1][2 * lane + 2]) = sv[m];
    }
    lds[buf][hAr][hAc] = (ZST && JAC) ? C.hA / (hAdxy + C.zc) : C.hA;
    lds[buf][hBr][hBc] = (ZST && JAC) ? C.hB / (hBdxy + C.zc) : C.hB;
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
  if (MODE == 5 || MODE == 8 || MODE == 10 || MODE == 11) return;  // no sums
#pragma unroll
  for (int a = 0; a < NACC; ++a) {
    acc[a] = wave_sum(acc[a]);
    if (lane == 0) red[a * NW + w] = acc[a];
  }
  __syncthreads();
  if (tid < NACC) {
    double t = 0.;
#pragma unroll
    for (int q = 0; q < NW; ++q) t += red[tid * NW + q];
    partial[(int64_t)tid * stride + blockIdx.x] = t;
  }
}
#ifndef FL_ST11_WPE
#define FL_ST11_WPE 4  // the restricted residual fits 128 VGPRs without spilling: two 512-thread blocks per CU (2: one block, 142 VGPRs)
#endif
template <int RY, int NW, bool JAC, int MODE>
__global__ void __launch_bounds__(64 * NW, (MODE == 11 && NW == 8) ? FL_ST11_WPE : 2) k_bcgs_st(GridP g, const double *__restrict__ stg, const double *e0, const double *e1, const double *e2, double *w0, double *w1, const KspScal *__restrict__ s,
                                                        double *__restrict__ partial, int stride, int nchunk, int zc, int tiles_x, int tiles, int remap, StAux ax)
{
  st_body<RY, NW, JAC, MODE>(g, stg, e0, e1, e2, w0, w1, s, partial, stride, nchunk, zc, tiles_x, tiles, remap, ax);
}
// One Chebyshev(-Jacobi) step on the LDS-staged walk (MODE 6): the buffers are picked on the device like in k_cheb (cur / dcur flip in the
// scalar kernels, the host enqueues steps without waiting)
template <int RY, int NW, bool JAC>
__global__ void __launch_bounds__(64 * NW, 2) k_cheb_st(GridP g, const double *X0, const double *X1, double *X0w, double *X1w, const double *__restrict__ b, double *D0, double *D1, const KspScal *__restrict__ s,
                                                        double *__restrict__ partial, int stride, int nchunk, int zc, int tiles_x, int tiles, int remap)
{
  if (s->reason != 0) return;
  const double *x  = s->cur ? X1 : X0;
  double       *xn = s->cur ? X0w : X1w;
  double       *d  = s->dcur ? D1 : D0;
  st_body<RY, NW, JAC, 6>(g, x, b, d, nullptr, xn, d, s, partial, stride, nchunk, zc, tiles_x, tiles, remap);
}

// BiCGStab scalar updates.  mode: 0 init (after OP 3)  1 after V0 = M S P   2 after S0   3 after T0 = M S S0   4 after OP 2
__global__ void __launch_bounds__(256) k_bcgs_fin(int mode, const double *__restrict__ partial, int nblocks, int stride, const double *__restrict__ sums, KspScal *__restrict__ s, double *__restrict__ hist, int nhist)
{
  __shared__ double out[NSLOT], red[NSLOT * 4];
  if (mode != 0 && s->reason != 0) return;
  if (nblocks > 0) reduce_partials(partial, nblocks, stride, 4, out, red);
  else {
    if (threadIdx.x < NSLOT) out[threadIdx.x] = sums[threadIdx.x];
    __syncthreads();
  }
  if (threadIdx.x != 0) return;
  const double N  = s->ncell_global;
  const bool   ns = s->nullspace != 0;
  if (mode == 0) {
    // R0 = M b: shift = mean, ||R||^2 = sum R0^2 - N mean^2 ; rho_0 = R.RP = ||R||^2
    const double m  = ns ? out[0] / N : 0.;
    const double rr = out[1] - N * m * m;
    s->zshift       = m;
    const double dp = sqrt(rr < 0. ? 0. : rr);
    s->rho          = rr;
    s->rho_old = s->alpha = s->omega_old = s->omega = 1.;
    s->beta   = 0.;
    s->vshift = s->rshift = s->tshift = 0.;
    s->it     = 0;
    s->rnorm0 = s->dp = dp;
    s->ttol   = fmax(s->rtol * dp, s->atol);
    if (hist && nhist > 0) hist[0] = dp;
    int reason = converged_default(s, dp);
    if (!reason && s->maxit <= 0) reason = FL_DIVERGED_ITS;
    if (!reason && s->rho == 0.) reason = FL_DIVERGED_BREAKDOWN;
    s->reason = reason;
    // first iteration: beta = (rho/1)(1/1) = rho, P = R - 1*beta*0 + beta*0 -> handled by P = V = 0 and the general formula
    s->beta = (s->rho / s->rho_old) * (s->alpha / s->omega_old);
  } else if (mode == 1) {
    // slots: 0 sum V0, 1 V0.RP (RP has zero mean: the shift drops out)
    s->vshift = ns ? out[0] / N : 0.;
    const double d1 = out[1];
    if (d1 == 0. || isnan(d1)) {
      s->reason = isnan(d1) ? FL_DIVERGED_NANORINF : FL_DIVERGED_BREAKDOWN;
      return;
    }
    s->alpha = s->rho / d1;
  } else if (mode == 2) {
    // S = S0 - rshift with S0 = R - alpha V0, V = V0 - vshift  ->  rshift = -alpha * vshift ; keep sum S0
    s->rshift = -s->alpha * s->vshift;
    s->d1     = out[0];  // sum S0
  } else if (mode == 3) {
    // slots: 0 sum T0, 2 S0.T0 (x.y), 3 T0.T0
    const double tb = ns ? out[0] / N : 0.;
    s->tshift       = tb;
    const double sb = s->rshift, sumS0 = s->d1, sumT0 = out[0];
    const double st = out[2] - sb * sumT0 - tb * sumS0 + N * sb * tb;
    const double tt = out[3] - 2. * tb * sumT0 + N * tb * tb;
    if (tt == 0.) {
      // S == 0: x + alpha P is the solution (handled by omega = 0 in the X update)
      s->omega = 0.;
      s->d2    = 0.;
    } else {
      s->omega = st / tt;
      s->d2    = tt;
    }
  } else {
    // slots: 0 R.R, 1 R.RP
    const double dp = s->d2 == 0. ? 0. : sqrt(out[0]);
    s->rho_old   = s->rho;
    s->rho       = out[1];
    s->omega_old = s->omega;
    s->it += 1;
    s->dp = dp;
    if (hist && s->it < nhist) hist[s->it] = dp;
    int reason = s->d2 == 0. ? FL_CONVERGED_RTOL : converged_default(s, dp);
    if (!reason) {
      if (s->it >= s->maxit) reason = FL_DIVERGED_ITS;
      else if (s->rho == 0.) reason = FL_DIVERGED_BREAKDOWN;
    }
    s->reason = reason;
    s->beta   = (s->rho / s->rho_old) * (s->alpha / s->omega_old);
  }
}

// Single-reduction CG scalar step.  mode 0: after the first MODE 9 pass (iteration 0), mode 1: after an update + MODE 9 pass.
// sums: 0 z.r  1 z.S  2 z.z  3 sum z  4 sum r  5 sum S  6 r.r  (z unshifted; the constant null space is taken out here: z' = z - m)
__global__ void __launch_bounds__(256) k_cgsr_fin(int mode, const double *__restrict__ partial, int nblocks, int stride, const double *__restrict__ sums, KspScal *__restrict__ s, double *__restrict__ hist, int nhist)
{
  __shared__ double out[NSLOT], red[NSLOT * 4];
  if (mode != 0 && s->reason != 0) return;
  if (nblocks > 0) reduce_partials(partial, nblocks, stride, 7, out, red);
  else {
    if (threadIdx.x < NSLOT) out[threadIdx.x] = sums[threadIdx.x];
    __syncthreads();
  }
  if (threadIdx.x != 0) return;
  const double N = s->ncell_global;
  const double m = s->nullspace ? out[3] / N : 0.;
  const double betan = out[0] - m * out[4];  // (z', r)
  const double delta = out[1] - m * out[5];  // (z', S), S = A z = A z'
  double       dp;
  switch (s->norm_type) {
  case FL_NORM_PRECONDITIONED: {
    const double zz = out[2] - N * m * m;
    dp              = sqrt(zz < 0. ? 0. : zz);
  } break;
  case FL_NORM_UNPRECONDITIONED: dp = sqrt(out[6]); break;
  case FL_NORM_NATURAL: dp = sqrt(fabs(betan)); break;
  default: dp = 0.;
  }
  s->zshift = m;
  if (mode == 0) {
    s->it     = 0;
    s->rnorm0 = dp;
    s->ttol   = fmax(s->rtol * dp, s->atol);
  } else s->it += 1;
  s->dp = dp;
  if (hist && s->it < nhist) hist[s->it] = dp;
  int reason = converged_default(s, dp);
  double b = 0., dpi = delta;
  if (!reason) {
    if (s->it >= s->maxit) reason = FL_DIVERGED_ITS;
    else if (betan < 0.) reason = FL_DIVERGED_INDEFINITE_PC;
    else {
      if (mode != 0) {
        b   = betan / s->rz;
        dpi = delta - b * b * s->pq;  // (p, A p) for p = z + b p_old: delta - 2 b beta'/a_old + b^2 (p.Ap)_old = delta - b^2 (p.Ap)_old
      }
      if (dpi <= 0. || isnan(dpi)) reason = isnan(dpi) ? FL_DIVERGED_NANORINF : FL_DIVERGED_INDEFINITE_MAT;
    }
  }
  s->reason = reason;
  if (reason) return;  // alpha / beta keep the values of the last update: nothing runs after this
  s->rz_old = s->rz;
  s->rz     = betan;
  s->pq     = dpi;
  s->beta   = b;
  s->alpha  = betan / dpi;
}

// ------------------------------------------------------------------------------------------------ Chebyshev
// One fused step: z = M (b - S x) ; d = rho d + c z ; x' = x + d   (x -> the other buffer, d in place)
// sums: 0 sum z  1 z.z  2 r.r
template <int RY, bool JAC>
__global__ void __launch_bounds__(256) k_cheb(GridP g, const double *X0, const double *X1, double *X0w, double *X1w, const double *__restrict__ b, double *D0, double *D1,
                                              const KspScal *__restrict__ s, double *__restrict__ partial, int stride, int nchunk, int zc, int tiles_x)
{
  __shared__ double red[3 * 4];
  if (s->reason != 0) return;
  const double *x  = s->cur ? X1 : X0;
  double       *xn = s->cur ? X0w : X1w;
  double *__restrict__ d = s->dcur ? D1 : D0;  // updated in place (the fused two-step kernel of fl_cheb2.hip flips dcur)
  const double  rho = s->cheb_rho, cc = s->cheb_c;
  const Tile    t = make_tile<RY>(g, nchunk, zc, tiles_x);
  const double  xl0 = g.sl[0][min(t.i, g.nx)], xc0 = g.sc[0][min(t.i, g.nx)], xh0 = g.sh[0][min(t.i, g.nx)];
  const double  xl1 = g.sl[0][min(t.i + 1, g.nx)], xc1 = g.sc[0][min(t.i + 1, g.nx)], xh1 = g.sh[0][min(t.i + 1, g.nx)];
  double        acc[3] = {0., 0., 0.};
  if (t.k0 < t.k1) {
    // Addressing: row and plane offsets are wave-uniform (scalar registers); the only per-lane part of any address in
    // this kernel is lo = 8 * il, one VGPR for all streams (LD2/ST2: uniform base + 32-bit lane byte offset).
    // Everything a plane needs from memory is loaded ONE plane ahead, together with the tile's own rows of that plane:
    // the y-halo rows (south/north) are the rows the neighbouring wave streams at that moment, and the x neighbours
    // inside the tile come from the adjacent lane (only the two cells outside the 128-wide tile are loaded, from one
    // uniform address).  Loading them a plane late instead -- after five streams went through -- finds the lines evicted
    // and fetches them from HBM again: 34.9 B/cell measured against 24 algorithmic + 3 of tile ring; 26.7 with this.
    double2       prev[RY], cur[RY], nxt[RY], so_c, no_c, so_n, no_n;
    double        we_c[RY], ee_c[RY], we_n[RY], ee_n[RY];
    unsigned      lo = 8u * (unsigned)t.il;
    const int     i0 = t.i0;
    const int64_t sxy = g.sxy;
    const int64_t rsou = g.off0 + (int64_t)min(t.j0w - 1, g.ny) * g.sx, rnor = g.off0 + (int64_t)min(t.j0w + RY, g.ny) * g.sx;
    int64_t       rj[RY];
#pragma unroll
    for (int m = 0; m < RY; ++m) rj[m] = g.off0 + (int64_t)min(t.j0w + m, g.ny) * g.sx;
    const int wofs = i0 - 1, eofs = min(i0 + 128, g.nx);
    {
      const int64_t p0 = (int64_t)t.k0 * sxy;
      so_c = LD2(x + rsou + p0, lo);
      no_c = LD2(x + rnor + p0, lo);
#pragma unroll
      for (int m = 0; m < RY; ++m) {
        prev[m] = LD2(x + rj[m] + p0 - sxy, lo);
        cur[m]  = LD2(x + rj[m] + p0, lo);
        we_c[m] = uniform(x[rj[m] + wofs + p0]);
        ee_c[m] = uniform(x[rj[m] + eofs + p0]);
      }
    }
    for (int k = t.k0; k < t.k1; ++k) {
      asm volatile("" : "+s"(k), "+v"(lo));  // keep the compiler from strength-reducing every stream into its own 64-bit VGPR pointer
      const int64_t pc = (int64_t)k * sxy;
      const double  zl = g.sl[2][k], zcc = g.sc[2][k], zh = g.sh[2][k];
      double2       bv[RY], dv[RY];
      // every load of the plane is issued here, unconditionally and on clamped (always valid) addresses: a load inside the
      // divergent ownership branch below would make the compiler drain the whole load queue first
      so_n = LD2(x + rsou + pc + sxy, lo);
      no_n = LD2(x + rnor + pc + sxy, lo);
#pragma unroll
      for (int m = 0; m < RY; ++m) {
        nxt[m]  = LD2(x + rj[m] + pc + sxy, lo);
        we_n[m] = x[rj[m] + wofs + pc + sxy];
        ee_n[m] = x[rj[m] + eofs + pc + sxy];
        bv[m]   = LD2(b + rj[m] + pc, lo);
        dv[m]   = LD2(d + rj[m] + pc, lo);
      }
#pragma unroll
      for (int m = 0; m < RY; ++m) {
        const int     j  = t.j0w + m, jc = min(j, g.ny);
        const double  yl = g.sl[1][jc], ycc = g.sc[1][jc], yh = g.sh[1][jc];
        const double2 so = m > 0 ? cur[m - 1] : so_c, no = m < RY - 1 ? cur[m + 1] : no_c;
        const double  up = __shfl_up(cur[m].y, 1), dn = __shfl_down(cur[m].x, 1);
        const double  west = t.lane == 0 ? we_c[m] : up, east = t.lane == 63 ? ee_c[m] : dn;
        const double  dyz = ycc + zcc;
        double2       v;
        v.x = (xc0 + dyz) * cur[m].x + xl0 * west + xh0 * cur[m].y + yl * so.x + yh * no.x + zl * prev[m].x + zh * nxt[m].x;
        v.y = (xc1 + dyz) * cur[m].y + xl1 * cur[m].x + xh1 * east + yl * so.y + yh * no.y + zl * prev[m].y + zh * nxt[m].y;
        if (j < g.ny && t.own0) {
          const bool   two = t.own1;
          const double r0 = bv[m].x - v.x, r1 = two ? bv[m].y - v.y : 0.;
          const double z0 = JAC ? r0 / (xc0 + dyz) : r0, z1 = two ? (JAC ? r1 / (xc1 + dyz) : r1) : 0.;
          double2      dn2, xo;
          dn2.x = (rho != 0. ? rho * dv[m].x : 0.) + cc * z0;  // first step: d is not looked at (it may hold anything)
          dn2.y = (rho != 0. ? rho * dv[m].y : 0.) + cc * z1;
          xo.x  = cur[m].x + dn2.x;
          xo.y  = cur[m].y + dn2.y;
          if (two) {
            ST2(d + rj[m] + pc, lo, dn2);
            ST2(xn + rj[m] + pc, lo, xo);
          } else {
            ST1(d + rj[m] + pc, lo, dn2.x);
            ST1(xn + rj[m] + pc, lo, xo.x);
          }
          acc[0] += z0 + z1;
          acc[1] += z0 * z0 + z1 * z1;
          acc[2] += r0 * r0 + r1 * r1;
        }
      }
#pragma unroll
      for (int m = 0; m < RY; ++m) {
        prev[m] = cur[m];
        cur[m]  = nxt[m];
        we_c[m] = uniform(we_n[m]);
        ee_c[m] = uniform(ee_n[m]);
      }
      so_c = so_n;
      no_c = no_n;
      // keep the four waves of the tile on the same plane, so that a wave's halo rows are in flight with its neighbour's own rows
      __syncthreads();
    }
  }
  block_sum<3>(acc, red);
  if (threadIdx.x == 0)
#pragma unroll
    for (int a = 0; a < 3; ++a) partial[(int64_t)a * stride + blockIdx.x] = acc[a];
}

// The first step from a zero initial guess (rho = 0, x = 0): z = M b ; d = c z ; x' = d.  The same numbers k_cheb produces
// from a zeroed x (S 0 = +0, b - 0 = b, 0 + d = d), without zeroing x, reading it and d, or the stencil: 8 B/cell read and
// 16 written instead of 8 (memset) + 24 + 16.  No sums: only for the smoother call (KSP_NORM_NONE, no null space).
// SUB: the right-hand side is updated on the way, b -= suba * subq (the outer CG's r -= alpha q in front of a multigrid cycle); the step length
// is read from DEVICE memory (the outer iteration's scalars never visit the host between two cycles, fl_mg.hip)
template <bool JAC, bool SUB = false>
__global__ void __launch_bounds__(256) k_cheb_first(GridP g, double *X0w, double *X1w, double *__restrict__ b, double *D0, double *D1, const KspScal *__restrict__ s, const double *__restrict__ subq = nullptr,
                                                    const double *__restrict__ suba_dev = nullptr)
{
  if (s->reason != 0) return;
  const double suba = SUB ? *suba_dev : 0.;
  double       *xn = s->cur ? X0w : X1w;
  double *__restrict__ d = s->dcur ? D1 : D0;
  const double  cc = s->cheb_c;
  // a wave per 128-cell row segment, grid-stride over (segment, row): the row and plane numbers are wave-uniform, so the index arithmetic and
  // the y / z table reads are scalar (the per-pair 64-bit divisions of a flat grid-stride loop cost more issue slots than the update itself)
  const int     lane = threadIdx.x & 63, w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), nw = blockDim.x >> 6;
  const int     nseg = (g.nx + 127) / 128;
  const int64_t nitem = (int64_t)nseg * g.ny * g.nz;
  for (int64_t it = (int64_t)blockIdx.x * nw + w; it < nitem; it += (int64_t)gridDim.x * nw) {
    const int     seg = (int)(it % nseg), row = (int)(it / nseg);
    const int     j = row % g.ny, k = row / g.ny, i = seg * 128 + 2 * lane;
    if (i >= g.nx) continue;
    const bool    two = i + 1 < g.nx;
    const int64_t off = g.off0 + (int64_t)k * g.sxy + (int64_t)j * g.sx + i;
    const double  dyz = g.sc[1][j] + g.sc[2][k];
    double2       bv;
    if (two) bv = *reinterpret_cast<const double2 *>(b + off);
    else bv = make_double2(b[off], 0.);
    if (SUB) {
      if (two) {
        const double2 qv = *reinterpret_cast<const double2 *>(subq + off);
        bv = make_double2(bv.x - suba * qv.x, bv.y - suba * qv.y);
        *reinterpret_cast<double2 *>(b + off) = bv;
      } else {
        bv.x -= suba * subq[off];
        b[off] = bv.x;
      }
    }
    const double  z0 = JAC ? bv.x / (g.sc[0][i] + dyz) : bv.x, z1 = two ? (JAC ? bv.y / (g.sc[0][i + 1] + dyz) : bv.y) : 0.;
    const double2 dn = make_double2(0. + cc * z0, 0. + cc * z1);
    if (two) {
      *reinterpret_cast<double2 *>(d + off)  = dn;
      *reinterpret_cast<double2 *>(xn + off) = dn;
    } else {
      d[off]  = dn.x;
      xn[off] = dn.x;
    }
  }
}

// after launch j of k_cheb: convergence test on the residual of x_j, then either stop (answer = old buffer, x_j) or accept
// the update (flip) and prepare rho, c of the next step.
__global__ void __launch_bounds__(256) k_cheb_fin(const double *__restrict__ partial, int nblocks, int stride, const double *__restrict__ sums, KspScal *__restrict__ s, double *__restrict__ hist, int nhist)
{
  __shared__ double out[NSLOT], red[NSLOT * 4];
  if (s->reason != 0) return;
  if (nblocks > 0) reduce_partials(partial, nblocks, stride, 3, out, red);
  else {
    if (threadIdx.x < NSLOT) out[threadIdx.x] = sums[threadIdx.x];
    __syncthreads();
  }
  if (threadIdx.x != 0) return;
  const double N = s->ncell_global;
  const double m = s->nullspace ? out[0] / N : 0.;
  const int    j = s->it;  // index of the step just executed
  if (s->norm_type != FL_NORM_NONE) {
    const double zz = out[1] - N * m * m;
    const double dp = s->norm_type == FL_NORM_UNPRECONDITIONED ? sqrt(out[2]) : sqrt(zz < 0. ? 0. : zz);
    if (j == 0) {
      s->rnorm0 = dp;
      s->ttol   = fmax(s->rtol * dp, s->atol);
    }
    s->dp = dp;
    if (hist && j < nhist) hist[j] = dp;
    int reason = converged_default(s, dp);
    if (!reason && j >= s->maxit) reason = FL_DIVERGED_ITS;
    if (reason) {
      s->reason = reason;  // answer: x_j in the buffer this step read (cur not flipped), constant shift xshift
      return;
    }
  }
  // accept x_{j+1} = x_j + d_{j+1};  lazy null-space bookkeeping: d_true = d - dshift, x_true = x - xshift
  s->dshift = s->cheb_rho * s->dshift + s->cheb_c * m;
  s->xshift += s->dshift;
  s->cur ^= 1;
  s->it = j + 1;
  if (s->norm_type == FL_NORM_NONE && s->it >= s->maxit) {
    s->reason = FL_CONVERGED_ITS;
    return;
  }
  // next step: c_{k+1} = 2 mu c_k - c_{k-1}; omega = omegaprod c_k / c_{k+1}; d' = (omega-1) d + omega*scale z
  double ck, ckm1, rho, c;
  cheb_advance(s, s->ck, s->ckm1, ck, ckm1, rho, c);
  s->ck       = ck;
  s->ckm1     = ckm1;
  s->cheb_rho = rho;
  s->cheb_c   = c;
}

// after one launch of k_cheb2 (steps j and j+1, KSP_NORM_NONE only): both updates are accepted; slots 0..2 step j, 3..5 step j+1
__global__ void __launch_bounds__(256) k_cheb_fin2(const double *__restrict__ partial, int nblocks, int stride, const double *__restrict__ sums, KspScal *__restrict__ s)
{
  __shared__ double out[NSLOT], red[NSLOT * 4];
  if (s->reason != 0) return;
  if (nblocks > 0) reduce_partials(partial, nblocks, stride, 6, out, red);
  else {
    if (threadIdx.x < NSLOT) out[threadIdx.x] = sums[threadIdx.x];
    __syncthreads();
  }
  if (threadIdx.x != 0) return;
  const double N  = s->ncell_global;
  const double m0 = s->nullspace ? out[0] / N : 0., m1 = s->nullspace ? out[3] / N : 0.;
  double       ck, ckm1, rho1, c1, rho2, c2;
  cheb_advance(s, s->ck, s->ckm1, ck, ckm1, rho1, c1);  // the coefficients k_cheb2 used for its second step
  s->dshift = s->cheb_rho * s->dshift + s->cheb_c * m0;
  s->xshift += s->dshift;
  s->dshift = rho1 * s->dshift + c1 * m1;
  s->xshift += s->dshift;
  s->cur ^= 1;   // x'' went to the other x buffer,
  s->dcur ^= 1;  // d'' to the other d buffer
  s->it += 2;
  if (s->it >= s->maxit) {
    s->reason = FL_CONVERGED_ITS;
    return;
  }
  cheb_advance(s, ck, ckm1, s->ck, s->ckm1, rho2, c2);
  s->cheb_rho = rho2;
  s->cheb_c   = c2;
}

}  // namespace fl

using namespace fl;

// ------------------------------------------------------------------------------------------------ drivers

namespace {

struct TP {
  int ry, nchunk, zc, tiles_x, nblocks;
};
TP tile_plan(const GridP &g)
{
  TP t;
  t.ry = g.ny >= 16 ? 2 : 1;
  t.tiles_x = (g.nx + 127) / 128;
  const int tiles_y = (g.ny + 4 * t.ry - 1) / (4 * t.ry), tiles = t.tiles_x * tiles_y;
  int       nchunk = std::max(1, (1024 + tiles / 2) / tiles);
  nchunk   = std::max(1, std::min(std::min(nchunk, std::max(1, g.nz / 8)), g.nz));
  t.zc     = (g.nz + nchunk - 1) / nchunk;
  t.nchunk = (g.nz + t.zc - 1) / t.zc;
  t.nblocks = tiles * t.nchunk;
  return t;
}

template <int RY, bool JAC>
void apply_pc_t(fl_poisson *h, const TP &tp, const double *x, double *y, const double *o, const KspScal *s, double *partial, int unpadded_y)
{
  hipLaunchKernelGGL((k_apply_pc<RY, JAC>), dim3(tp.nblocks), dim3(256), 0, h->stream, h->g, x, y, o, s, partial, h->partial_stride, tp.nchunk, tp.zc, tp.tiles_x, unpadded_y);
}
void launch_apply_pc(fl_poisson *h, const TP &tp, bool jac, const double *x, double *y, const double *o, const KspScal *s, double *partial, int unpadded_y)
{
  if (jac) {
    if (tp.ry == 2) apply_pc_t<2, true>(h, tp, x, y, o, s, partial, unpadded_y);
    else apply_pc_t<1, true>(h, tp, x, y, o, s, partial, unpadded_y);
  } else {
    if (tp.ry == 2) apply_pc_t<2, false>(h, tp, x, y, o, s, partial, unpadded_y);
    else apply_pc_t<1, false>(h, tp, x, y, o, s, partial, unpadded_y);
  }
}

template <int RY, int OP, bool JAC>
void pw_t(fl_poisson *h, const TP &tp, const double *a0, const double *a1, const double *a2, const double *a3, double *w0, double *w1)
{
  hipLaunchKernelGGL((k_bcgs_pw<RY, OP, JAC>), dim3(tp.nblocks), dim3(256), 0, h->stream, h->g, a0, a1, a2, a3, w0, w1, h->scal, h->partial, h->partial_stride, tp.nchunk, tp.zc, tp.tiles_x);
}
template <int OP>
void launch_pw(fl_poisson *h, const TP &tp, bool jac, const double *a0, const double *a1, const double *a2, const double *a3, double *w0, double *w1)
{
  if (jac) {
    if (tp.ry == 2) pw_t<2, OP, true>(h, tp, a0, a1, a2, a3, w0, w1);
    else pw_t<1, OP, true>(h, tp, a0, a1, a2, a3, w0, w1);
  } else {
    if (tp.ry == 2) pw_t<2, OP, false>(h, tp, a0, a1, a2, a3, w0, w1);
    else pw_t<1, OP, false>(h, tp, a0, a1, a2, a3, w0, w1);
  }
}

template <int RY, bool JAC>
void cheb_t(fl_poisson *h, const TP &tp, double *X0, double *X1, const double *B, double *D0, double *D1)
{
  hipLaunchKernelGGL((k_cheb<RY, JAC>), dim3(tp.nblocks), dim3(256), 0, h->stream, h->g, X0, X1, X0, X1, B, D0, D1, h->scal, h->partial, h->partial_stride, tp.nchunk, tp.zc, tp.tiles_x);
}
template <int RY, int NW>
void cheb_st_t(fl_poisson *h, const PlanA &p, bool jac, double *X0, double *X1, const double *B, double *D0, double *D1)
{
  const int  tiles = p.tiles_x * p.tiles_y;
  const dim3 gr(p.nblocks), bl(64 * NW);
  if (jac) hipLaunchKernelGGL((k_cheb_st<RY, NW, true>), gr, bl, 0, h->stream, h->g, X0, X1, X0, X1, B, D0, D1, h->scal, h->partial, h->partial_stride, p.nchunk, p.zc, p.tiles_x, tiles, p.remap);
  else hipLaunchKernelGGL((k_cheb_st<RY, NW, false>), gr, bl, 0, h->stream, h->g, X0, X1, X0, X1, B, D0, D1, h->scal, h->partial, h->partial_stride, p.nchunk, p.zc, p.tiles_x, tiles, p.remap);
}
// variant switch "cheb_staged": 1 (shipped) the one-step kernel on the LDS-staged walk (k_cheb_st), 0 round 1's k_cheb (kbench build only; the product keeps
// k_cheb for the one case the staged walk does not cover: more blocks than partial-sum slots)
static inline int cheb_staged_mode() { return FL_VARIANT(cheb_staged, 1); }
// one Chebyshev step; returns the number of blocks whose partial sums the scalar kernel has to add up
int launch_cheb(fl_poisson *h, const TP &tp, bool jac, double *X0, double *X1, const double *B, double *D0, double *D1)
{
  if (cheb_staged_mode() != 0) {
    const PlanA pa = plan_cg_A(h->g, 0, 0);
    if (pa.nblocks <= h->partial_stride) {
      switch (pa.ry * 10 + pa.nw) {
      case 28: cheb_st_t<2, 8>(h, pa, jac, X0, X1, B, D0, D1); break;
      case 24: cheb_st_t<2, 4>(h, pa, jac, X0, X1, B, D0, D1); break;
      default: cheb_st_t<1, 4>(h, pa, jac, X0, X1, B, D0, D1); break;
      }
      return pa.nblocks;
    }
  }
  if (jac) {
    if (tp.ry == 2) cheb_t<2, true>(h, tp, X0, X1, B, D0, D1);
    else cheb_t<1, true>(h, tp, X0, X1, B, D0, D1);
  } else {
    if (tp.ry == 2) cheb_t<2, false>(h, tp, X0, X1, B, D0, D1);
    else cheb_t<1, false>(h, tp, X0, X1, B, D0, D1);
  }
  return tp.nblocks;
}

// "cheb_fuse" (fl_tuning_set): 0 never use the fused two-step kernel, 1 (default) where it pays (grids of at least 32^3 cells), 2 wherever it is
// legal (tests)
static inline int cheb_fuse_mode() { return knob(K_cheb_fuse); }
// -> 1 fused, 0 one step per launch, < 0 error.  Collective on several ranks the first time a handle asks (fl_cheb2_agree): call it where every
// rank passes, before any test that could differ between ranks.
int cheb_fuse(fl_poisson *h)
{
  FL_CHK(fl_cheb2_agree(h));
  const int m = cheb_fuse_mode();
  if (m <= 0) return 0;
  return h->cheb2_agreed[m >= 2 ? 0 : 1];
}

// the scalar block handed over BY VALUE (kernel argument): nothing reads the host copy after the launch returns, so the next
// call may refill it while this one is still queued
__global__ void k_scal_set(KspScal *dst, KspScal v) { *dst = v; }

// partial sums -> scalar kernel, with the all-reduce in between when there is more than one rank
template <class F>
int fin_step(fl_poisson *h, int nblocks, int nslot, F &&launch_fin)
{
  if (!h->multi) {
    launch_fin(h->partial, nblocks, h->partial_stride, (const double *)nullptr);
    return 0;
  }
  launch_reduce(h->stream, h->partial, nblocks, h->partial_stride, nslot, h->sums);
  FL_CHK(h->comm.allreduce(h->stream, h->sums, NSLOT));
  launch_fin((const double *)nullptr, 0, 0, (const double *)h->sums);
  return 0;
}

void init_scal(fl_poisson *h, const fl_ksp_opts *o)
{
  KspScal &S = *h->scal_host;
  std::memset(&S, 0, sizeof(S));
  S.rtol         = o->rtol;
  S.atol         = o->atol;
  S.dtol         = o->dtol;
  S.ncell_global = (double)h->ax[0].n * (double)h->ax[1].n * (double)h->ax[2].n;
  S.maxit        = o->maxit;
  S.norm_type    = o->norm_type;
  S.nullspace    = o->remove_nullspace;
}

int finish_stats(fl_poisson *h, const fl_ksp_opts *o, fl_ksp_stats *st)
{
  FL_HIP(hipEventRecord(h->ev1, h->stream));
  FL_CHK(fl_poll_scal(h));
  FL_HIP(hipGetLastError());
  float ms = 0.f;
  FL_HIP(hipEventElapsedTime(&ms, h->ev0, h->ev1));
  const KspScal &R = *h->scal_host;
  st->iters        = R.it;
  st->reason       = R.reason ? R.reason : FL_DIVERGED_ITS;
  if (R.reason == FL_DIVERGED_NANORINF || R.reason == FL_DIVERGED_DTOL || !std::isfinite(R.dp)) h->poisoned = true;  // see solve_cg
  st->rnorm0       = R.rnorm0;
  st->rnorm        = R.dp;
  st->seconds      = ms * 1e-3;
  if (o->history && o->nhistory > 0) {
    const int n = std::min(o->nhistory, R.it + 1);
    FL_HIP(hipMemcpy(o->history, h->hist, sizeof(double) * n, hipMemcpyDeviceToHost));
  }
  return 0;
}

}  // namespace


int fl_apply_tiled(fl_poisson *h, const double *xpad, double *y, int unpadded_y)
{
  const TP tp = tile_plan(h->g);
  launch_apply_pc(h, tp, false, xpad, y, nullptr, nullptr, nullptr, unpadded_y);
  return 0;
}

// r = b - S x, all three unpadded (x is padded into w0 first)
int fl_residual(fl_poisson *h, const double *x, const double *b, double *r)
{
  FL_CHK(fl_ensure_vec(h, &h->w0));
  launch_pad_copy(h->stream, h->g, x, h->w0);
  FL_CHK(fl_fill_ghosts(h, h->w0));
  const TP tp = tile_plan(h->g);
  launch_apply_pc(h, tp, false, h->w0, r, b, nullptr, nullptr, 2);
  return 0;
}

namespace {
template <int RY, int NW, int MODE>
void bcgs_st_t(fl_poisson *h, const PlanA &p, bool jac, const double *stg, const double *e0, const double *e1, const double *e2, double *w0, double *w1, StAux ax = StAux())
{
  const int  tiles = p.tiles_x * p.tiles_y;
  const dim3 gr(p.nblocks), bl(64 * NW);
  if (jac) hipLaunchKernelGGL((k_bcgs_st<RY, NW, true, MODE>), gr, bl, 0, h->stream, h->g, stg, e0, e1, e2, w0, w1, h->scal, h->partial, h->partial_stride, p.nchunk, p.zc, p.tiles_x, tiles, p.remap, ax);
  else hipLaunchKernelGGL((k_bcgs_st<RY, NW, false, MODE>), gr, bl, 0, h->stream, h->g, stg, e0, e1, e2, w0, w1, h->scal, h->partial, h->partial_stride, p.nchunk, p.zc, p.tiles_x, tiles, p.remap, ax);
}
template <int MODE>
void launch_bcgs_st(fl_poisson *h, const PlanA &p, bool jac, const double *stg, const double *e0, const double *e1, const double *e2, double *w0, double *w1)
{
  switch (p.ry * 10 + p.nw) {
  case 28: bcgs_st_t<2, 8, MODE>(h, p, jac, stg, e0, e1, e2, w0, w1); break;
  case 24: bcgs_st_t<2, 4, MODE>(h, p, jac, stg, e0, e1, e2, w0, w1); break;
  default: bcgs_st_t<1, 4, MODE>(h, p, jac, stg, e0, e1, e2, w0, w1); break;
  }
}
}  // namespace

// ---- padded-vector entry points of the multigrid cycle (fl_mg.hip): no pad / unpad copies, no statistics ----------------

// r = b - S x on padded vectors (x's ghosts are filled here)
int fl_residual_padded(fl_poisson *h, double *xpad, const double *bpad, double *rpad)
{
  FL_CHK(fl_fill_ghosts(h, xpad));
  if (cheb_staged_mode() != 0) {
    launch_bcgs_st<8>(h, plan_cg_A(h->g, 0, 0), false, xpad, bpad, nullptr, nullptr, rpad, nullptr);  // LDS-staged walk
    return 0;
  }
  const TP tp = tile_plan(h->g);
  launch_apply_pc(h, tp, false, xpad, rpad, bpad, nullptr, nullptr, 3);
  return 0;
}

// The coarse right-hand side of a multigrid cycle in one pass: cpad (padded, level hc) = R (b - S x), R = k_mg_restrict's weighted sum over the
// eight children.  Only where every axis is halved, the block is even in every direction and the walk pairs rows and planes the way the
// children pair (two rows per wave, even z chunks); returns 1 where it does not apply (the caller runs the residual and the restriction).
int fl_residual_restrict_padded(fl_poisson *h, double *xpad, const double *bpad, const double *wx, const double *wy, const double *wz, fl_poisson *hc, double *cpad)
{
  const int on = FL_VARIANT(mg_fused_restrict, 1);  // 0: residual and restriction as two passes (A/B runs)
  const GridP &g = h->g, &gc = hc->g;
  const PlanA  p = plan_cg_A(g, 0, 0);
  if (!on || cheb_staged_mode() == 0 || p.ry != 2 || (p.nw != 8 && p.nw != 4) || (p.nchunk > 1 && (p.zc & 1))) return 1;
  if ((g.nx & 1) || (g.ny & 1) || (g.nz & 1) || gc.nx * 2 != g.nx || gc.ny * 2 != g.ny || gc.nz * 2 != g.nz) return 1;
  FL_CHK(fl_fill_ghosts(h, xpad));
  StAux ax;
  ax.csx  = gc.sx;
  ax.csxy = gc.sxy;
  ax.coff = gc.off0;
  if (p.nw == 8) bcgs_st_t<2, 8, 11>(h, p, false, xpad, bpad, wx, wy, cpad, const_cast<double *>(wz), ax);
  else bcgs_st_t<2, 4, 11>(h, p, false, xpad, bpad, wx, wy, cpad, const_cast<double *>(wz), ax);
  return 0;
}

// y = S x on padded vectors (x's ghosts are filled here) and x.y summed over all ranks, on the host
int fl_apply_padded_dot(fl_poisson *h, double *xpad, double *ypad, double *xy)
{
  FL_CHK(fl_fill_ghosts(h, xpad));
  const TP    tp = tile_plan(h->g);
  const PlanA pa = plan_cg_A(h->g, 0, 0);
  FL_CHK(fl_ensure_partials(h, std::max(tp.nblocks, pa.nblocks)));
  int nbl = tp.nblocks;
  if (cheb_staged_mode() != 0) {
    launch_bcgs_st<7>(h, pa, false, xpad, nullptr, nullptr, nullptr, ypad, nullptr);  // LDS-staged walk
    nbl = pa.nblocks;
  } else launch_apply_pc(h, tp, false, xpad, ypad, nullptr, nullptr, h->partial, 0);
  launch_reduce(h->stream, h->partial, nbl, h->partial_stride, 4, h->sums);
  if (h->multi) FL_CHK(h->comm.allreduce(h->stream, h->sums, NSLOT));
  if (!xy) return 0;  // the caller takes x.y from h->sums[2] on the device (no host wait)
  FL_HIP(hipMemcpyAsync(xy, h->sums + 2, sizeof(double), hipMemcpyDeviceToHost, h->stream));
  FL_HIP(hipStreamSynchronize(h->stream));
  return 0;
}

// nu Chebyshev(-Jacobi) steps on the handle's padded work vectors: right-hand side in h->r, initial guess in h->xp (taken as
// zero when guess_zero), the result is left in h->xp (the two x buffers h->xp / h->P0 swap roles as needed).  No convergence
// test, no null-space bookkeeping (a constant in x never reaches a residual: the caller projects once at the end), the host
// never waits.  Spectrum bounds as in fl_solve_cheb.
// mgdots (in: the caller wants the five sums of k_mg_dots over the result and the right-hand side; out: whether h->sums holds them -- only
// when the last sweep is the fused Jacobi kernel, which forms them on its way, fl_cheb2.hip)
// subq / suba: the right-hand side h->r is first updated in place, r -= *suba * subq (padded; suba points to DEVICE memory), on the first step's
// pass (guess_zero only)
int fl_cheb_smooth_padded(fl_poisson *h, int nu, bool jac, bool guess_zero, bool *mgdots, const double *subq, const double *suba)
{
  if (subq && !(guess_zero && nu > 0)) return FL_ERR_ARG_WRONGSTATE;
  const bool want = mgdots && *mgdots;
  if (mgdots) *mgdots = false;
  for (double **v : {&h->r, &h->P0, &h->q, &h->xp}) FL_CHK(fl_ensure_vec(h, v));
  if (nu <= 0) {
    if (guess_zero) FL_CHK(fl_zero_vec(h, h->xp));
    return 0;
  }
  const TP tp = tile_plan(h->g);
  FL_CHK(fl_ensure_partials(h, std::max(tp.nblocks, 1024)));
  const int nhist = nu + 2;
  FL_CHK(fl_ensure_hist(h, nhist));
  hipStream_t  s = h->stream;
  fl_ksp_opts  o;
  fl_ksp_opts_default(&o);
  o.norm_type        = FL_NORM_NONE;
  o.maxit            = nu;
  o.remove_nullspace = 0;
  // PETSc's -mg_levels_ksp_chebyshev_esteig 0,0.1,0,1.1 applied to the bound (experiments: FLUCA_MG_CHEB_LO / _HI, fractions of the bound)
  const char  *elo = variant_env("FLUCA_MG_CHEB_LO"), *ehi = variant_env("FLUCA_MG_CHEB_HI");
  const double flo = elo ? std::atof(elo) : 0.1, fhi = ehi ? std::atof(ehi) : 1.1;
  const double lam = fl_gershgorin_bound(h, jac), emin = flo * lam, emax = fhi * lam;
  init_scal(h, &o);
  KspScal S0 = *h->scal_host;
  S0.scale     = 2. / (emax + emin);
  const double alpha = 1. - S0.scale * emin;
  S0.mu        = 1. / alpha;
  S0.omegaprod = 2. / alpha;
  S0.ckm1      = 1.;
  S0.ck        = S0.mu;
  S0.cheb_rho  = 0.;
  S0.cheb_c    = S0.scale;
  const int fuse_ok = cheb_fuse(h);
  if (fuse_ok < 0) return fuse_ok;
  const bool fuse = fuse_ok && nu - (guess_zero ? 1 : 0) >= 2;
  Cheb2Plan  cp{};
  if (fuse) {
    cp = fl_cheb2_plan(h->g);
    FL_CHK(fl_ensure_vec(h, &h->cd1));
    FL_CHK(fl_ensure_partials(h, cp.nblocks));
  }
  double    *X0 = h->xp, *X1 = h->P0, *B = h->r, *D0 = h->q, *D1 = fuse ? h->cd1 : h->q;
  const bool ghosts = fl_any_ghost_exchange(h);
  int        cur = 0, dcur = 0;
  const int trace = knob(K_comm_trace);
  auto mark = [&](const char *what, int j) {  // debugging aid: where a sweep stops making progress (each mark waits for the stream)
    if (!trace) return;
    const bool sync = trace >= 2;
    const hipError_t e = sync ? hipStreamSynchronize(s) : hipSuccess;
    struct timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    std::fprintf(stderr, "[%.3f smooth r%d n=%dx%dx%d] %s j=%d cur=%d dcur=%d -> %s\n", ts.tv_sec % 1000 + 1e-9 * ts.tv_nsec, h->comm.rank, h->g.nx, h->g.ny, h->g.nz, what, j, cur, dcur, hipGetErrorString(e));
    std::fflush(stderr);
  };
  // The smoother looks at no norm and carries no null space: what k_cheb_fin / k_cheb_fin2 would do between two sweeps -- flip the buffers, count
  // the step, advance the Chebyshev recurrence -- depends on nothing the device computes.  So the scalar block of EVERY sweep of the call is formed
  // here, once per (nu, first-step form, fused pattern, interval) and handle, kept on the device, and each sweep is launched on its own block: no scalar
  // kernel between the sweeps (a 512^3 multigrid solve made 330 of them).  `kind`: 1 the first step from a zero guess, 2 a fused pair, 0 a single step.
  // 3 (round 5): the first THREE steps from a zero guess in one sweep -- the first step has no stencil, so the fused kernel forms it where it would
  // load x (fl_cheb2.hip, Z); one rank only (the ring of b two deep is not exchanged)
  const bool zero3 = fuse && guess_zero && nu >= 3 && !h->multi && knob(K_cheb_zero3) != 0;
  auto kind_of = [&](int j) { return (j == 0 && guess_zero) ? (zero3 ? 3 : 1) : ((fuse && j + 2 <= nu && !(want && jac && !h->multi && ((nu - j) & 1))) ? 2 : 0); };
  const KspScal *seq = nullptr;
  {
    SmoothSeq *hit = nullptr;
    for (SmoothSeq &q : h->smooth_seq)
      if (q.nu == nu && q.guess_zero == guess_zero && q.jac == jac && q.fuse == fuse && q.want == want && q.zero3 == zero3 && q.emin == emin && q.emax == emax) hit = &q;
    if (!hit) {
      SmoothSeq q;
      q.nu = nu; q.guess_zero = guess_zero; q.jac = jac; q.fuse = fuse; q.want = want; q.zero3 = zero3; q.emin = emin; q.emax = emax;
      KspScal S = S0;
      auto advance = [](const KspScal &P, double ck, double ckm1, double &ck_out, double &ckm1_out, double &rho, double &c) {  // cheb_advance, on the host
        const double ckp1 = 2. * P.mu * ck - ckm1, omega = P.omegaprod * ck / ckp1;
        ckm1_out = ck;
        ck_out   = ckp1;
        rho      = omega - 1.;
        c        = omega * P.scale;
      };
      for (int j = 0; j < nu;) {
        q.host.push_back(S);
        if (kind_of(j) == 3) {  // one flip of each buffer pair, three steps of the recurrence
          double ck, ckm1, ck2, ckm2, rho, c;
          advance(S, S.ck, S.ckm1, ck, ckm1, rho, c);
          advance(S, ck, ckm1, ck2, ckm2, rho, c);
          S.cur ^= 1;
          S.dcur ^= 1;
          S.it += 3;
          advance(S, ck2, ckm2, S.ck, S.ckm1, S.cheb_rho, S.cheb_c);
          j += 3;
        } else if (kind_of(j) == 2) {  // k_cheb_fin2: both steps accepted
          double ck, ckm1, rho1, c1, rho2, c2;
          advance(S, S.ck, S.ckm1, ck, ckm1, rho1, c1);
          S.cur ^= 1;
          S.dcur ^= 1;
          S.it += 2;
          advance(S, ck, ckm1, S.ck, S.ckm1, rho2, c2);
          S.cheb_rho = rho2;
          S.cheb_c   = c2;
          j += 2;
        } else {  // k_cheb_fin
          double ck, ckm1, rho, c;
          S.cur ^= 1;
          S.it += 1;
          advance(S, S.ck, S.ckm1, ck, ckm1, rho, c);
          S.ck       = ck;
          S.ckm1     = ckm1;
          S.cheb_rho = rho;
          S.cheb_c   = c;
          j += 1;
        }
      }
      FL_HIP(hipMalloc((void **)&q.dev, sizeof(KspScal) * q.host.size()));
      h->smooth_seq.push_back(std::move(q));
      hit = &h->smooth_seq.back();
      // (the host copy lives as long as the handle: the asynchronous upload may read it whenever it likes)
      FL_HIP(hipMemcpyAsync(hit->dev, hit->host.data(), sizeof(KspScal) * hit->host.size(), hipMemcpyHostToDevice, s));
    }
    seq = hit->dev;
  }
  // the launchers take the scalar block from h->scal: point it at the sweep's own block for the length of a launch
  struct ScalGuard {
    fl_poisson *h;
    KspScal    *keep;
    ~ScalGuard() { h->scal = keep; }
  } guard{h, h->scal};
  // several ranks under the fused sweep: its ring comes from the ghost layers -- two of x with the shell's edges, one of d, one of b (b once
  // per call, after the first step has possibly updated it in place)
  const bool deep = fuse && h->multi;
  bool       bghost = false;
  int        launch = 0;
  for (int j = 0; j < nu;) {
    h->scal = const_cast<KspScal *>(seq) + launch++;
    const int kind = kind_of(j);
    if (kind == 3) {
      double *Bw = nullptr;
      if (subq) {
        FL_CHK(fl_ensure_vec(h, &h->rb));
        Bw = h->rb;
      }
      fl_launch_cheb2_from_zero(h, cp, jac, X0, X1, B, D0, D1, subq, suba, Bw);
      if (subq) {  // the updated right-hand side lives in the other array from here on
        std::swap(h->r, h->rb);
        B = h->r;
      }
      dcur ^= 1;
      j += 3;
    } else if (kind == 1) {
      const int64_t items = (int64_t)((h->g.nx + 127) / 128) * h->g.ny * h->g.nz;  // 128-cell row segments, one per wave and trip
      const int     nb    = (int)std::max<int64_t>(1, std::min<int64_t>((items + 3) / 4, 8192));
      if (subq) {
        if (jac) hipLaunchKernelGGL((k_cheb_first<true, true>), dim3(nb), dim3(256), 0, s, h->g, X0, X1, B, D0, D1, h->scal, subq, suba);
        else hipLaunchKernelGGL((k_cheb_first<false, true>), dim3(nb), dim3(256), 0, s, h->g, X0, X1, B, D0, D1, h->scal, subq, suba);
      } else if (jac) hipLaunchKernelGGL((k_cheb_first<true>), dim3(nb), dim3(256), 0, s, h->g, X0, X1, B, D0, D1, h->scal, (const double *)nullptr, (const double *)nullptr);
      else hipLaunchKernelGGL((k_cheb_first<false>), dim3(nb), dim3(256), 0, s, h->g, X0, X1, B, D0, D1, h->scal, (const double *)nullptr, (const double *)nullptr);
      j += 1;
    } else if (kind == 2) {  // (asked for the sums: an odd step count takes its single step first, so that a fused sweep ends the call)
      // two steps in one sweep; on one rank it reads no ghost layer (fl_cheb2.hip)
      const bool md = want && jac && j + 2 == nu && !h->multi;
      if (deep) {
        if (!bghost) FL_CHK(fl_fill_ghosts(h, B));
        bghost = true;
        FL_CHK(fl_fill_ghosts_deep(h, cur ? X1 : X0));
        FL_CHK(fl_fill_ghosts(h, dcur ? D1 : D0));
      }
      mark("ghosts filled, before fused", j);
      fl_launch_cheb2(h, cp, jac, X0, X1, B, D0, D1, md);
      mark("fused done", j);
      dcur ^= 1;
      if (md) {
        launch_reduce(s, h->partial, cp.nblocks, h->partial_stride, 5, h->sums);
        *mgdots = true;
      }
      j += 2;
    } else {
      if (ghosts) FL_CHK(fl_fill_ghosts(h, cur ? X1 : X0));
      mark("before single", j);
      (void)launch_cheb(h, tp, jac, X0, X1, B, D0, D1);
      mark("single done", j);
      j += 1;
    }
    cur ^= 1;
  }
  if (cur) std::swap(h->xp, h->P0);
  h->poisoned = true;  // no norm is monitored in here: see fl_solve_cheb
  return 0;
}


int fl_solve_bcgs(fl_poisson *h, const double *b, double *x, const fl_ksp_opts *o, fl_ksp_stats *st)
{
  // variant 0 (default): V0 = M S P and T0 = M S S0 are formed wherever they are needed and never stored (k_bcgs_st, 120 B/cell/iteration);
  // any other value: the stored products of round 1 (k_apply_pc + k_bcgs_pw, 152 B/cell)
  const int variant_forced = FL_VARIANT(bcgs_variant, -1);
  const int variant = (o->variant == 0 && variant_forced >= 0) ? variant_forced : o->variant;
  if (variant == 0) {
    const GridP &g   = h->g;
    const bool   jac = o->pc == FL_PC_JACOBI;
    // vectors: r=R, P0=RP, P1/q=P (two buffers: a tile reads its neighbours' old P while they write the new one), xp=X, w0=S0
    for (double **v : {&h->r, &h->P0, &h->P1, &h->q, &h->xp, &h->w0}) FL_CHK(fl_ensure_vec(h, v));
    const TP    tp = tile_plan(g);
    const PlanA pa = plan_cg_A(g, 0, 0);
    FL_CHK(fl_ensure_partials(h, std::max(tp.nblocks, pa.nblocks)));
    const int nhist = o->maxit + 1;
    FL_CHK(fl_ensure_hist(h, nhist));
    hipStream_t s = h->stream;
    init_scal(h, o);
    FL_HIP(hipEventRecord(h->ev0, s));
    FL_HIP(hipMemcpyAsync(h->scal, h->scal_host, sizeof(KspScal), hipMemcpyHostToDevice, s));
    for (double *v : {h->P1, h->q, h->xp}) FL_CHK(fl_zero_vec(h, v));
    double *R = h->r, *RP = h->P0, *P = h->P1, *Pn = h->q, *X = h->xp, *S0 = h->w0;
    auto    fin = [&](int mode) {
      return [=](const double *partial, int nb, int stride, const double *sums) { hipLaunchKernelGGL(k_bcgs_fin, dim3(1), dim3(256), 0, s, mode, partial, nb, stride, sums, h->scal, h->hist, nhist); };
    };
    launch_pw<3>(h, tp, jac, b, nullptr, nullptr, nullptr, R, nullptr);
    FL_CHK(fin_step(h, tp.nblocks, 3, fin(0)));
    launch_pw<4>(h, tp, jac, nullptr, nullptr, nullptr, nullptr, R, RP);
    const bool ghosts = fl_any_ghost_exchange(h);
    const int  every  = o->check_every > 0 ? o->check_every : 16;
    int        it = 0;
    bool       done = false;
    while (!done) {
      const int stop = std::min(o->maxit, it + every);
      for (; it < stop; ++it) {
        // P' = R - omega_old beta (M S P - vshift) + beta P   (the old P's ghosts are still those filled for the iteration before)
        launch_bcgs_st<5>(h, pa, jac, P, R, nullptr, nullptr, Pn, nullptr);
        std::swap(P, Pn);
        if (ghosts) FL_CHK(fl_fill_ghosts(h, P));
        launch_bcgs_st<1>(h, pa, jac, P, RP, nullptr, nullptr, nullptr, nullptr);
        FL_CHK(fin_step(h, pa.nblocks, 4, fin(1)));
        launch_bcgs_st<2>(h, pa, jac, P, R, nullptr, nullptr, S0, nullptr);
        FL_CHK(fin_step(h, pa.nblocks, 3, fin(2)));
        if (ghosts) FL_CHK(fl_fill_ghosts(h, S0));
        launch_bcgs_st<3>(h, pa, jac, S0, nullptr, nullptr, nullptr, nullptr, nullptr);
        FL_CHK(fin_step(h, pa.nblocks, 4, fin(3)));
        launch_bcgs_st<4>(h, pa, jac, S0, P, X, RP, X, R);
        FL_CHK(fin_step(h, pa.nblocks, 3, fin(4)));
      }
      FL_CHK(fl_poll_scal(h));
      if (h->scal_host->reason != 0 || it >= o->maxit) done = true;
    }
    launch_unpad_copy(s, g, X, x, nullptr);
    return finish_stats(h, o, st);
  }
#ifndef FL_KBENCH_VARIANTS
  return FL_ERR_SUP;  // the stored-product form of round 1 lives in the kbench build only (include/fluca_hip.h, fl_ksp_opts.variant)
#else
  const GridP &g   = h->g;
  const bool   jac = o->pc == FL_PC_JACOBI;
  // vectors: r=R, P0=RP, P1=P, q=V0, xp=X, w0=S0, w1=T0
  for (double **v : {&h->r, &h->P0, &h->P1, &h->q, &h->xp, &h->w0, &h->w1}) FL_CHK(fl_ensure_vec(h, v));
  const TP tp = tile_plan(g);
  FL_CHK(fl_ensure_partials(h, tp.nblocks));
  const int nhist = o->maxit + 1;
  FL_CHK(fl_ensure_hist(h, nhist));
  hipStream_t s = h->stream;
  init_scal(h, o);
  FL_HIP(hipEventRecord(h->ev0, s));
  FL_HIP(hipMemcpyAsync(h->scal, h->scal_host, sizeof(KspScal), hipMemcpyHostToDevice, s));
  for (double *v : {h->P1, h->q, h->xp}) FL_CHK(fl_zero_vec(h, v));
  double *R = h->r, *RP = h->P0, *P = h->P1, *V0 = h->q, *X = h->xp, *S0 = h->w0, *T0 = h->w1;
  auto    fin = [&](int mode) {
    return [=](const double *partial, int nb, int stride, const double *sums) { hipLaunchKernelGGL(k_bcgs_fin, dim3(1), dim3(256), 0, s, mode, partial, nb, stride, sums, h->scal, h->hist, nhist); };
  };
  launch_pw<3>(h, tp, jac, b, nullptr, nullptr, nullptr, R, nullptr);
  FL_CHK(fin_step(h, tp.nblocks, 3, fin(0)));
  launch_pw<4>(h, tp, jac, nullptr, nullptr, nullptr, nullptr, R, RP);
  const bool ghosts = fl_any_ghost_exchange(h);
  const int  every  = o->check_every > 0 ? o->check_every : 16;
  int        it = 0;
  bool       done = false;
  while (!done) {
    const int stop = std::min(o->maxit, it + every);
    for (; it < stop; ++it) {
      launch_pw<0>(h, tp, jac, R, V0, nullptr, nullptr, P, nullptr);
      if (ghosts) FL_CHK(fl_fill_ghosts(h, P));
      launch_apply_pc(h, tp, jac, P, V0, RP, h->scal, h->partial, 0);
      FL_CHK(fin_step(h, tp.nblocks, 4, fin(1)));
      launch_pw<1>(h, tp, jac, R, V0, nullptr, nullptr, S0, nullptr);
      FL_CHK(fin_step(h, tp.nblocks, 3, fin(2)));
      if (ghosts) FL_CHK(fl_fill_ghosts(h, S0));
      launch_apply_pc(h, tp, jac, S0, T0, nullptr, h->scal, h->partial, 0);
      FL_CHK(fin_step(h, tp.nblocks, 4, fin(3)));
      launch_pw<2>(h, tp, jac, P, S0, T0, RP, X, R);
      FL_CHK(fin_step(h, tp.nblocks, 3, fin(4)));
    }
    FL_CHK(fl_poll_scal(h));
    if (h->scal_host->reason != 0 || it >= o->maxit) done = true;
  }
  launch_unpad_copy(s, g, X, x, nullptr);
  return finish_stats(h, o, st);
#endif
}

// KSPCG with -ksp_cg_single_reduction (fl_ksp_opts.cg_single_reduction): one reduction point, hence one all-reduce and one scalar
// kernel per iteration on several ranks (two of each in the default pair).  Vectors: r / q = the two r buffers, P0 = p, P1 = W, xp = x.
int fl_solve_cg_sr(fl_poisson *h, const double *b, double *x, const fl_ksp_opts *o, fl_ksp_stats *st)
{
  const GridP &g   = h->g;
  const bool   jac = o->pc == FL_PC_JACOBI;
  for (double **v : {&h->r, &h->P0, &h->P1, &h->q, &h->xp}) FL_CHK(fl_ensure_vec(h, v));
  const PlanA pa = plan_cg_A(g, 0, 0);
  FL_CHK(fl_ensure_partials(h, pa.nblocks));
  const int nhist = o->maxit + 1;
  FL_CHK(fl_ensure_hist(h, nhist));
  hipStream_t s = h->stream;
  init_scal(h, o);
  FL_HIP(hipEventRecord(h->ev0, s));
  FL_HIP(hipMemcpyAsync(h->scal, h->scal_host, sizeof(KspScal), hipMemcpyHostToDevice, s));
  // p, W, x enter the first update with factor b = 0 / as the zero initial guess: they must be finite, so they are cleared; the ghost
  // layers of both r buffers likewise (a wall ghost is divided by its infinite diagonal, whatever finite value it holds)
  for (double *v : {h->r, h->q, h->P0, h->P1, h->xp}) FL_CHK(fl_zero_vec(h, v));
  h->poisoned = false;  // every work vector a poisoned handle would have cleared has just been cleared
  double *R = h->r, *Rn = h->q, *P = h->P0, *W = h->P1, *X = h->xp;
  launch_pad_copy(s, g, b, R);
  const bool ghosts = fl_any_ghost_exchange(h);
  if (ghosts) FL_CHK(fl_fill_ghosts(h, R));
  auto fin = [&](int mode) {
    return [=](const double *partial, int nb, int stride, const double *sums) { hipLaunchKernelGGL(k_cgsr_fin, dim3(1), dim3(256), 0, s, mode, partial, nb, stride, sums, h->scal, h->hist, nhist); };
  };
  // several ranks: the ghost exchange of the new residual runs on a second stream behind the update kernel (MODE 10), as the pair hides it behind
  // k_cg_Bq -- MODE 9 keeps S = A z on the block's boundary layers in the r buffer that is dead until MODE 10 refills it, and the pack forms
  // r - a (S + b W) there.  "overlap" = 0 restores the sequential order (A/B runs).  (Periodic axes inside the block are wrapped at the end.)
  const bool overlap_env = knob(K_overlap) != 0;
  const bool overlap = h->multi && overlap_env;
  launch_bcgs_st<9>(h, pa, jac, R, nullptr, nullptr, nullptr, overlap ? Rn : nullptr, nullptr);
  FL_CHK(fin_step(h, pa.nblocks, 7, fin(0)));
  const int every = o->check_every > 0 ? o->check_every : 16;
  int       it = 0;
  bool      done = false;
  while (!done) {
    const int stop = std::min(o->maxit, it + every);
    for (; it < stop; ++it) {
      if (overlap) FL_CHK(fl_exchange_sr_begin(h, R, Rn, W, Rn));  // packs before MODE 10 overwrites the kept S with the new residual
      launch_bcgs_st<10>(h, pa, jac, R, P, W, X, Rn, nullptr);
      std::swap(R, Rn);
      if (overlap) FL_CHK(fl_exchange_r_end(h, R));
      else if (ghosts) FL_CHK(fl_fill_ghosts(h, R));
      launch_bcgs_st<9>(h, pa, jac, R, nullptr, nullptr, nullptr, overlap ? Rn : nullptr, nullptr);
      FL_CHK(fin_step(h, pa.nblocks, 7, fin(1)));
    }
    FL_CHK(fl_poll_scal(h));
    if (h->scal_host->reason != 0 || it >= o->maxit) done = true;
  }
  launch_unpad_copy(s, g, X, x, nullptr);
  return finish_stats(h, o, st);
}

// Upper bound of the spectrum of M S from the GLOBAL 1-D tables (identical on every rank):
//   Jacobi: sum_d a_d / sum_d c_d <= max_d max_i a_d(i)/c_d(i)   (mediant inequality), a = |sl|+|sc|+|sh|, c = |sc|
//   none  : sum_d max_i a_d(i)
// For the uniform Neumann / periodic Laplacian this is the exact Gershgorin value 2.
double fl_gershgorin_bound(const fl_poisson *h, bool jac)
{
  double lam = 0.;
  for (int d = 0; d < 3; ++d) {
    const Axis &A = h->ax[d];
    double      md = 0.;
    for (int64_t i = 0; i < A.n; ++i) {
      const double a = std::fabs(A.sl[i]) + std::fabs(A.sc[i]) + std::fabs(A.sh[i]);
      md             = std::max(md, jac ? a / std::fabs(A.sc[i]) : a);
    }
    lam = jac ? std::max(lam, md) : lam + md;
  }
  return lam;
}

int fl_solve_cheb(fl_poisson *h, const double *b, double *x, const fl_ksp_opts *o, fl_ksp_stats *st)
{
  const GridP &g   = h->g;
  const bool   jac = o->pc == FL_PC_JACOBI;
  if (o->norm_type == FL_NORM_NATURAL) return FL_ERR_SUP;
  // vectors: xp=X0, P0=X1, r=b (padded), q=d
  for (double **v : {&h->r, &h->P0, &h->q, &h->xp}) FL_CHK(fl_ensure_vec(h, v));
  const TP tp = tile_plan(g);
  FL_CHK(fl_ensure_partials(h, std::max(tp.nblocks, 1024)));
  const int nhist = o->maxit + 2;
  FL_CHK(fl_ensure_hist(h, nhist));
  hipStream_t s = h->stream;
  double emin = o->emin, emax = o->emax;
  if (emin == 0. && emax == 0.) {
    const double lam = fl_gershgorin_bound(h, jac);
    emin = 0.1 * lam;  // -ksp_chebyshev_esteig 0,0.1,0,1.1 applied to the bound
    emax = 1.1 * lam;
  }
  FL_CHK(fl_cheb_begin(h, o, emin, emax));
  KspScal &S = *h->scal_host;
  // X1 (P0) is fully written by the first step before anything reads it; its wall ghosts are zero since allocation
  for (double *v : {h->q, h->xp}) FL_CHK(fl_zero_vec(h, v));
  launch_pad_copy(s, g, b, h->r);
  const int  every  = o->check_every > 0 ? o->check_every : 16;
  const int  total  = o->norm_type == FL_NORM_NONE ? o->maxit : o->maxit + 1;  // with a norm, launch maxit is only the final check
  // without a convergence test between the steps two of them share one sweep over memory (fl_cheb2.hip)
  const int fuse_ok = cheb_fuse(h);
  if (fuse_ok < 0) return fuse_ok;
  const bool fuse = o->norm_type == FL_NORM_NONE && total >= 2 && fuse_ok;
  Cheb2Plan  cp{};
  if (fuse) {
    cp = fl_cheb2_plan(g);
    FL_CHK(fl_ensure_vec(h, &h->cd1));
    FL_CHK(fl_ensure_partials(h, cp.nblocks));
  }
  double    *X0 = h->xp, *X1 = h->P0, *B = h->r, *D0 = h->q, *D1 = fuse ? h->cd1 : h->q;
  const bool ghosts = fl_any_ghost_exchange(h);
  auto       finl   = [=](const double *partial, int nb, int stride, const double *sums) { hipLaunchKernelGGL(k_cheb_fin, dim3(1), dim3(256), 0, s, partial, nb, stride, sums, h->scal, h->hist, nhist); };
  auto       fin2   = [=](const double *partial, int nb, int stride, const double *sums) { hipLaunchKernelGGL(k_cheb_fin2, dim3(1), dim3(256), 0, s, partial, nb, stride, sums, h->scal); };
  int        j = 0, hostcur = 0, hostdcur = 0, nprof = 0;
  const bool deep = fuse && h->multi;  // the fused sweep's ring comes from ghost layers: see fl_cheb_smooth_padded
  if (deep) FL_CHK(fl_fill_ghosts(h, B));
  bool       done = total <= 0;
  ProfEvents prof;  // profile = 1: HIP events around every launch of the dominant kernel (the fused sweep where it is used)
  if (o->profile) FL_CHK(prof.create(2 * (size_t)std::min(total, 2048)));
  // check_every < 0 with KSP_NORM_NONE: a smoother call -- exactly maxit steps, nothing to test, so the host never waits for
  // the device (the buffer that holds the answer follows from the step count) and no statistics are gathered
  const bool nopoll = o->check_every < 0 && o->norm_type == FL_NORM_NONE;
  while (!done) {
    const int stop = std::min(total, j + every);
    while (j < stop) {
      if (fuse && j + 2 <= total) {
        const bool pr = (size_t)(2 * nprof + 1) < prof.ev.size();
        if (deep) {
          FL_CHK(fl_fill_ghosts_deep(h, hostcur ? X1 : X0));
          FL_CHK(fl_fill_ghosts(h, hostdcur ? D1 : D0));
        }
        if (pr) FL_HIP(hipEventRecord(prof.ev[2 * nprof], s));
        fl_launch_cheb2(h, cp, jac, X0, X1, B, D0, D1);
        hostdcur ^= 1;
        if (pr) FL_HIP(hipEventRecord(prof.ev[2 * nprof++ + 1], s));
        FL_CHK(fin_step(h, cp.nblocks, 6, fin2));
        j += 2;
      } else {
        if (ghosts && j > 0) FL_CHK(fl_fill_ghosts(h, hostcur ? X1 : X0));
        const bool pr = !fuse && (size_t)(2 * nprof + 1) < prof.ev.size();
        if (pr) FL_HIP(hipEventRecord(prof.ev[2 * nprof], s));
        const int nbc = launch_cheb(h, tp, jac, X0, X1, B, D0, D1);
        if (pr) FL_HIP(hipEventRecord(prof.ev[2 * nprof++ + 1], s));
        FL_CHK(fin_step(h, nbc, 3, finl));
        j += 1;
      }
      hostcur ^= 1;
    }
    if (nopoll) done = j >= total;
    else {
      FL_CHK(fl_poll_scal(h));
      if (h->scal_host->reason != 0 || j >= total) done = true;
    }
  }
  // without a monitored norm nothing tells whether NaN / Inf went through the work vectors CG shares with this solver (P0 = X1, q = d,
  // xp = X0): the next CG solve on the handle clears its direction buffers instead of relying on "beta = 0 times something finite"
  if (o->norm_type == FL_NORM_NONE) h->poisoned = true;
  if (nopoll) {
    launch_unpad_copy(s, g, hostcur ? X1 : X0, x, S.nullspace ? &h->scal->xshift : nullptr);
    st->iters  = total;
    st->reason = FL_CONVERGED_ITS;
    st->rnorm0 = st->rnorm = st->seconds = 0.;  // nothing was measured: the host never waited
    return 0;
  }
  FL_CHK(fl_poll_scal(h));
  const KspScal &R = *h->scal_host;
  // answer: buffer `cur`, minus the accumulated constant (null-space removal done lazily)
  launch_unpad_copy(s, g, R.cur ? X1 : X0, x, R.nullspace ? &h->scal->xshift : nullptr);
  FL_CHK(finish_stats(h, o, st));
  st->kernel_ms       = 0.;
  st->kernel_launches = 0;
  if (o->profile) prof.mean(nprof, &st->kernel_ms, &st->kernel_launches);  // fused: per launch = per TWO steps
  return 0;
}

// ------------------------------------------------------------------------------------------------ shared with fl_momentum.hip
// The BiCGStab scalar recurrences do not care which operator produced the inner products: the momentum solve reuses them.

int fl_ksp_begin(fl_poisson *h, const fl_ksp_opts *o)
{
  init_scal(h, o);
  FL_HIP(hipEventRecord(h->ev0, h->stream));
  FL_HIP(hipMemcpyAsync(h->scal, h->scal_host, sizeof(KspScal), hipMemcpyHostToDevice, h->stream));
  return 0;
}

int fl_bcgs_fin_step(fl_poisson *h, int mode, int nblocks, int nslot, int nhist)
{
  hipStream_t s = h->stream;
  return fin_step(h, nblocks, nslot, [=](const double *partial, int nb, int stride, const double *sums) {
    hipLaunchKernelGGL(k_bcgs_fin, dim3(1), dim3(256), 0, s, mode, partial, nb, stride, sums, h->scal, h->hist, nhist);
  });
}

int fl_ksp_finish(fl_poisson *h, const fl_ksp_opts *o, fl_ksp_stats *st) { return finish_stats(h, o, st); }

// KSPCHEBYSHEV's scalars for the interval [emin, emax] (PETSc's recurrence: scale = 2 / (emax + emin), mu = 1 / (1 - scale emin),
// c_0 = 1, c_1 = mu, omega_k = (2 mu) c_k / c_{k+1}); step 0 is x_1 = x_0 + scale z_0.  The momentum solve shares them.
int fl_cheb_begin(fl_poisson *h, const fl_ksp_opts *o, double emin, double emax)
{
  init_scal(h, o);
  KspScal &S = *h->scal_host;
  S.scale     = 2. / (emax + emin);
  const double alpha = 1. - S.scale * emin;
  S.mu        = 1. / alpha;
  S.omegaprod = 2. / alpha;
  S.ckm1      = 1.;
  S.ck        = S.mu;
  S.cheb_rho  = 0.;        // step 0: d_1 = scale * z_0
  S.cheb_c    = S.scale;
  FL_HIP(hipEventRecord(h->ev0, h->stream));
  hipLaunchKernelGGL(k_scal_set, dim3(1), dim3(1), 0, h->stream, h->scal, S);  // by value: the host copy may be refilled at once
  return 0;
}

int fl_cheb_fin_step(fl_poisson *h, int nblocks, int nhist)
{
  hipStream_t s = h->stream;
  return fin_step(h, nblocks, 3, [=](const double *partial, int nb, int stride, const double *sums) {
    hipLaunchKernelGGL(k_cheb_fin, dim3(1), dim3(256), 0, s, partial, nb, stride, sums, h->scal, h->hist, nhist);
  });
}
