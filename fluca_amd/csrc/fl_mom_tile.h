// fl_mom_tile.h -- k_mom2: the momentum block  y = [1/diag] (cI + cC C + cL L) x  on 128 x NW tiles, two x-adjacent cells per lane.
//
//   ComputeVelocityLaplacianOperator_Private   cnlinearcart3d.c:425-632      L: one 1-D second-derivative row per axis
//   ComputeConvectionOperator_Private          cnlinearcart3d.c:873-1294     (C v)_c = 1/2 d/dx_d (v_c V0_d + v0interp_c v_d)
//   NSFormJacobian_CNLinear_Cart3d_Internal    cnlinearcart3d.c:2930-2941    A = I + dt C - (mu dt / 2 rho) L
//
// The same arithmetic per cell as k_mom_apply (mom_axis), in the shape of k_cg_A:
//  * a wave owns one row of 128 cells, a lane two x-adjacent cells: every stream (3 velocity components, 12 face fields, 3 outputs)
//    moves with 16-byte accesses, one 1 KiB wave-instruction per row and plane;
//  * ONE register set per stream: a stream's load for the next plane is issued as soon as the phase that consumes the current
//    plane's values is over (y-faces, x-faces, then the z-march registers), in the order the next plane needs them, so a whole plane
//    of the tile (15 x 16 B per lane) is in flight all the time and no wait ever drains younger loads;
//  * x/y neighbours of the velocity and the high y-face come from LDS (double-buffered, one barrier per plane), the high x-face of a
//    lane's second cell from the next lane (DPP), z neighbours ride in registers; the tile's one-cell ring (rows -1 / NW, columns
//    -1 / 128, the face row / column behind the tile) is fetched by 8-byte loads shared out over the block's threads;
//  * 1-D table numbers: x from LDS (they differ per cell), y wave-uniform in SGPRs, z scalar loads per plane.
#pragma once

namespace fl {

template <int NW>
struct Mom2Lds {
  static constexpr int TX = 128, TY = NW, LXU = TX + 4;
  double u[2][3][TY + 2][LXU];  // velocity plane incl. its ring: column ii = -1..128 at index ii + 2, row jj = -1..TY at jj + 1
  double fy[2][4][TY + 1][TX];  // low y-faces of rows 0..TY (row TY = high face of the tile's last row)
  double fxe[2][4][TY];         // low x-face of column TX (= high face of the tile's last column)
  double tabx[MOM_NTAB][TX];    // x-axis table numbers of the tile's columns
  double red[4 * NW];
};

__device__ __forceinline__ double2 LD2(const double *base, unsigned byteoff) { return *reinterpret_cast<const double2 *>(reinterpret_cast<const char *>(base) + byteoff); }
__device__ __forceinline__ void    ST2(double *base, unsigned byteoff, double2 v) { *reinterpret_cast<double2 *>(reinterpret_cast<char *>(base) + byteoff) = v; }
template <int NT>
__device__ __forceinline__ void ST2nt(double *base, unsigned byteoff, double2 v)
{
  st2<NT>(reinterpret_cast<double *>(reinterpret_cast<char *>(base) + byteoff), v);
}

// lane l receives v of lane l + 1, lane 63 receives fill (DPP wave_shl:1)
__device__ __forceinline__ double from_next_lane(double v, double fill)
{
  const int lo = __builtin_amdgcn_update_dpp(__double2loint(fill), __double2loint(v), 0x130, 0xf, 0xf, false);
  const int hi = __builtin_amdgcn_update_dpp(__double2hiint(fill), __double2hiint(v), 0x130, 0xf, 0xf, false);
  return __hiloint2double(hi, lo);
}

// ---- the rows of A in coefficient form ------------------------------------------------------------------------------------------
// Along one axis D the row of component c is (mom_axis, term by term)
//   cC [ vl (Il . u_c) + vh (Ih . u_c) + wl_c (Nl . u_D) + wh_c (Nh . u_D) ] + cL (L . u_c)
// with vl / vh = V0_D on the low / high face, wl_c / wh_c = v0interp_{c,D} there, Il / Ih the face-interpolation rows of the
// component's rule (tangential T, or normal N when c == D) and Nl / Nh those of the face-normal component.  V0 is the same for the three
// components, so the entries on u_c's own column -- am = vl Il0 + vh Ih0 + L0, ac, ap -- are formed ONCE per cell and axis (per rule)
// and the coupling to the face-normal component through G = N . u_D once: 23 multiply-adds per cell and axis on the fast path where
// the term-by-term form takes 42.  The table numbers arrive pre-scaled (k_mom_scale_tab: second-derivative rows times cL, convection
// rows times cC), fast-path numbers first: t[0..6] = L0 L1 L2 Tl0 Tl1 Th1 Th2.
typedef const double __attribute__((address_space(4))) cdouble4;  // constant address space: uniform addresses become scalar loads
__device__ __forceinline__ const cdouble4 *as_const(const double *p) { return (const cdouble4 *)(uintptr_t)p; }
constexpr int MOM_STAB = 32;  // doubles per cell of a scaled table: 7 fast-path numbers, pad, the 20 general numbers at 8.., pad

// pair arithmetic: a lane's two cells at once
__device__ __forceinline__ double2 fma2(double2 a, double2 b, double2 c) { return make_double2(fma(a.x, b.x, c.x), fma(a.y, b.y, c.y)); }
__device__ __forceinline__ double2 fma2(double2 a, double b, double2 c) { return make_double2(fma(a.x, b, c.x), fma(a.y, b, c.y)); }
__device__ __forceinline__ double2 fma2(double2 a, double b, double c) { return make_double2(fma(a.x, b, c), fma(a.y, b, c)); }
__device__ __forceinline__ double2 mul2(double2 a, double2 b) { return make_double2(a.x * b.x, a.y * b.y); }
__device__ __forceinline__ double2 mul2(double2 a, double b) { return make_double2(a.x * b, a.y * b); }

// fast path, in two stages so that only one component's neighbours are live at a time:
// the entries shared by the three components (am, ac, ap on the component's own column; gl, gh = N . u_D on the low / high face) ...
struct MomRow {
  double2 am, ac, ap, gl, gh;
};
// ... with per-cell table numbers t (x axis) or wave-uniform ones (y, z axes); nm / nc / np = the face-normal component u_D
__device__ __forceinline__ MomRow mom_row_coef(const double2 (&t)[7], double2 vl, double2 vh, double2 nm, double2 nc, double2 np)
{
  MomRow r;
  r.am = fma2(vl, t[3], t[0]);
  r.ac = fma2(vh, t[5], fma2(vl, t[4], t[1]));
  r.ap = fma2(vh, t[6], t[2]);
  r.gl = fma2(t[4], nc, mul2(t[3], nm));
  r.gh = fma2(t[6], np, mul2(t[5], nc));
  return r;
}
__device__ __forceinline__ MomRow mom_row_coef(const double (&t)[7], double2 vl, double2 vh, double2 nm, double2 nc, double2 np)
{
  MomRow r;
  r.am = fma2(vl, t[3], t[0]);
  r.ac = fma2(vh, t[5], fma2(vl, t[4], t[1]));
  r.ap = fma2(vh, t[6], t[2]);
  r.gl = fma2(nc, t[4], mul2(nm, t[3]));
  r.gh = fma2(np, t[6], mul2(nc, t[5]));
  return r;
}
// ... and one component's share: y += am um + ac uc + ap up + wl gl + wh gh
__device__ __forceinline__ void mom_row_comp(const MomRow &r, double2 um, double2 uc, double2 up, double2 wl, double2 wh, double2 &y)
{
  y = fma2(wh, r.gh, fma2(wl, r.gl, fma2(r.ap, up, fma2(r.ac, uc, fma2(r.am, um, y)))));
}

// Sum of the absolute values of one axis' row of component c OUTSIDE the diagonal (the Gershgorin radius of the row, OUT == 3): the
// entries on the component's own columns, am and ap, and -- through G = N . u_D -- wl t3, wl t4 + wh t5, wh t6 on the columns m, c, p of the
// face-normal component D.  own = (c == D): those three fall on the component's own columns, m and p merge with am and ap, the centre one
// is part of the diagonal (the term the DG path adds to da).  t3..t6: the face-interpolation numbers of mom_row_coef.
__device__ __forceinline__ double2 abs2(double2 a) { return make_double2(fabs(a.x), fabs(a.y)); }
__device__ __forceinline__ double2 add2(double2 a, double2 b) { return make_double2(a.x + b.x, a.y + b.y); }
__device__ __forceinline__ double2 mom_row_abs(const MomRow &r, double2 t3, double2 t4, double2 t5, double2 t6, double2 wl, double2 wh, bool own)
{
  if (own) return add2(abs2(fma2(wl, t3, r.am)), abs2(fma2(wh, t6, r.ap)));
  return add2(add2(abs2(r.am), abs2(r.ap)), add2(add2(abs2(mul2(wl, t3)), abs2(fma2(wh, t5, mul2(wl, t4)))), abs2(mul2(wh, t6))));
}
__device__ __forceinline__ double2 mom_row_abs(const MomRow &r, double t3, double t4, double t5, double t6, double2 wl, double2 wh, bool own)
{
  return mom_row_abs(r, make_double2(t3, t3), make_double2(t4, t4), make_double2(t5, t5), make_double2(t6, t6), wl, wh, own);
}

// general rows (a cell next to an end of the axis): T(q) = scaled number q of build_axis_momentum, uf = the far column of the
// one-sided second-derivative rows.  AB: aacc[c] += the row's absolute off-diagonal sum (see mom_row_abs; the far column counts)
template <int D, bool DG, bool AB = false, class TF>
__device__ __forceinline__ void mom_row_wall(TF T, const double (&um)[3], const double (&uc)[3], const double (&up)[3], const double (&uf)[3], double vl, double vh, const double (&wl)[3],
                                             const double (&wh)[3], double (&yacc)[3], double (&dacc)[3], double *aacc = nullptr)
{
  const double Nl0 = T(11), Nl1 = T(12), Nl2 = T(13), Nh0 = T(17), Nh1 = T(18), Nh2 = T(19);
  const double amt = fma(vh, T(14), fma(vl, T(8), T(0))), act = fma(vh, T(15), fma(vl, T(9), T(1))), apt = fma(vh, T(16), fma(vl, T(10), T(2))), aft = T(3);
  const double amn = fma(vh, Nh0, fma(vl, Nl0, T(4))), acn = fma(vh, Nh1, fma(vl, Nl1, T(5))), apn = fma(vh, Nh2, fma(vl, Nl2, T(6))), afn = T(7);
  const double Glo = fma(Nl2, up[D], fma(Nl1, uc[D], Nl0 * um[D])), Ghi = fma(Nh2, up[D], fma(Nh1, uc[D], Nh0 * um[D]));
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    const double am = c == D ? amn : amt, ac = c == D ? acn : act, ap = c == D ? apn : apt, af = c == D ? afn : aft;
    yacc[c] = fma(wh[c], Ghi, fma(wl[c], Glo, fma(af, uf[c], fma(ap, up[c], fma(ac, uc[c], fma(am, um[c], yacc[c]))))));
    if (DG) dacc[c] += ac;
    if (AB) {
      const double gm = fma(wh[c], Nh0, wl[c] * Nl0), gc = fma(wh[c], Nh1, wl[c] * Nl1), gp = fma(wh[c], Nh2, wl[c] * Nl2);
      aacc[c] += c == D ? fabs(am + gm) + fabs(ap + gp) + fabs(af) : fabs(am) + fabs(ap) + fabs(af) + fabs(gm) + fabs(gc) + fabs(gp);
    }
  }
  if (DG) dacc[D] += fma(wh[D], Nh1, wl[D] * Nl1);
}

// raw tables (slot-major, build_axis_momentum) -> scaled, cell-major tables of one axis
__global__ void __launch_bounds__(256) k_mom_scale_tab(const double *__restrict__ raw, int len, double cC, double cL, double *__restrict__ out)
{
  const int q = blockIdx.x * 256 + threadIdx.x, i = q / MOM_STAB, a = q % MOM_STAB;
  if (i >= len) return;
  const int fast[7] = {0, 1, 2, 8, 9, 15, 16};
  double    v = 0.;
  if (a < 7) v = raw[(int64_t)fast[a] * len + i] * (fast[a] < 8 ? cL : cC);
  else if (a >= 8 && a < 8 + MOM_NTAB) v = raw[(int64_t)(a - 8) * len + i] * (a - 8 < 8 ? cL : cC);
  out[q] = v;
}

// flags: bit 0 XCD-contiguous block order, bit 1 the unpadded output (OUT == 1) may be stored in 16-byte pairs
// DOT: bit 0 partial slots 0 sum y, 1 y.o (o padded, may be NULL); bit 1 slots 2 x.y, 3 y.y; slots not asked for are written as 0
// m.stab: the scaled, cell-major tables (k_mom_scale_tab) of this launch's (cC, cL)
template <int NW, int DOT, bool JAC, int OUT, int NT>
__global__ void __launch_bounds__(64 * NW, 2) k_mom2(GridP g, MomP m, const double *__restrict__ x, double *__restrict__ y, const double *__restrict__ F, int64_t cs, const double *__restrict__ o,
                                                     const KspScal *__restrict__ s, double *__restrict__ partial, int pstride, int tiles_x, int nchunk, int zc, int flags)
{
  using LT              = Mom2Lds<NW>;
  constexpr int TX = LT::TX, TY = LT::TY, NTH = 64 * NW;
  constexpr bool AB = OUT == 3;  // OUT == 3: y = (sum of |a_ij|, j != i) / |a_ii|, the Gershgorin radius of the Jacobi-scaled row
  constexpr bool DG = JAC || OUT == 2 || AB;
  __shared__ __attribute__((aligned(16))) LT lds;
  if (s && s->reason != 0) return;

  int b = blockIdx.x, chunk, tile;
  {
    const int nb = gridDim.x, tiles = nb / nchunk;
    if (flags & 1) b = xcd_remap(b, nb);
    chunk = b / tiles;  // chunk-major: consecutive logical blocks are neighbouring tiles of one z chunk
    tile  = b % tiles;
  }
  const int  tid = threadIdx.x, lane = tid & 63, w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int  i0 = (tile % tiles_x) * TX, j0 = (tile / tiles_x) * TY, j = j0 + w, i = i0 + 2 * lane;
  const bool own0 = i < g.nx, own1 = i + 1 < g.nx, rown = j < g.ny;
  // loads are unconditional on clamped, always-valid addresses (the ghost column / row holds the right neighbour of the last cell)
  const int      il = min(i, g.nx & ~1), jl = min(j, g.ny), jt = min(j, g.ny - 1);
  const int      k0 = chunk * zc, k1 = min(k0 + zc, g.nz);
  const unsigned lo0 = (unsigned)il * 8u;
  const bool     xwall = i0 == 0 || i0 + TX >= g.nx;  // block-uniform: the tile touches an end of the x axis
  const bool     ywall = jt == 0 || jt == g.ny - 1;   // wave-uniform
  const int64_t  sx = g.sx, sxy = g.sxy;
  const double   cI = m.cI;
  const int64_t  ncell = (int64_t)g.nx * g.ny * g.nz;
  const int64_t  rb0 = g.off0 + (int64_t)jl * sx;  // wave-uniform offset of this wave's row in plane 0 (the lane adds il)

  // ---- x-axis table numbers of the tile's columns: the 7 fast-path numbers, or the 20 general ones in a tile at an end of the axis
  {
    const int nq = xwall ? MOM_NTAB : 7, qoff = xwall ? 8 : 0;
    for (int q = tid >> 7; q < nq; q += NTH / 128) lds.tabx[q][tid & 127] = m.stab[0][(int64_t)min(i0 + (tid & 127), g.nx - 1) * MOM_STAB + qoff + q];
  }
  // ---- y-axis: the numbers of this wave's row (fixed for the whole chunk), scalar loads
  const cdouble4 *taby = as_const(m.stab[1]) + (int64_t)jt * MOM_STAB;
  double          tyi[7];
#pragma unroll
  for (int a = 0; a < 7; ++a) tyi[a] = taby[a];

  // ---- far column / row of the one-sided wall rows (general rows only; their coefficient is zero away from a wall)
  // x: column ia + 2 for ia == 0, ia - 2 otherwise, read from the LDS row (index clamped into the staged range: a clamped value only
  //    ever meets a zero coefficient) -- except when the tile holds a single column, then column nx - 3 lies outside it
  int fxi[2];
#pragma unroll
  for (int a = 0; a < 2; ++a) {
    const int ia = min(i + a, g.nx - 1), fi = ia == 0 ? ia + 2 : ia - 2;
    fxi[a]       = min(max(fi - i0 + 2, 1), TX + 2);
  }
  const bool xfar_g = xwall && g.nx - 1 == i0;
  const int  fj = jt == 0 ? jt + 2 : jt - 2, fyr = min(max(fj - j0 + 1, 0), TY + 1);
  const bool yfar_g = ywall && jt == g.ny - 1 && g.ny - 1 == j0;

  // ---- the ring items of this thread: 8-byte loads, staged in LDS
  constexpr int NI_UR = 3 * 2 * 128, NI_UC = 6 * TY, NI_FX = 4 * TY, NI_FY = 4 * 128, NI = NI_UR + NI_UC + NI_FX + NI_FY;
  constexpr int NS = (NI + NTH - 1) / NTH;
  // face fields: f = 0: V0, 1..3: v0interp_{0,1,2}; along axis d they are the arrays d, 3 + d, 6 + d, 9 + d of F
  const double *rp[NS];           // address in plane 0
  unsigned      rl[NS], rlb[NS];  // LDS byte offset in buffer 0, and what buffer 1 adds
  {
    char *const lb = reinterpret_cast<char *>(&lds);
#pragma unroll
    for (int sidx = 0; sidx < NS; ++sidx) {
      const int     q = tid + sidx * NTH;
      const double *base;
      int           row, col;
      const char   *dst;
      unsigned      bufb;
      if (q < NI_UR) {
        const int c = q >> 8, side = (q >> 7) & 1, ii = q & 127;
        base = x + (int64_t)c * cs;
        row  = side ? min(j0 + TY, g.ny) : j0 - 1;
        col  = min(i0 + ii, g.nx);
        dst  = reinterpret_cast<const char *>(&lds.u[0][c][side ? TY + 1 : 0][ii + 2]);
        bufb = sizeof(lds.u[0]);
      } else if (q < NI_UR + NI_UC) {
        const int qq = q - NI_UR, c = qq / (2 * TY), side = (qq / TY) & 1, jj = qq % TY;
        base = x + (int64_t)c * cs;
        row  = min(j0 + jj, g.ny);
        col  = side ? min(i0 + TX, g.nx) : i0 - 1;
        dst  = reinterpret_cast<const char *>(&lds.u[0][c][jj + 1][side ? TX + 2 : 1]);
        bufb = sizeof(lds.u[0]);
      } else if (q < NI_UR + NI_UC + NI_FX) {
        const int qq = q - NI_UR - NI_UC, f = qq / TY, jj = qq % TY;
        base = F + (int64_t)(3 * f) * cs;
        row  = min(j0 + jj, g.ny);
        col  = min(i0 + TX, g.nx);
        dst  = reinterpret_cast<const char *>(&lds.fxe[0][f][jj]);
        bufb = sizeof(lds.fxe[0]);
      } else if (q < NI) {
        const int qq = q - NI_UR - NI_UC - NI_FX, f = qq >> 7, ii = qq & 127;
        base = F + (int64_t)(3 * f + 1) * cs;
        row  = min(j0 + TY, g.ny);
        col  = min(i0 + ii, g.nx);
        dst  = reinterpret_cast<const char *>(&lds.fy[0][f][TY][ii]);
        bufb = sizeof(lds.fy[0]);
      } else {  // no item: a valid address and the dead corner slot (row -1, column -2) nobody reads
        base = x;
        row  = 0;
        col  = 0;
        dst  = reinterpret_cast<const char *>(&lds.u[0][0][0][0]);
        bufb = sizeof(lds.u[0]);
      }
      rp[sidx]  = base + g.off0 + (int64_t)row * sx + col;
      rl[sidx]  = (unsigned)(dst - lb);
      rlb[sidx] = bufb;
    }
  }

  // ---- registers of the z march
  double2 uzm[3], ucc[3], uzp[3], fzl[4], fzh[4], fxl[4], fyl[4], oc[3];
  double  rv[NS];
  double  acc[4] = {0., 0., 0., 0.};
  {
    const int64_t rb = rb0 + (int64_t)k0 * sxy;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      const double *X = x + (int64_t)c * cs;
      uzm[c] = LD2(X + rb - sxy, lo0);
      ucc[c] = LD2(X + rb, lo0);
      uzp[c] = LD2(X + rb + sxy, lo0);
      oc[c]  = ((DOT & 1) && o) ? LD2(o + (int64_t)c * cs + rb, lo0) : make_double2(0., 0.);
    }
#pragma unroll
    for (int f = 0; f < 4; ++f) {
      const double *Ff = F + (int64_t)(3 * f) * cs + rb;
      fyl[f] = LD2(Ff + cs, lo0);
      fxl[f] = LD2(Ff, lo0);
      fzl[f] = LD2(Ff + 2 * cs, lo0);
      fzh[f] = LD2(Ff + 2 * cs + sxy, lo0);
    }
#pragma unroll
    for (int sidx = 0; sidx < NS; ++sidx) rv[sidx] = rp[sidx][(int64_t)k0 * sxy];
  }
  __syncthreads();  // tabx

  for (int kl = k0; kl < k1; ++kl) {
    // Opaque copies of the plane index, the lane offset and the array stride: without them the loop optimiser turns every stream into
    // its own 64-bit pointer carried around the loop (dozens of registers, scalar ones spilled, no scalar-base addressing).
    int      k = kl;
    unsigned lo = lo0;
    int64_t  csl = cs;
    asm volatile("" : "+s"(k), "+v"(lo), "+s"(csl));
    const int     bf = k & 1;
    const int64_t rb = rb0 + (int64_t)k * sxy;
    const int64_t rb1 = rb + sxy, rb2 = rb0 + (int64_t)min(k + 2, g.nz) * sxy;  // plane k + 2 is clamped to the high ghost plane
    const bool    zwall = k == 0 || k == g.nz - 1;
    const int64_t cs3 = 3 * csl;

    // ---- A: publish this wave's row of plane k and the ring items fetched for it
#pragma unroll
    for (int c = 0; c < 3; ++c) *reinterpret_cast<double2 *>(&lds.u[bf][c][w + 1][2 * lane + 2]) = ucc[c];
#pragma unroll
    for (int f = 0; f < 4; ++f) *reinterpret_cast<double2 *>(&lds.fy[bf][f][w][2 * lane]) = fyl[f];
    {
      char *const lb = reinterpret_cast<char *>(&lds);
#pragma unroll
      for (int sidx = 0; sidx < NS; ++sidx) *reinterpret_cast<double *>(lb + rl[sidx] + (bf ? rlb[sidx] : 0u)) = rv[sidx];
    }
    // ---- B: the ring of plane k + 1
    {
      const int64_t pn = (int64_t)min(k + 1, g.nz) * sxy;
#pragma unroll
      for (int sidx = 0; sidx < NS; ++sidx) rv[sidx] = rp[sidx][pn];
    }
    __syncthreads();

    double2 ya[3] = {{0., 0.}, {0., 0.}, {0., 0.}}, da[3] = {{0., 0.}, {0., 0.}, {0., 0.}}, aa[3] = {{0., 0.}, {0., 0.}, {0., 0.}};
    // general rows of one cell (a = 0 / 1) of the pair: scalar arithmetic on copies of the pair accumulators
#define MOM_WALL_CELL(D_, T_, UM, UC, UP, UF, VL, VH, WL, WH)                                        \
  {                                                                                                  \
    double y_[3] = {a ? ya[0].y : ya[0].x, a ? ya[1].y : ya[1].x, a ? ya[2].y : ya[2].x};            \
    double d_[3] = {a ? da[0].y : da[0].x, a ? da[1].y : da[1].x, a ? da[2].y : da[2].x};            \
    double a_[3] = {a ? aa[0].y : aa[0].x, a ? aa[1].y : aa[1].x, a ? aa[2].y : aa[2].x};            \
    mom_row_wall<D_, DG, AB>(T_, UM, UC, UP, UF, VL, VH, WL, WH, y_, d_, a_);                        \
    _Pragma("unroll") for (int c = 0; c < 3; ++c)                                                    \
    {                                                                                                \
      if (a) {                                                                                       \
        ya[c].y = y_[c];                                                                             \
        da[c].y = d_[c];                                                                             \
        if (AB) aa[c].y = a_[c];                                                                     \
      } else {                                                                                       \
        ya[c].x = y_[c];                                                                             \
        da[c].x = d_[c];                                                                             \
        if (AB) aa[c].x = a_[c];                                                                     \
      }                                                                                              \
    }                                                                                                \
  }
    // ---- D: the y axis
    if (ywall) {
      double2 us[3], un[3], fyh[4];
#pragma unroll
      for (int c = 0; c < 3; ++c) {
        us[c] = *reinterpret_cast<const double2 *>(&lds.u[bf][c][w][2 * lane + 2]);
        un[c] = *reinterpret_cast<const double2 *>(&lds.u[bf][c][w + 2][2 * lane + 2]);
      }
#pragma unroll
      for (int f = 0; f < 4; ++f) fyh[f] = *reinterpret_cast<const double2 *>(&lds.fy[bf][f][w + 1][2 * lane]);
#pragma unroll
      for (int a = 0; a < 2; ++a) {
        double um[3], uc[3], up[3], uf[3], wl[3], wh[3];
#pragma unroll
        for (int c = 0; c < 3; ++c) {
          um[c] = a ? us[c].y : us[c].x;
          uc[c] = a ? ucc[c].y : ucc[c].x;
          up[c] = a ? un[c].y : un[c].x;
          wl[c] = a ? fyl[c + 1].y : fyl[c + 1].x;
          wh[c] = a ? fyh[c + 1].y : fyh[c + 1].x;
          uf[c] = lds.u[bf][c][fyr][2 * lane + 2 + a];
          // the tile holds a single row: row ny - 3 lies outside it (rare; a synchronous load)
          if (yfar_g) uf[c] = (x + (int64_t)c * csl + g.off0 + (int64_t)k * sxy + (int64_t)max(g.ny - 3, -1) * sx)[il + a];
        }
        const double vl = a ? fyl[0].y : fyl[0].x, vh = a ? fyh[0].y : fyh[0].x;
        auto         T = [&](int q) { return (double)taby[8 + q]; };
        MOM_WALL_CELL(1, T, um, uc, up, uf, vl, vh, wl, wh)
      }
    } else {
      const double2 us1 = *reinterpret_cast<const double2 *>(&lds.u[bf][1][w][2 * lane + 2]);
      const double2 un1 = *reinterpret_cast<const double2 *>(&lds.u[bf][1][w + 2][2 * lane + 2]);
      const double2 vh  = *reinterpret_cast<const double2 *>(&lds.fy[bf][0][w + 1][2 * lane]);
      const MomRow  r   = mom_row_coef(tyi, fyl[0], vh, us1, ucc[1], un1);
#pragma unroll
      for (int c = 0; c < 3; ++c) {
        const double2 us = c == 1 ? us1 : *reinterpret_cast<const double2 *>(&lds.u[bf][c][w][2 * lane + 2]);
        const double2 un = c == 1 ? un1 : *reinterpret_cast<const double2 *>(&lds.u[bf][c][w + 2][2 * lane + 2]);
        const double2 wh = *reinterpret_cast<const double2 *>(&lds.fy[bf][c + 1][w + 1][2 * lane]);
        mom_row_comp(r, us, ucc[c], un, fyl[c + 1], wh, ya[c]);
        if (AB) aa[c] = add2(aa[c], mom_row_abs(r, tyi[3], tyi[4], tyi[5], tyi[6], fyl[c + 1], wh, c == 1));
        if (DG) {
          da[c].x += r.ac.x;
          da[c].y += r.ac.y;
          if (c == 1) da[1] = fma2(wh, tyi[5], fma2(fyl[2], tyi[4], da[1]));
        }
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    __builtin_amdgcn_sched_barrier(0);
    // ---- E: y-faces of plane k + 1
    {
      const double *Ff = F + csl + rb1;
#pragma unroll
      for (int f = 0; f < 4; ++f) fyl[f] = LD2(Ff + f * cs3, lo);
    }
    __builtin_amdgcn_sched_barrier(0);
    // ---- F: the x axis
    if (xwall) {
      double uw[3], ue[3], fxn[4];
#pragma unroll
      for (int c = 0; c < 3; ++c) {
        uw[c] = lds.u[bf][c][w + 1][2 * lane + 1];
        ue[c] = lds.u[bf][c][w + 1][2 * lane + 4];
      }
#pragma unroll
      for (int f = 0; f < 4; ++f) fxn[f] = from_next_lane(fxl[f].x, lds.fxe[bf][f][w]);
#pragma unroll
      for (int a = 0; a < 2; ++a) {
        double um[3], uc[3], up[3], uf[3], wl[3], wh[3];
#pragma unroll
        for (int c = 0; c < 3; ++c) {
          um[c] = a ? ucc[c].x : uw[c];
          uc[c] = a ? ucc[c].y : ucc[c].x;
          up[c] = a ? ue[c] : ucc[c].y;
          wl[c] = a ? fxl[c + 1].y : fxl[c + 1].x;
          wh[c] = a ? fxn[c + 1] : fxl[c + 1].y;
          uf[c] = lds.u[bf][c][w + 1][fxi[a]];
          if (xfar_g) uf[c] = (x + (int64_t)c * csl + rb)[max(g.nx - 3, -1)];  // single-column tile (rare; a synchronous load)
        }
        const double vl = a ? fxl[0].y : fxl[0].x, vh = a ? fxn[0] : fxl[0].y;
        auto         T = [&](int q) { return lds.tabx[q][2 * lane + a]; };
        MOM_WALL_CELL(0, T, um, uc, up, uf, vl, vh, wl, wh)
      }
    } else {
      double2 tx[7];
#pragma unroll
      for (int q = 0; q < 7; ++q) tx[q] = *reinterpret_cast<const double2 *>(&lds.tabx[q][2 * lane]);
      MomRow r;
      {
        const double  uw = lds.u[bf][0][w + 1][2 * lane + 1], ue = lds.u[bf][0][w + 1][2 * lane + 4];
        const double2 vh = make_double2(fxl[0].y, from_next_lane(fxl[0].x, lds.fxe[bf][0][w]));
        r = mom_row_coef(tx, fxl[0], vh, make_double2(uw, ucc[0].x), ucc[0], make_double2(ucc[0].y, ue));
      }
#pragma unroll
      for (int c = 0; c < 3; ++c) {
        const double  uw = lds.u[bf][c][w + 1][2 * lane + 1], ue = lds.u[bf][c][w + 1][2 * lane + 4];
        const double2 wh = make_double2(fxl[c + 1].y, from_next_lane(fxl[c + 1].x, lds.fxe[bf][c + 1][w]));
        mom_row_comp(r, make_double2(uw, ucc[c].x), ucc[c], make_double2(ucc[c].y, ue), fxl[c + 1], wh, ya[c]);
        if (AB) aa[c] = add2(aa[c], mom_row_abs(r, tx[3], tx[4], tx[5], tx[6], fxl[c + 1], wh, c == 0));
        if (DG) {
          da[c].x += r.ac.x;
          da[c].y += r.ac.y;
          if (c == 0) da[0] = fma2(wh, tx[5], fma2(fxl[1], tx[4], da[0]));
        }
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    __builtin_amdgcn_sched_barrier(0);
    // ---- G: x-faces of plane k + 1
    {
      const double *Ff = F + rb1;
#pragma unroll
      for (int f = 0; f < 4; ++f) fxl[f] = LD2(Ff + f * cs3, lo);
    }
    __builtin_amdgcn_sched_barrier(0);
    // ---- H: the z axis
    {
      const cdouble4 *tabz = as_const(m.stab[2]) + (int64_t)k * MOM_STAB;
      if (zwall) {  // first / last plane of the axis: the far plane comes straight from memory (a synchronous load, twice per column)
#pragma unroll
        for (int a = 0; a < 2; ++a) {
          double um[3], uc[3], up[3], uf[3], wl[3], wh[3];
#pragma unroll
          for (int c = 0; c < 3; ++c) {
            um[c] = a ? uzm[c].y : uzm[c].x;
            uc[c] = a ? ucc[c].y : ucc[c].x;
            up[c] = a ? uzp[c].y : uzp[c].x;
            wl[c] = a ? fzl[c + 1].y : fzl[c + 1].x;
            wh[c] = a ? fzh[c + 1].y : fzh[c + 1].x;
            uf[c] = (x + (int64_t)c * csl + rb + (k == 0 ? 2 : -2) * sxy)[il + a];
          }
          const double vl = a ? fzl[0].y : fzl[0].x, vh = a ? fzh[0].y : fzh[0].x;
          auto         T = [&](int q) { return (double)tabz[8 + q]; };
          MOM_WALL_CELL(2, T, um, uc, up, uf, vl, vh, wl, wh)
        }
      } else {
        double tz[7];
#pragma unroll
        for (int q = 0; q < 7; ++q) tz[q] = tabz[q];
        const MomRow r = mom_row_coef(tz, fzl[0], fzh[0], uzm[2], ucc[2], uzp[2]);
#pragma unroll
        for (int c = 0; c < 3; ++c) {
          mom_row_comp(r, uzm[c], ucc[c], uzp[c], fzl[c + 1], fzh[c + 1], ya[c]);
          if (AB) aa[c] = add2(aa[c], mom_row_abs(r, tz[3], tz[4], tz[5], tz[6], fzl[c + 1], fzh[c + 1], c == 2));
          if (DG) {
            da[c].x += r.ac.x;
            da[c].y += r.ac.y;
            if (c == 2) da[2] = fma2(fzh[3], tz[5], fma2(fzl[3], tz[4], da[2]));
          }
        }
      }
    }
#undef MOM_WALL_CELL
    __builtin_amdgcn_sched_barrier(0);
    // ---- I: the result of plane k
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      double2 yv = fma2(ucc[c], cI, ya[c]);
      if (DG) {
        const double d0 = cI + da[c].x, d1 = cI + da[c].y;
        if (OUT == 2) {
          yv.x = d0;
          yv.y = d1;
        } else if (AB) {
          yv.x = aa[c].x / fabs(d0);
          yv.y = aa[c].y / fabs(d1);
        } else {  // PCJacobi: VecReciprocal(diag) once, VecPointwiseMult per apply
          yv.x = yv.x * recip(d0);
          yv.y = yv.y * recip(d1);
        }
      }
      if (rown) {
        if (OUT == 1) {
          double *yo = y + (int64_t)c * ncell + ((int64_t)k * g.ny + j) * g.nx + i;
          if (own1 && (flags & 2)) st2<NT>(yo, yv);
          else {
            if (own0) yo[0] = yv.x;
            if (own1) yo[1] = yv.y;
          }
        } else {
          if (own1) ST2nt<NT>(y + (int64_t)c * csl + rb, lo, yv);
          else if (own0) (y + (int64_t)c * csl + rb)[il] = yv.x;
        }
      }
      if (DOT) {
        const bool   o0 = rown && own0, o1 = rown && own1;
        const double y0 = o0 ? yv.x : 0., y1 = o1 ? yv.y : 0.;
        if (DOT & 1) {
          acc[0] += y0 + y1;
          acc[1] += y0 * oc[c].x + y1 * oc[c].y;
        }
        if (DOT & 2) {
          acc[2] += ucc[c].x * y0 + ucc[c].y * y1;
          acc[3] += y0 * y0 + y1 * y1;
        }
      }
    }
    __builtin_amdgcn_sched_barrier(0);
    // ---- J: rotate the z march, fetch plane k + 2 into the registers that became free
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      uzm[c] = ucc[c];
      ucc[c] = uzp[c];
      uzp[c] = LD2(x + (int64_t)c * csl + rb2, lo);
      if ((DOT & 1) && o) oc[c] = LD2(o + (int64_t)c * csl + rb1, lo);
    }
    {
      const double *Ff = F + 2 * csl + rb2;
#pragma unroll
      for (int f = 0; f < 4; ++f) {
        fzl[f] = fzh[f];
        fzh[f] = LD2(Ff + f * cs3, lo);
      }
    }
  }
  if (DOT) {
#pragma unroll
    for (int a = 0; a < 4; ++a) {
      const double v = wave_sum(acc[a]);
      if (lane == 0) lds.red[a * NW + w] = v;
    }
    __syncthreads();
    if (tid < 4) {
      double v = 0.;
#pragma unroll
      for (int q = 0; q < NW; ++q) v += lds.red[tid * NW + q];
      partial[(int64_t)tid * pstride + blockIdx.x] = v;
    }
  }
}

}  // namespace fl
