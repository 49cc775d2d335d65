// fl_cheb2.hip -- two Chebyshev(-Jacobi) steps of KSPCHEBYSHEV in ONE sweep over memory (temporal blocking).
//
// One step (k_cheb, fl_ksp.hip; the recurrence restated in oracle/fluca_oracle.c, KSPCHEBYSHEV + PCJACOBI on the Schur
// complement of fluca/src/ns/utils/abfpc/abfpc.c:77) is
//     z = M (b - S x) ;  d' = rho d + c z ;  x' = x + d'            reads x, b, d   writes x', d'   = 40 B/cell.
// When no convergence test sits between two steps (KSP_NORM_NONE: the fixed-length sweeps of BASELINE config 3 and the
// multigrid smoother) step j+1 only needs x', d' on the cell and x' on its six neighbours, so a tile that holds x, b, d
// with a ring of cells around it can apply BOTH steps before anything goes back to memory: 40 B/cell per TWO steps.
//
// Plan (the LDS-staged plane march of k_cg_A, fl_kernels.hip): a block owns a 128 x (NW*RY) tile and marches through a z
// chunk.  Trip kk loads plane kk of x, b, d (tile + ring, one trip ahead into a second register set), forms
//     step 1 on plane kk-1, on the tile AND its one-cell ring (x of plane kk-1 and kk-2 from LDS, plane kk from registers),
//     step 2 on plane kk-2, on the tile (x' of plane kk-2, kk-3 from LDS, plane kk-1 from registers),
// stages x(kk) and x'(kk-1) in LDS (three planes each, one barrier per trip) and stores x'', d'' of plane kk-2.
// Ring cells are fetched from where they live: across a periodic seam inside the block the index wraps, outside a wall the cell does not
// exist (its stencil coefficient is 0; x' there is set to 0) -- on one rank no ghost layer is read or written and no ghost fill is needed in
// front of this kernel.  Where a neighbouring RANK owns the ring (round 4) the values come from the ghost layers of the wide layout: two
// layers of x with the edge cells of that shell (fl_fill_ghosts_deep), one layer of b and d (fl_fill_ghosts), filled by the caller.
// d is double-buffered like x (a neighbouring block still reads the old d of this tile's cells as its ring).
#include <type_traits>

#include "fl_device.h"
#include "fl_handle.h"

namespace fl {

typedef double v2d_t __attribute__((ext_vector_type(2)));
template <int NT>
__device__ __forceinline__ double2 c2_ld2(const double *p)
{
  if (NT) {
    const v2d_t v = __builtin_nontemporal_load(reinterpret_cast<const v2d_t *>(p));
    return make_double2(v.x, v.y);
  }
  return *reinterpret_cast<const double2 *>(p);
}
template <int NT>
__device__ __forceinline__ void c2_st2(double *p, double2 v)
{
  if (NT) {
    v2d_t t;
    t.x = v.x;
    t.y = v.y;
    __builtin_nontemporal_store(t, reinterpret_cast<v2d_t *>(p));
  } else *reinterpret_cast<double2 *>(p) = v;
}

// uniform base (scalar registers) + 32-bit per-lane byte offset: global_load/store ... v_off, s[base:base+1]
template <int NT>
__device__ __forceinline__ double2 c2_LD2(const double *base, unsigned byteoff) { return c2_ld2<NT>(reinterpret_cast<const double *>(reinterpret_cast<const char *>(base) + byteoff)); }
__device__ __forceinline__ double c2_LD1(const double *base, unsigned byteoff) { return *reinterpret_cast<const double *>(reinterpret_cast<const char *>(base) + byteoff); }
template <int NT>
__device__ __forceinline__ void c2_ST2(double *base, unsigned byteoff, double2 v) { c2_st2<NT>(reinterpret_cast<double *>(reinterpret_cast<char *>(base) + byteoff), v); }
__device__ __forceinline__ void c2_ST1(double *base, unsigned byteoff, double v) { *reinterpret_cast<double *>(reinterpret_cast<char *>(base) + byteoff) = v; }

// logical index gi in [-2, n+1] -> index of the cell that holds the value; in = false: no such cell (outside a wall).
// per: what lies behind the low end (bits 0-1) and the high end (bits 2-3) of the axis -- 0 a wall, 1 the periodic image inside this block
// (the index wraps), 2 a neighbouring rank: the value sits in the ghost layers of the padded array (two of them: fl_fill_ghosts_deep), the
// index stays as it is.
__device__ __forceinline__ int c2_wrap(int gi, int n, int per, bool &in)
{
  if (gi < 0) {
    const int m = per & 3;
    in = m != 0;
    return m == 1 ? gi + n : (m == 2 ? gi : 0);
  }
  if (gi >= n) {
    const int m = (per >> 2) & 3;
    in = m != 0;
    return m == 1 ? gi - n : (m == 2 ? gi : n - 1);
  }
  in = true;
  return gi;
}

__device__ __forceinline__ int c2_xcd_remap(int b, int nblocks) { return (nblocks & 7) ? b : (b & 7) * (nblocks >> 3) + (b >> 3); }

// sums: 0 sum z  1 z.z  2 r.r of step 1;  3..5 the same of step 2
// MGD (the last smoothing sweep of a multigrid cycle inside PCG): instead, the five sums the outer iteration wants of the sweep's result
// x'' and its right-hand side b -- 0 sum x''  1 x''.x''  2 b.x''  3 sum b  4 b.b (slot 5 is 0) -- which saves a pass over both vectors
// Z (round 5; the pre-smoother of a multigrid cycle, one rank): THREE steps from a zero initial guess in the one sweep.  The first step needs no
// stencil -- x_1 = d_1 = c_0 M b -- so the sweep forms it wherever the regular kernel would LOAD x (tile + two rings, plane kk) from b loaded
// there instead, takes d_1 = x_1 from the staged plane, and runs its two stencil steps as steps 2 and 3: reads b (tile + two rings), writes
// x_3, d_3 -- 26 B/cell where k_cheb_first + the regular sweep move 16 + 24 + 44.  Z == 2: the right-hand side is first updated, b' = b - *za * zq
// (the outer CG's r -= alpha q, alpha in device memory), read from b and zq with the rings and WRITTEN TO ANOTHER ARRAY bw on the tile -- a
// neighbouring block still reads the old b of this tile's cells as its ring, so the update cannot be made in place: 42 B/cell for what took 84.
template <int RY, int NW, bool JAC, int NT, bool MGD = false, int Z = 0>
__global__ void __launch_bounds__(64 * NW) k_cheb2(GridP g, int perx, int pery, int perz, const double *X0, const double *X1, double *X0w, double *X1w, const double *__restrict__ b, const double *D0, const double *D1, double *D0w,
                                                  double *D1w, const KspScal *__restrict__ s, double *__restrict__ partial, int stride, int zc, int tiles_x, int tiles, int remap, const double *__restrict__ zq = nullptr,
                                                  const double *__restrict__ za_dev = nullptr, double *__restrict__ bw = nullptr)
{
  constexpr bool ZR = Z != 0, ZS = Z == 2;
  static_assert(!(ZR && MGD), "the three-step sweep is the pre-smoother: no sums of the outer iteration");
  constexpr int TX = 128, TY = NW * RY, LX = TX + 4, NTH = 64 * NW;
  constexpr int NTL = NT >= 2, NTS = NT >= 1;
  // XJ: x of planes kk, kk-1, kk-2 on the tile + two rings; row jj+2, column ii+2 for local (jj, ii)
  // XN: x' of planes kk-1, kk-2, kk-3 on the tile + one ring; row jj+1, column ii+2
  __shared__ __attribute__((aligned(16))) double XJ[3][TY + 4][LX];
  __shared__ __attribute__((aligned(16))) double XN[3][TY + 2][LX];
  __shared__ double                              red[6 * NW], cB[2 * TY][5];
  if (s->reason != 0) return;

  const double *x  = s->cur ? X1 : X0;
  double       *xn = s->cur ? X0w : X1w;
  const double *d  = s->dcur ? D1 : D0;
  double       *dn = s->dcur ? D0w : D1w;
  double        rho0 = s->cheb_rho, c0 = s->cheb_c;
  double        rho1, c1;
  const double  cf = s->cheb_c;            // ZR: the factor of the first (stencil-free) step
  const double  za = ZS ? *za_dev : 0.;
  {
    double ck, ckm1;
    cheb_advance(s, s->ck, s->ckm1, ck, ckm1, rho1, c1);
    if (ZR) {  // the block is the one of the FIRST step: steps 2 and 3 take the next two sets of the recurrence
      rho0 = rho1;
      c0   = c1;
      double ck2, ckm2;
      cheb_advance(s, ck, ckm1, ck2, ckm2, rho1, c1);
    }
  }

  const int bb    = remap ? c2_xcd_remap(blockIdx.x, gridDim.x) : blockIdx.x;
  const int chunk = bb / tiles, tile = bb % tiles;
  const int i0 = (tile % tiles_x) * TX, j0 = (tile / tiles_x) * TY;
  const int k0 = chunk * zc, k1 = min(k0 + zc, g.nz);
  const int tid = threadIdx.x, lane = tid & 63;
  const int w   = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int ie = min(TX, g.nx - i0), je = min(TY, g.ny - j0);  // local column / row of the high ring-1 line

  // ---- this thread's tile cells: a pair of x-adjacent cells in each of RY rows ---------------------------------------
  const int    i   = i0 + 2 * lane;
  const bool   in0 = i < g.nx, in1 = i + 1 < g.nx;
  const int    il  = min(i, g.nx & ~1);
  const int    ic0 = min(i, g.nx), ic1 = min(i + 1, g.nx);
  const double xl0 = g.sl[0][ic0], xc0 = g.sc[0][ic0], xh0 = g.sh[0][ic0];
  const double xl1 = g.sl[0][ic1], xc1 = g.sc[0][ic1], xh1 = g.sh[0][ic1];
  int64_t      rob[RY];  // wave-uniform: cell (0, row, plane 0); the lane adds lo bytes
  bool         rin[RY];
  unsigned     lo = 8u * (unsigned)il;
  double       yl[RY], yc[RY], yh[RY];
#pragma unroll
  for (int m = 0; m < RY; ++m) {
    const int j = j0 + w * RY + m, jc = min(j, g.ny);
    rin[m]      = j < g.ny;
    rob[m]      = g.off0 + (int64_t)jc * g.sx;
    yl[m]       = g.sl[1][jc];
    yc[m]       = g.sc[1][jc];
    yh[m]       = g.sh[1][jc];
  }

  // ---- halo cells of this thread ---------------------------------------------------------------------------------------
  // rows (16-byte pairs, one row per wave, see below): jj = -1 / je (ring 1: step 1 is formed there), jj = -2 / je+1 (ring 2: x only)
  // B (step 1 is formed here): columns ii = -1 / ie, rows jj = 0..je-1                                    -- 8-byte items, one per thread
  // E (x only): columns ii = -2 / ie+1, rows 0..je-1, and the four corners (-1|je, -1|ie) of ring 1       -- 8-byte items
  constexpr int HB0 = NTH > 256 ? 256 : 0, HE0 = NTH > 256 ? 256 + 2 * TY : 0;
  struct Halo {
    unsigned off;  // byte offset of the cell inside a plane, relative to cell (-2,-2) (a ghost cell may have negative coordinates)
    bool ok, in;
    int  lr, lc;  // XJ position
    int  pi, pj;  // the cell that holds the value (ZR: its diagonal comes from the tables at these indices)
  };
  auto mk = [&](bool ok, int jj, int ii) {
    Halo      H;
    bool      inx, iny;
    const int pi = c2_wrap(i0 + ii, g.nx, perx, inx), pj = c2_wrap(j0 + jj, g.ny, pery, iny);
    H.ok  = ok;
    H.in  = ok && inx && iny;
    H.off = ok ? 8u * (unsigned)((pj + 2) * g.sx + (pi + 2)) : 8u * (unsigned)(2 * g.sx + 2);
    H.lr  = ok ? jj + 2 : 1;  // a thread without such a cell reads around (1, 1) and stages nothing
    H.lc  = ok ? ii + 2 : 1;
    H.pi  = ok ? pi : 0;
    H.pj  = ok ? pj : 0;
    return H;
  };
  const int  tb = tid - HB0, te = tid - HE0;
  // ring ROWS travel as 16-byte pairs, one row per wave (wave-uniform kind): wave 0 row -1 and wave 1 row je (ring 1: x, b, d, step 1 is formed
  // there), wave 2 row -2 and wave 3 row je + 1 (ring 2: x only).  A lane holds the same two columns as in its tile rows, so the x coefficients
  // and the lane offset `lo` are the tile's.  (Until round 4 these were 8-byte items, four loads per thread of waves 0..3.)
  const bool rowA = w < 2, rowC = w == 2 || w == 3;  // wave-uniform
  const int  rjj = w == 0 ? -1 : w == 1 ? je : w == 2 ? -2 : je + 1;
  bool       riny;
  const int  rpj  = c2_wrap(j0 + rjj, g.ny, pery, riny);
  const int64_t rrow = g.off0 + (int64_t)rpj * g.sx;  // cell (0, row, plane 0) of the wave's ring row
  const bool rok0 = w < 4 && 2 * lane < ie, rok1 = w < 4 && 2 * lane + 1 < ie;
  const bool rinn0 = rok0 && riny, rinn1 = rok1 && riny;
  const int  rlr = min(rjj + 2, TY + 3);  // XJ row of the ring row (waves without one never use it)
  const bool hBok = tb >= 0 && tb < 2 * TY && (tb >> 1) < je;
  const int  hBjj = tb >> 1, hBii = (tb & 1) ? ie : -1;
  const Halo HB = mk(hBok, hBjj, hBii);
  bool       hEok;
  int        hEjj, hEii;
  if (te >= 0 && te < 2 * TY) {
    hEok = (te >> 1) < je;
    hEjj = te >> 1;
    hEii = (te & 1) ? ie + 1 : -2;
  } else {
    const int c = te - 2 * TY;
    hEok        = c >= 0 && c < 4;
    hEjj        = (c & 2) ? je : -1;
    hEii        = (c & 1) ? ie : -1;
  }
  const Halo HE = mk(hEok, hEjj, hEii);
  // which kinds of column items this WAVE holds at all (wave-uniform): a wave without any skips their loads, their step-1 arithmetic and their
  // staging instead of executing them on a dummy cell -- with 512 threads columns live on waves 4 and 5, nothing on 6 and 7
  const bool anyB = __builtin_amdgcn_ballot_w64(hBok) != 0, anyE = __builtin_amdgcn_ballot_w64(hEok) != 0;
  // 1-D y coefficients of the ring-1 row of this wave (wave-uniform; table index = local cell index, -1..n); the five numbers of a B cell wait in LDS
  const int    cAj = __builtin_amdgcn_readfirstlane(min(max(j0 + rjj, -1), g.ny));
  const double Ayl = g.sl[1][cAj], Ayh = g.sh[1][cAj], Ayc = g.sc[1][cAj];
  if (hBok) {
    const int cBi = min(max(i0 + hBii, -1), g.nx), cBj = min(max(j0 + hBjj, -1), g.ny);
    cB[tb][0] = g.sl[0][cBi];
    cB[tb][1] = g.sh[0][cBi];
    cB[tb][2] = g.sl[1][cBj];
    cB[tb][3] = g.sh[1][cBj];
    cB[tb][4] = g.sc[0][cBi] + g.sc[1][cBj];
  }
  const int tbc = hBok ? tb : 0;
  // ZR: x and y parts of the diagonal at this thread's ring cells (walls: any valid cell, the value is never used)
  const double zrRy = ZR ? g.sc[1][rpj] : 0.;
  const double zrBx = ZR ? g.sc[0][HB.pi] : 0., zrBy = ZR ? g.sc[1][HB.pj] : 0.;
  const double zrEx = ZR ? g.sc[0][HE.pi] : 0., zrEy = ZR ? g.sc[1][HE.pj] : 0.;

  // inputs of one trip, fetched one trip ahead: x of plane kk (z-high neighbour of step 1), b and d of plane kk-1 (step 1 itself)
  struct Raw {
    double2 x[RY], b[RY], d[RY];
    double2 hxR, hbR, hdR;  // the wave's ring row: x (rows of ring 1 and 2), b and d (ring 1 only)
    double  hxB, hbB, hdB, hxE;
    double  hqE;  // ZS: q at the E item
    double  zl, zc, zh;
    double  zcw;  // ZR: the z part of the diagonal of the plane the values were read from
    int64_t pl;   // ZS: the plane's offset
    bool    pin;
  };
  auto load = [&](int kk_, Raw &R) {
    int kk = min(kk_, k1 + 1);  // the prefetch of the last trip re-reads a cached plane
    asm volatile("" : "+s"(kk), "+v"(lo));  // keep the loop optimiser from turning every stream into its own 64-bit VGPR pointer
    bool          pin, dum;
    const int     pk  = c2_wrap(kk, g.nz, perz, pin);
    const int     pkb = c2_wrap(min(max(kk - 1, k0 - 1), k1), g.nz, perz, dum);  // b, d of plane kk-1 (used on planes k0-1 .. k1 only)
    const int64_t pl = (int64_t)pk * g.sxy, plb = (int64_t)pkb * g.sxy;
    const int64_t hbase = g.off0 - 2 * (int64_t)g.sx - 2;  // cell (-2,-2) of plane 0: the origin of the items' offsets
    R.hxB = R.hbB = R.hdB = R.hxE = R.hqE = 0.;
    R.hxR = R.hbR = R.hdR = make_double2(0., 0.);
    if constexpr (ZR) {  // b (and q) of plane kk where the regular sweep reads x: the x and d slots carry them until convert() has run
#pragma unroll
      for (int m = 0; m < RY; ++m) {
        R.x[m] = c2_LD2<NTL>(b + rob[m] + pl, lo);
        if (ZS) R.d[m] = c2_LD2<NTL>(zq + rob[m] + pl, lo);
        R.b[m] = make_double2(0., 0.);
      }
      const double *hb = b + hbase + pl, *hq = zq + hbase + pl;
      if (rowA || rowC) {
        R.hxR = c2_LD2<0>(b + rrow + pl, lo);
        if (ZS) R.hdR = c2_LD2<0>(zq + rrow + pl, lo);
      }
      if (anyB) {
        R.hxB = c2_LD1(hb, HB.off);
        if (ZS) R.hdB = c2_LD1(hq, HB.off);
      }
      if (anyE) {
        R.hxE = c2_LD1(hb, HE.off);
        if (ZS) R.hqE = c2_LD1(hq, HE.off);
      }
    } else {
#pragma unroll
      for (int m = 0; m < RY; ++m) {
        R.x[m] = c2_LD2<NTL>(x + rob[m] + pl, lo);
        R.b[m] = c2_LD2<NTL>(b + rob[m] + plb, lo);
        R.d[m] = c2_LD2<NTL>(d + rob[m] + plb, lo);
      }
      const double *hx = x + hbase + pl, *hb = b + hbase + plb, *hd = d + hbase + plb;
      if (rowA || rowC) R.hxR = c2_LD2<0>(x + rrow + pl, lo);
      if (rowA) {
        R.hbR = c2_LD2<0>(b + rrow + plb, lo);
        R.hdR = c2_LD2<0>(d + rrow + plb, lo);
      }
      if (anyB) {
        R.hxB = c2_LD1(hx, HB.off);
        R.hbB = c2_LD1(hb, HB.off);
        R.hdB = c2_LD1(hd, HB.off);
      }
      if (anyE) R.hxE = c2_LD1(hx, HE.off);
    }
    R.zcw = ZR ? g.sc[2][pk] : 0.;
    R.pl  = pl;
    const int kz = min(max(kk, -1), g.nz);
    R.zl  = g.sl[2][kz];
    R.zc  = g.sc[2][kz];
    R.zh  = g.sh[2][kz];
    R.pin = pin;
  };

  double  acc[6] = {0., 0., 0., 0., 0., 0.};
  double2 sb2[RY], d1p[RY];  // b of plane kk-2; d' of plane kk-2
  double2 zb1[RY], zb1R = make_double2(0., 0.);  // ZR: b' of plane kk-1 on the tile rows, the ring-1 row and the B item (formed one trip earlier)
  double  zb1B = 0.;
  double  z1l = 0., z1c = 0., z1h = 0., z2l = 0., z2c = 0., z2h = 0.;  // z rows of planes kk-1, kk-2
  bool    pin1 = false;
#pragma unroll
  for (int m = 0; m < RY; ++m) sb2[m] = d1p[m] = zb1[m] = make_double2(0., 0.);

  // DO1 / DO2: the first trips of a chunk only stage planes (kk = k0-2, k0-1), the next two form step 1 only
  auto step = [&](auto do1, auto do2, int kk, Raw &C, Raw &N) {
    constexpr bool DO1 = decltype(do1)::value, DO2 = decltype(do2)::value;
    load(kk + 1, N);
    // ZR: plane kk arrived as b (and q): form b' = b - za q, store it on the owned tile cells, and turn the x slots into x_1 = cf M b'
    double2 zbn[RY], zbnR = make_double2(0., 0.);
    double  zbnB = 0.;
    if constexpr (ZR) {
      const bool ownk = kk >= k0 && kk < k1;
#pragma unroll
      for (int m = 0; m < RY; ++m) {
        double2 bv = C.x[m];
        if (ZS) {
          bv.x -= za * C.d[m].x;
          bv.y -= za * C.d[m].y;
          if (ownk && rin[m] && in0) {
            if (in1) c2_ST2<NTS>(bw + rob[m] + C.pl, lo, bv);
            else c2_ST1(bw + rob[m] + C.pl, lo, bv.x);
          }
        }
        zbn[m] = bv;
        const double dyz = yc[m] + C.zcw;
        C.x[m].x = 0. + cf * (JAC ? bv.x / (xc0 + dyz) : bv.x);
        C.x[m].y = 0. + cf * (JAC ? bv.y / (xc1 + dyz) : bv.y);
      }
      if (rowA || rowC) {
        double2 bv = C.hxR;
        if (ZS) {
          bv.x -= za * C.hdR.x;
          bv.y -= za * C.hdR.y;
        }
        zbnR = bv;
        const double dyz = zrRy + C.zcw;
        C.hxR.x = 0. + cf * (JAC ? bv.x / (xc0 + dyz) : bv.x);
        C.hxR.y = 0. + cf * (JAC ? bv.y / (xc1 + dyz) : bv.y);
      }
      if (anyB) {
        double bv = C.hxB;
        if (ZS) bv -= za * C.hdB;
        zbnB  = bv;
        C.hxB = 0. + cf * (JAC ? bv / (zrBx + (zrBy + C.zcw)) : bv);
      }
      if (anyE) {
        double bv = C.hxE;
        if (ZS) bv -= za * C.hqE;
        C.hxE = 0. + cf * (JAC ? bv / (zrEx + (zrEy + C.zcw)) : bv);
      }
    }
    const int kc = kk - 1, k2 = kk - 2;
    const int lc = 2 * lane + 2;
    double2   x1v[RY], d1v[RY];
    double    hx1B = 0.;
    double2   hx1R = make_double2(0., 0.);
#pragma unroll
    for (int m = 0; m < RY; ++m) x1v[m] = d1v[m] = make_double2(0., 0.);

    // ---- step 1 on plane kc = kk-1: tile + ring 1 -------------------------------------------------------------------
    if constexpr (DO1) {
      const int  bc = (kc + 3) % 3, bp = (kc + 2) % 3;
      const bool own = kc >= k0 && kc < k1;
#pragma unroll
      for (int m = 0; m < RY; ++m) {
        const int     lr    = w * RY + m + 2;
        const double2 cen   = *reinterpret_cast<const double2 *>(&XJ[bc][lr][lc]);
        const double2 south = *reinterpret_cast<const double2 *>(&XJ[bc][lr - 1][lc]);
        const double2 north = *reinterpret_cast<const double2 *>(&XJ[bc][lr + 1][lc]);
        const double2 below = *reinterpret_cast<const double2 *>(&XJ[bp][lr][lc]);
        const double  west = XJ[bc][lr][lc - 1], east = XJ[bc][lr][lc + 2];
        const double  dyz = yc[m] + z1c;
        const double  v0 = (xc0 + dyz) * cen.x + xl0 * west + xh0 * cen.y + yl[m] * south.x + yh[m] * north.x + z1l * below.x + z1h * C.x[m].x;
        const double  v1 = (xc1 + dyz) * cen.y + xl1 * cen.x + xh1 * east + yl[m] * south.y + yh[m] * north.y + z1l * below.y + z1h * C.x[m].y;
        const double2 bm = ZR ? zb1[m] : C.b[m], dm = ZR ? cen : C.d[m];  // ZR: d_1 = x_1
        const double  r0 = bm.x - v0, r1 = bm.y - v1;
        const double  z0 = JAC ? r0 / (xc0 + dyz) : r0, z1 = JAC ? r1 / (xc1 + dyz) : r1;
        const double  e0 = (rho0 != 0. ? rho0 * dm.x : 0.) + c0 * z0;  // first step ever: d is not looked at
        const double  e1 = (rho0 != 0. ? rho0 * dm.y : 0.) + c0 * z1;
        const bool    ok0 = pin1 && rin[m] && in0, ok1 = pin1 && rin[m] && in1;
        d1v[m].x = e0;
        d1v[m].y = e1;
        x1v[m].x = ok0 ? cen.x + e0 : 0.;
        x1v[m].y = ok1 ? cen.y + e1 : 0.;
        if (!MGD && own && ok0) {
          acc[0] += z0;
          acc[1] += z0 * z0;
          acc[2] += r0 * r0;
        }
        if (!MGD && own && ok1) {
          acc[0] += z1;
          acc[1] += z1 * z1;
          acc[2] += r1 * r1;
        }
      }
      if (rowA) {  // step 1 on the wave's ring-1 row: the tile's arithmetic with the row's y coefficients
        const double2 cen   = *reinterpret_cast<const double2 *>(&XJ[bc][rlr][lc]);
        const double2 south = *reinterpret_cast<const double2 *>(&XJ[bc][rlr - 1][lc]);
        const double2 north = *reinterpret_cast<const double2 *>(&XJ[bc][rlr + 1][lc]);
        const double2 below = *reinterpret_cast<const double2 *>(&XJ[bp][rlr][lc]);
        const double  west = XJ[bc][rlr][lc - 1], east = XJ[bc][rlr][lc + 2];
        const double  dyz = Ayc + z1c;
        const double  v0 = (xc0 + dyz) * cen.x + xl0 * west + xh0 * cen.y + Ayl * south.x + Ayh * north.x + z1l * below.x + z1h * C.hxR.x;
        const double  v1 = (xc1 + dyz) * cen.y + xl1 * cen.x + xh1 * east + Ayl * south.y + Ayh * north.y + z1l * below.y + z1h * C.hxR.y;
        const double2 bm = ZR ? zb1R : C.hbR, dm = ZR ? cen : C.hdR;
        const double  r0 = bm.x - v0, r1 = bm.y - v1;
        const double  z0 = JAC ? r0 / (xc0 + dyz) : r0, z1 = JAC ? r1 / (xc1 + dyz) : r1;
        const double  e0 = (rho0 != 0. ? rho0 * dm.x : 0.) + c0 * z0;
        const double  e1 = (rho0 != 0. ? rho0 * dm.y : 0.) + c0 * z1;
        hx1R.x = (rinn0 && pin1) ? cen.x + e0 : 0.;
        hx1R.y = (rinn1 && pin1) ? cen.y + e1 : 0.;
      }
      if (anyB) {
        const double cen = XJ[bc][HB.lr][HB.lc], dyz = cB[tbc][4] + z1c;
        const double v = dyz * cen + cB[tbc][0] * XJ[bc][HB.lr][HB.lc - 1] + cB[tbc][1] * XJ[bc][HB.lr][HB.lc + 1] + cB[tbc][2] * XJ[bc][HB.lr - 1][HB.lc] + cB[tbc][3] * XJ[bc][HB.lr + 1][HB.lc] + z1l * XJ[bp][HB.lr][HB.lc] + z1h * C.hxB;
        const double r = (ZR ? zb1B : C.hbB) - v, z = JAC ? r / dyz : r;
        const double e = (rho0 != 0. ? rho0 * (ZR ? cen : C.hdB) : 0.) + c0 * z;
        hx1B = (HB.in && pin1) ? cen + e : 0.;
      }
    }

    // ---- step 2 on plane k2 = kk-2: tile --------------------------------------------------------------------------------
    if constexpr (DO2) {
      const int     bc = (k2 + 3) % 3, bp = (k2 + 2) % 3;
      const int64_t p2 = (int64_t)k2 * g.sxy;
#pragma unroll
      for (int m = 0; m < RY; ++m) {
        const int     lr    = w * RY + m + 1;
        const double2 cen   = *reinterpret_cast<const double2 *>(&XN[bc][lr][lc]);
        const double2 south = *reinterpret_cast<const double2 *>(&XN[bc][lr - 1][lc]);
        const double2 north = *reinterpret_cast<const double2 *>(&XN[bc][lr + 1][lc]);
        const double2 below = *reinterpret_cast<const double2 *>(&XN[bp][lr][lc]);
        const double  west = XN[bc][lr][lc - 1], east = XN[bc][lr][lc + 2];
        const double  dyz = yc[m] + z2c;
        const double  v0 = (xc0 + dyz) * cen.x + xl0 * west + xh0 * cen.y + yl[m] * south.x + yh[m] * north.x + z2l * below.x + z2h * x1v[m].x;
        const double  v1 = (xc1 + dyz) * cen.y + xl1 * cen.x + xh1 * east + yl[m] * south.y + yh[m] * north.y + z2l * below.y + z2h * x1v[m].y;
        const double  r0 = sb2[m].x - v0, r1 = sb2[m].y - v1;
        const double  z0 = JAC ? r0 / (xc0 + dyz) : r0, z1 = JAC ? r1 / (xc1 + dyz) : r1;
        double2       e, xo;
        e.x  = rho1 * d1p[m].x + c1 * z0;
        e.y  = rho1 * d1p[m].y + c1 * z1;
        xo.x = cen.x + e.x;
        xo.y = cen.y + e.y;
        if (rin[m] && in0) {
          if (in1) {
            c2_ST2<NTS>(dn + rob[m] + p2, lo, e);
            c2_ST2<NTS>(xn + rob[m] + p2, lo, xo);
            if (MGD) {
              acc[0] += xo.x + xo.y;
              acc[1] += xo.x * xo.x + xo.y * xo.y;
              acc[2] += sb2[m].x * xo.x + sb2[m].y * xo.y;
              acc[3] += sb2[m].x + sb2[m].y;
              acc[4] += sb2[m].x * sb2[m].x + sb2[m].y * sb2[m].y;
            } else {
              acc[3] += z0 + z1;
              acc[4] += z0 * z0 + z1 * z1;
              acc[5] += r0 * r0 + r1 * r1;
            }
          } else {
            c2_ST1(dn + rob[m] + p2, lo, e.x);
            c2_ST1(xn + rob[m] + p2, lo, xo.x);
            if (MGD) {
              acc[0] += xo.x;
              acc[1] += xo.x * xo.x;
              acc[2] += sb2[m].x * xo.x;
              acc[3] += sb2[m].x;
              acc[4] += sb2[m].x * sb2[m].x;
            } else {
              acc[3] += z0;
              acc[4] += z0 * z0;
              acc[5] += r0 * r0;
            }
          }
        }
      }
    }

    // ---- stage x(kk) and x'(kk-1) --------------------------------------------------------------------------------------
    {
      const int bj = (kk + 3) % 3, bn = (kc + 3) % 3;
#pragma unroll
      for (int m = 0; m < RY; ++m) {
        if (rin[m]) {
          if (in1) {
            *reinterpret_cast<double2 *>(&XJ[bj][w * RY + m + 2][lc]) = C.x[m];
            if constexpr (DO1) *reinterpret_cast<double2 *>(&XN[bn][w * RY + m + 1][lc]) = x1v[m];
          } else if (in0) {
            XJ[bj][w * RY + m + 2][lc] = C.x[m].x;
            if constexpr (DO1) XN[bn][w * RY + m + 1][lc] = x1v[m].x;
          }
        }
      }
      if (rok1) {
        *reinterpret_cast<double2 *>(&XJ[bj][rlr][lc]) = C.hxR;
        if constexpr (DO1)
          if (rowA) *reinterpret_cast<double2 *>(&XN[bn][rlr - 1][lc]) = hx1R;  // XN rows are XJ rows - 1
      } else if (rok0) {
        XJ[bj][rlr][lc] = C.hxR.x;
        if constexpr (DO1)
          if (rowA) XN[bn][rlr - 1][lc] = hx1R.x;
      }
      if (HB.ok) {
        XJ[bj][HB.lr][HB.lc]     = C.hxB;
        if constexpr (DO1) XN[bn][HB.lr - 1][HB.lc] = hx1B;
      }
      if (HE.ok) XJ[bj][HE.lr][HE.lc] = C.hxE;
    }
    __syncthreads();
#pragma unroll
    for (int m = 0; m < RY; ++m) {
      sb2[m] = ZR ? zb1[m] : C.b[m];
      d1p[m] = d1v[m];
      if (ZR) zb1[m] = zbn[m];
    }
    if (ZR) {
      zb1R = zbnR;
      zb1B = zbnB;
    }
    z2l  = z1l;
    z2c  = z1c;
    z2h  = z1h;
    z1l  = C.zl;
    z1c  = C.zc;
    z1h  = C.zh;
    pin1 = C.pin;
  };

  {
    using T = std::true_type;
    using F = std::false_type;
    Raw A, B;
    __syncthreads();  // cB
    load(k0 - 2, A);
    step(F(), F(), k0 - 2, A, B);
    step(F(), F(), k0 - 1, B, A);
    step(T(), F(), k0, A, B);
    step(T(), F(), k0 + 1, B, A);  // k0 + 1 <= k1: forms step 1 on the last owned plane or, in a one-plane chunk, on the ring plane k1
    for (int kk = k0 + 2; kk <= k1 + 1; kk += 2) {
      step(T(), T(), kk, A, B);
      if (kk + 1 <= k1 + 1) step(T(), T(), kk + 1, B, A);
    }
  }

  // fixed-order block sums
#pragma unroll
  for (int a = 0; a < 6; ++a) {
    acc[a] = wave_sum(acc[a]);
    if (lane == 0) red[a * NW + w] = acc[a];
  }
  __syncthreads();
  if (tid == 0) {
#pragma unroll
    for (int a = 0; a < 6; ++a) {
      double t = 0.;
#pragma unroll
      for (int q = 0; q < NW; ++q) t += red[a * NW + q];
      partial[(int64_t)a * stride + blockIdx.x] = t;
    }
  }
}

}  // namespace fl

using namespace fl;

// ------------------------------------------------------------------------------------------------ plan + launch

// The fused kernel needs every ring cell inside this rank's block (or no cell at all, at a wall): single rank only, at
// least two cells along a periodic axis, planes addressable with 32-bit offsets.
bool fl_cheb2_usable(const fl_poisson *h)
{
  // several ranks: the wide layout and a communicator for the two-deep exchange (not the one-rank loopback rehearsal: its layout is narrow)
  if (h->multi && (h->gw < 2 || h->loopback || h->comm.kind == Comm::NONE)) return false;
  const GridP &g = h->g;
  const int    n[3] = {g.nx, g.ny, g.nz};
  for (int d = 0; d < 3; ++d)
    if (n[d] < 2) return false;
  if (g.sxy + 4 * (int64_t)g.sx >= ((int64_t)1 << 28)) return false;  // 32-bit byte offsets inside a plane
  return true;
}

// Several ranks: every rank must take the same branch -- the fused path exchanges TWO layers of x per pair of steps under its own tags, the
// one-step path one layer per step, and fl_fill_ghosts_deep takes for granted that the NEIGHBOUR owns two layers too.  "Legal" and "large
// enough" are per-block quantities (an uneven split can straddle the threshold: 64 x 32 x 31 over two z ranks is 32768 against 30720 cells), so
// the ranks vote once per handle (and per multigrid level, whose handles borrow the communicator): fused only where nobody objects.
int fl_cheb2_agree(fl_poisson *h)
{
  if (h->cheb2_agreed[0] >= 0) return 0;
  const bool legal = fl_cheb2_usable(h);
  double     no[2] = {legal ? 0. : 1., (legal && h->ncell >= 32768) ? 0. : 1.};  // objections, summed over the ranks
  if (h->multi) {
    FL_CHK(fl_allreduce_sum(h, &no[0]));
    FL_CHK(fl_allreduce_sum(h, &no[1]));
  }
  h->cheb2_agreed[0] = no[0] == 0. ? 1 : 0;
  h->cheb2_agreed[1] = no[1] == 0. ? 1 : 0;
  return 0;
}

Cheb2Plan fl_cheb2_plan(const GridP &g)
{
  Cheb2Plan p;
  const int force_nw = FL_VARIANT(cheb2_nw, 0);  // experiments: 4 = 128 x 8 tiles (two 256-thread blocks per CU), 8 = 128 x 16
  p.nw      = (force_nw == 4 || force_nw == 8) ? force_nw : (g.ny > 8 ? 8 : 4);
  p.tiles_x = (g.nx + 127) / 128;
  const int ty    = 2 * p.nw;
  const int tiles = p.tiles_x * ((g.ny + ty - 1) / ty);
  p.tiles         = tiles;
  // one block per CU is resident (120 KB of LDS): aim at one full wave of 256 blocks; a chunk re-reads 4 planes of x and 2 of
  // b and d, so chunks stay >= 16 planes
  const int force = FL_VARIANT(cheb2_nchunk, 0);
  int nchunk = force > 0 ? force : std::max(1, (256 + tiles / 2) / tiles);
  if (force <= 0 && tiles * nchunk > 256 && tiles <= 256) nchunk = std::max(1, 256 / tiles);  // never a second round of blocks (384^3: 72 tiles x 4 chunks = 288 -> 216;
                                                                                             // the CG pair gained 35 % from the same rule, profiles/r04_cg_plans.txt)
  if (force <= 0) nchunk = std::min(nchunk, std::max(1, g.nz / 16));
  nchunk    = std::max(1, std::min(nchunk, g.nz));
  p.zc      = (g.nz + nchunk - 1) / nchunk;
  p.nchunk  = (g.nz + p.zc - 1) / p.zc;
  p.nblocks = tiles * p.nchunk;
  return p;
}

template <int NW, bool JAC, bool MGD = false, int Z = 0>
static void cheb2_t(fl_poisson *h, const Cheb2Plan &p, double *X0, double *X1, const double *B, double *D0, double *D1, const double *zq = nullptr, const double *za_dev = nullptr, double *bw = nullptr)
{
  int per[3];
  for (int d = 0; d < 3; ++d) {  // per side: 0 wall, 1 periodic image inside the block, 2 a neighbouring rank (ghost layers)
    int m[2];
    for (int side = 0; side < 2; ++side) m[side] = (h->multi && h->nbr[2 * d + side] >= 0 && !h->wrap_local[d]) ? 2 : (h->ax[d].periodic ? 1 : 0);
    per[d] = m[0] | (m[1] << 2);
  }
  hipLaunchKernelGGL((k_cheb2<2, NW, JAC, 2, MGD, Z>), dim3(p.nblocks), dim3(64 * NW), 0, h->stream, h->g, per[0], per[1], per[2], X0, X1, X0, X1, B, D0, D1, D0, D1, h->scal, h->partial, h->partial_stride, p.zc, p.tiles_x, p.tiles, 1,
                     zq, za_dev, bw);
}

// three steps from a zero initial guess in one sweep (one rank): h->scal is the block of the FIRST step; the answer x_3 lands in the buffer the
// block's `cur` does not name, d_3 likewise.  subq: b' = B - *suba_dev * subq is what the steps see, and it is written to Bw (never B itself)
void fl_launch_cheb2_from_zero(fl_poisson *h, const Cheb2Plan &p, bool jac, double *X0, double *X1, const double *B, double *D0, double *D1, const double *subq, const double *suba_dev, double *Bw)
{
  if (subq) {
    if (p.nw == 8) {
      if (jac) cheb2_t<8, true, false, 2>(h, p, X0, X1, B, D0, D1, subq, suba_dev, Bw);
      else cheb2_t<8, false, false, 2>(h, p, X0, X1, B, D0, D1, subq, suba_dev, Bw);
    } else {
      if (jac) cheb2_t<4, true, false, 2>(h, p, X0, X1, B, D0, D1, subq, suba_dev, Bw);
      else cheb2_t<4, false, false, 2>(h, p, X0, X1, B, D0, D1, subq, suba_dev, Bw);
    }
    return;
  }
  if (p.nw == 8) {
    if (jac) cheb2_t<8, true, false, 1>(h, p, X0, X1, B, D0, D1);
    else cheb2_t<8, false, false, 1>(h, p, X0, X1, B, D0, D1);
  } else {
    if (jac) cheb2_t<4, true, false, 1>(h, p, X0, X1, B, D0, D1);
    else cheb2_t<4, false, false, 1>(h, p, X0, X1, B, D0, D1);
  }
}

void fl_launch_cheb2(fl_poisson *h, const Cheb2Plan &p, bool jac, double *X0, double *X1, const double *B, double *D0, double *D1, bool mgdots)
{
  if (mgdots) {  // the multigrid smoother is the Jacobi one
    if (p.nw == 8) cheb2_t<8, true, true>(h, p, X0, X1, B, D0, D1);
    else cheb2_t<4, true, true>(h, p, X0, X1, B, D0, D1);
    return;
  }
  if (p.nw == 8) {
    if (jac) cheb2_t<8, true>(h, p, X0, X1, B, D0, D1);
    else cheb2_t<8, false>(h, p, X0, X1, B, D0, D1);
  } else {
    if (jac) cheb2_t<4, true>(h, p, X0, X1, B, D0, D1);
    else cheb2_t<4, false>(h, p, X0, X1, B, D0, D1);
  }
}
