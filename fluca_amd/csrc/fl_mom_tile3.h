// fl_mom_tile3.h -- k_mom3: k_mom2 with the nine v0interp face fields interpolated inside the kernel.
//
//   v0interp = B v0 + vbc        cnlinearcart3d.c:2826-2829 (MatMult(cnl->B, v0, cnl->v0interp); VecAXPY(v0interp, 1, vbc))
//   B                            ComputeFaceInterpolationOperator rows, restated in build_axis_faceinterp (fl_coeff.cpp):
//                                on an inner face f of an axis  w0[f] v0[f-1] + w1[f] v0[f]  (normal rule for the component along
//                                the axis, tangential rule otherwise); vbc is zero there.
//
// A cell's row needs v0interp_{c,d} on its six faces: 9 of the 15 streams k_mom2 reads.  Here the three components of v0 ride beside
// the three of x (same registers / LDS planes / ring), and every INNER face value is formed from two of them with the B row of that
// face -- the same two products in the same order as k_face_interp, so the numbers equal the stored ones bit for bit.  Faces at an end
// of this rank's block (face 0 and face n of an axis: wall rules, outlet extrapolation, vbc, the periodic image, a neighbour rank's
// cell) are still READ from the stored fields: a ring item per row for the x axis, one staged row for the y axis, a direct load on
// the first / last plane for the z axis.  Streams per cell: x 3, y 3, v0 3, V0 3 = 96 B where k_mom2 moves 144.
//
// Differences from k_mom2 besides that: the x axis is worked first (its general rows at the two end cells REPLACE the fast-path
// result of that lane instead of a whole tile taking the scalar path), ring rows travel as 16-byte pairs.
#pragma once

namespace fl {

// FL_MOM3_NT (experiment): 1 = the per-plane loads of the tile's own cells carry the non-temporal hint (they are read once by this block; what a
// neighbouring block reads again are the tile's edge rows / columns), 2 = only the waves of the tile's inner rows do, 0 = plain loads
#ifndef FL_MOM3_NT
#define FL_MOM3_NT 0
#endif
__device__ __forceinline__ double2 LD2s(const double *base, unsigned byteoff, bool inner)
{
  const double *p = reinterpret_cast<const double *>(reinterpret_cast<const char *>(base) + byteoff);
  if (FL_MOM3_NT == 1 || (FL_MOM3_NT == 2 && inner)) return ld2<1>(p);
  return ld2<0>(p);
}

template <int NW>
struct Mom3Lds {
  static constexpr int TX = 128, TY = NW, LXU = TX + 4;
  double u[2][3][TY + 2][LXU];  // x plane incl. its ring: column ii = -1..128 at index ii + 2, row jj = -1..TY at jj + 1
  double v[2][3][TY + 2][LXU];  // v0, same layout
  double fy[2][TY + 1][TX];     // V0 on the low y-faces of rows 0..TY
  double fyw[2][3][TX];         // stored v0interp_{c,y} on the tile's block-end y-face (face 0 if the tile starts at row 0, else face ny)
  double fxe[2][7][TY];         // stored x-face values: [0] V0 at column min(i0 + TX, nx); [1 + c] v0interp_{c,x} there; [4 + c] at column i0
  double tabx[7][TX];           // fast-path numbers of the tile's columns (after the loop: the block's partial sums)
  double tabw[4][TX + 2];       // B rows on the low x-face of columns 0..TX: w0, w1 of the normal rule, w0, w1 of the tangential rule
  double tabg[2][MOM_NTAB];     // general x rows of cell 0 and of cell nx - 1
};
static_assert(sizeof(Mom3Lds<8>) <= 160 * 1024, "one block per CU: the tile must fit the LDS");

template <int NW, int DOT, bool JAC, int OUT, int NT>
__global__ void __launch_bounds__(64 * NW, 2) k_mom3(GridP g, MomP m, const double *__restrict__ x, double *__restrict__ y, const double *__restrict__ F, const double *__restrict__ v0,
                                                     int64_t cs, const double *__restrict__ o, const KspScal *__restrict__ s, double *__restrict__ partial, int pstride, int tiles_x, int nchunk, int zc,
                                                     int flags)
{
  static_assert(NW == 8, "the ring plan below is laid out for 512 threads");
  using LT              = Mom3Lds<NW>;
  constexpr int TX = LT::TX, TY = LT::TY, NTH = 64 * NW;
  constexpr bool AB = OUT == 3;  // OUT == 3: y = (sum of |a_ij|, j != i) / |a_ii| (see k_mom2)
  // OUT == 4: one step of KSPCHEBYSHEV fused into the product (PETSc's three-term recurrence, x_{k+1} = x_k + rho (x_k - x_{k-1}) + c M (b - A x_k) with
  // rho = omega - 1, c = omega Gamma scale from KspScal): x = x_k (staged with its ring as ever), o = b, y holds x_{k-1} on entry and x_{k+1} on
  // exit (each lane reads and writes its own pair only), M = 1 / diag(A) with JAC.  Partial slots: 0 sum z, 1 z.z, 2 r.r (z = M r, r = b - A x_k):
  // what k_cheb_fin tests.  144 B/cell per step where a BiCGStab iteration moves 552 for two products.
  constexpr bool CH = OUT == 4;
  constexpr bool DG = JAC || OUT == 2 || AB;
  __shared__ __attribute__((aligned(16))) LT lds;
  if (s && s->reason != 0) return;

  int b = blockIdx.x, chunk, tile;
  {
    const int nb = gridDim.x, tiles = nb / nchunk;
    if (flags & 1) b = xcd_remap(b, nb);
    chunk = b / tiles;
    tile  = b % tiles;
  }
  const int  tid = threadIdx.x, lane = tid & 63, w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int  i0 = (tile % tiles_x) * TX, j0 = (tile / tiles_x) * TY, j = j0 + w, i = i0 + 2 * lane;
  const bool own0 = i < g.nx, own1 = i + 1 < g.nx, rown = j < g.ny;
  const int      il = min(i, g.nx & ~1), jl = min(j, g.ny), jt = min(j, g.ny - 1);
  const int      k0 = chunk * zc, k1 = min(k0 + zc, g.nz);
  const unsigned lo0 = (unsigned)il * 8u;
  const bool     xlo = i0 == 0, xhi = i0 + TX >= g.nx;  // block-uniform: the tile holds cell 0 / cell nx - 1
  const bool     ylo = jt == 0, yhi = jt == g.ny - 1;   // wave-uniform (the host only launches this kernel when ny > TY: never both)
  const bool     ywall = ylo || yhi;
  const bool     inner = w != 0 && w != NW - 1;  // wave-uniform: the tile's edge rows are what the y-neighbour tiles read again as their ring
  const int64_t  sx = g.sx, sxy = g.sxy;
  const double   cI = m.cI;
  const double   ch_rho = CH ? s->cheb_rho : 0., ch_c = CH ? s->cheb_c : 0.;
  const int64_t  ncell = (int64_t)g.nx * g.ny * g.nz;
  const int64_t  rb0 = g.off0 + (int64_t)jl * sx;

  // ---- x-axis tables of the tile's columns
  for (int q = tid >> 7; q < 7; q += NTH / 128) lds.tabx[q][tid & 127] = m.stab[0][(int64_t)min(i0 + (tid & 127), g.nx - 1) * MOM_STAB + q];
  for (int q = tid; q < 4 * (TX + 2); q += NTH) {
    const int r = q / (TX + 2), col = q % (TX + 2), f = min(i0 + col, g.fx - 1);  // face nx of a block without one is never used (block end: stored value)
    lds.tabw[r][col] = m.bw[0][(int64_t)4 * f + r];
  }
  if (tid < 2 * MOM_NTAB) lds.tabg[tid / MOM_NTAB][tid % MOM_NTAB] = m.stab[0][(int64_t)(tid < MOM_NTAB ? 0 : g.nx - 1) * MOM_STAB + 8 + tid % MOM_NTAB];
  // ---- y axis: the numbers of this wave's row (fixed for the whole chunk), scalar loads
  const cdouble4 *taby = as_const(m.stab[1]) + (int64_t)jt * MOM_STAB;
  double          tyi[7];
#pragma unroll
  for (int a = 0; a < 7; ++a) tyi[a] = taby[a];
  const int fyl_ = min(jt, g.fy - 1), fyh_ = min(jt + 1, g.fy - 1);  // B rows of the row's low / high y-face

  // far column / row of the one-sided wall rows (see k_mom2)
  int fxi[2];
#pragma unroll
  for (int a = 0; a < 2; ++a) {
    const int ia = min(i + a, g.nx - 1), fi = ia == 0 ? ia + 2 : ia - 2;
    fxi[a]       = min(max(fi - i0 + 2, 1), TX + 2);
  }
  const bool xfar_g = xhi && g.nx - 1 == i0;
  const int  fj = jt == 0 ? jt + 2 : jt - 2, fyr = min(max(fj - j0 + 1, 0), TY + 1);
  const bool yfar_g = yhi && g.ny - 1 == j0;

  // ---- ring plan.  Rows: 16 row-loads of 64 lanes x 16 bytes, two per wave (wave-uniform kind):
  //   0..5 x rows (component r >> 1, side r & 1), 6..11 v0 rows, 12 V0 on the y-face behind the tile, 13..15 stored v0interp_{c,y} on the
  //   tile's block-end y-face.  Columns: 160 single items on the first 160 threads: x columns 48, v0 columns 48, stored x-faces 64.
  const double *rp2[2];              // wave-uniform: the row's column 0 in plane 0 (the lane adds lo)
  unsigned      rl2[2], rlb2[2];     // wave-uniform: LDS byte offset of the row's lane-0 slot in buffer 0, and what buffer 1 adds
  const double *rp1;
  unsigned      rl1, rlb1;
  {
    char *const lb = reinterpret_cast<char *>(&lds);
#pragma unroll
    for (int sidx = 0; sidx < 2; ++sidx) {
      const int     r = w + 8 * sidx;
      const double *base;
      int           row;
      const char   *dst;
      unsigned      bufb;
      if (r < 12) {
        const int  rr = r < 6 ? r : r - 6, c = rr >> 1, side = rr & 1;
        base = (r < 6 ? x : v0) + (int64_t)c * cs;
        row  = side ? min(j0 + TY, g.ny) : j0 - 1;
        dst  = r < 6 ? reinterpret_cast<const char *>(&lds.u[0][c][side ? TY + 1 : 0][2]) : reinterpret_cast<const char *>(&lds.v[0][c][side ? TY + 1 : 0][2]);
        bufb = sizeof(lds.u[0]);
      } else if (r == 12) {
        base = F + cs;
        row  = min(j0 + TY, g.ny);
        dst  = reinterpret_cast<const char *>(&lds.fy[0][TY][0]);
        bufb = sizeof(lds.fy[0]);
      } else {
        // only a tile at an end of the y axis reads its stored row; an inner tile re-reads the V0 row next door (a cache hit, never used)
        const int  c = r - 13;
        const bool yend = j0 == 0 || j0 + TY >= g.ny;
        base = yend ? F + (int64_t)(3 * (c + 1) + 1) * cs : F + cs;
        row  = j0 == 0 ? 0 : min(j0 + TY, g.ny);
        dst  = reinterpret_cast<const char *>(&lds.fyw[0][c][0]);
        bufb = sizeof(lds.fyw[0]);
      }
      rp2[sidx]  = base + g.off0 + (int64_t)row * sx;
      rl2[sidx]  = __builtin_amdgcn_readfirstlane((unsigned)(dst - lb));
      rlb2[sidx] = bufb;
    }
    {
      const double *base;
      int           row, col;
      const char   *dst;
      unsigned      bufb;
      if (tid < 12 * TY) {
        const int qq = tid % (6 * TY), c = qq / (2 * TY), side = (qq / TY) & 1, jj = qq % TY;
        const bool isu = tid < 6 * TY;
        base = (isu ? x : v0) + (int64_t)c * cs;
        row  = min(j0 + jj, g.ny);
        col  = side ? min(i0 + TX, g.nx) : i0 - 1;
        dst  = isu ? reinterpret_cast<const char *>(&lds.u[0][c][jj + 1][side ? TX + 2 : 1]) : reinterpret_cast<const char *>(&lds.v[0][c][jj + 1][side ? TX + 2 : 1]);
        bufb = sizeof(lds.u[0]);
      } else if (tid < 12 * TY + 7 * TY) {
        const int qq = tid - 12 * TY, e = qq / TY, jj = qq % TY, f = e < 4 ? e : e - 3;  // e: 0 V0 hi, 1..3 v0interp hi, 4..6 v0interp lo
        // stored v0interp: only a tile at an end of the x axis reads them; an inner tile re-reads V0 (same line as item e = 0 / the lane's own)
        base = F + (int64_t)(3 * ((e >= 4 ? xlo : xhi) ? f : 0)) * cs;
        row  = min(j0 + jj, g.ny);
        col  = e < 4 ? min(i0 + TX, g.nx) : i0;
        dst  = reinterpret_cast<const char *>(&lds.fxe[0][e][jj]);
        bufb = sizeof(lds.fxe[0]);
      } else {  // no item: a valid address and the dead corner slot (row -1, column -2) nobody reads
        base = x;
        row  = 0;
        col  = 0;
        dst  = reinterpret_cast<const char *>(&lds.u[0][0][0][0]);
        bufb = sizeof(lds.u[0]);
      }
      rp1  = base + g.off0 + (int64_t)row * sx + col;
      rl1  = (unsigned)(dst - lb);
      rlb1 = bufb;
    }
  }

  // ---- registers of the z march
  double2 uA[3], uB[3], uC[3], vA[3], vB[3], vC[3], fzA, fzB, fzC, fxl, fyl;  // planes k - 1, k, k + 1 of x and v0; V0 on z-faces k, k + 1, (k + 2)
  double2 rv2[2];
  double  rv1;
  double  acc[4] = {0., 0., 0., 0.};
  {
    // issued in the order the loop leaves its loads behind (ring, x-faces, y-faces, then the plane ahead): the wait counts at the loop's
    // head are the minimum over both ways in, and a prologue that ends with the ring would make every third plane wait for everything
    const int64_t rb = rb0 + (int64_t)k0 * sxy;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      const double *X = x + (int64_t)c * cs, *V = v0 + (int64_t)c * cs;
      uA[c] = LD2(X + rb - sxy, lo0);
      uB[c] = LD2(X + rb, lo0);
      vA[c] = LD2(V + rb - sxy, lo0);
      vB[c] = LD2(V + rb, lo0);
    }
    fzA = LD2(F + 2 * cs + rb, lo0);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int sidx = 0; sidx < 2; ++sidx) rv2[sidx] = LD2(rp2[sidx] + (int64_t)k0 * sxy, lo0);
    rv1 = rp1[(int64_t)k0 * sxy];
    __builtin_amdgcn_sched_barrier(0);
    fxl = LD2(F + rb, lo0);
    fyl = LD2(F + cs + rb, lo0);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      uC[c] = LD2(x + (int64_t)c * cs + rb + sxy, lo0);
      vC[c] = LD2(v0 + (int64_t)c * cs + rb + sxy, lo0);
    }
    fzB = LD2(F + 2 * cs + rb + sxy, lo0);
    fzC = fzB;
  }
  __syncthreads();  // tables

  // One plane.  The z march is unrolled three times below with the roles of the register sets rotated by NAME: a rotation by copies
  // (zm = cc; cc = zp; zp = load) makes the compiler load plane k + 2 into a scratch register and copy it at the loop's back edge --
  // a wait for the loads just issued, i.e. no prefetch at all.  The loads of plane k + 2 go straight into the set that held plane k - 1.
  auto plane = [&](int kl, double2 (&uzm)[3], double2 (&ucc)[3], double2 (&uzp)[3], double2 (&vzm)[3], double2 (&vcc)[3], double2 (&vzp)[3], double2 &fzl, double2 &fzh, double2 &fzn)
                   __attribute__((always_inline)) {
    double2  oc[3], xo[3];
    int      k = kl;
    unsigned lo = lo0;
    int64_t  csl = cs;
    asm volatile("" : "+s"(k), "+v"(lo), "+s"(csl));
    const int     bf = k & 1;
    const int64_t rb = rb0 + (int64_t)k * sxy;
    const int64_t rb1 = rb + sxy, rb2 = rb0 + (int64_t)min(k + 2, g.nz) * sxy;
    const bool    zlo = k == 0, zhi = k == g.nz - 1;

    // ---- A: publish this wave's row of plane k and the ring items fetched for it
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      *reinterpret_cast<double2 *>(&lds.u[bf][c][w + 1][2 * lane + 2]) = ucc[c];
      *reinterpret_cast<double2 *>(&lds.v[bf][c][w + 1][2 * lane + 2]) = vcc[c];
    }
    *reinterpret_cast<double2 *>(&lds.fy[bf][w][2 * lane]) = fyl;
    {
      char *const lb = reinterpret_cast<char *>(&lds);
#pragma unroll
      for (int sidx = 0; sidx < 2; ++sidx) *reinterpret_cast<double2 *>(lb + (rl2[sidx] + (bf ? rlb2[sidx] : 0u)) + 16u * (unsigned)lane) = rv2[sidx];
      *reinterpret_cast<double *>(lb + rl1 + (bf ? rlb1 : 0u)) = rv1;
    }
    // ---- B: the ring of plane k + 1
    {
      const int64_t pn = (int64_t)min(k + 1, g.nz) * sxy;
#pragma unroll
      for (int sidx = 0; sidx < 2; ++sidx) rv2[sidx] = LD2(rp2[sidx] + pn, lo);
      rv1 = rp1[pn];
    }
    __syncthreads();

    double2 ya[3], da[3] = {{0., 0.}, {0., 0.}, {0., 0.}}, aa[3] = {{0., 0.}, {0., 0.}, {0., 0.}};
    // ---- F: the x axis (first: the general rows of the two end cells overwrite what the fast path left in their lanes)
    {
      double2 tx[7];
#pragma unroll
      for (int q = 0; q < 7; ++q) tx[q] = *reinterpret_cast<const double2 *>(&lds.tabx[q][2 * lane]);
      MomRow r;
      double vhx;  // V0 on the high face of the lane's second cell
      {
        const double uw = lds.u[bf][0][w + 1][2 * lane + 1], ue = lds.u[bf][0][w + 1][2 * lane + 4];
        vhx = from_next_lane(fxl.x, lds.fxe[bf][0][w]);
        r   = mom_row_coef(tx, fxl, make_double2(fxl.y, vhx), make_double2(uw, ucc[0].x), ucc[0], make_double2(ucc[0].y, ue));
      }
#pragma unroll
      for (int c = 0; c < 3; ++c) {
        const int     kr = c == 0 ? 0 : 2;
        const double2 a0 = *reinterpret_cast<const double2 *>(&lds.tabw[kr][2 * lane]), a1 = *reinterpret_cast<const double2 *>(&lds.tabw[kr + 1][2 * lane]);
        const double  a0c = lds.tabw[kr][2 * lane + 2], a1c = lds.tabw[kr + 1][2 * lane + 2];
        const double  uw = lds.u[bf][c][w + 1][2 * lane + 1], ue = lds.u[bf][c][w + 1][2 * lane + 4];
        const double  vw = lds.v[bf][c][w + 1][2 * lane + 1], ve = lds.v[bf][c][w + 1][2 * lane + 4];
        // v0interp_{c,x} on the three x-faces of the lane's pair (at a block end: the stored value)
        double fA = fma(a1.x, vcc[c].x, a0.x * vw), fB = fma(a1.y, vcc[c].y, a0.y * vcc[c].x), fC = fma(a1c, ve, a0c * vcc[c].y);
        if (xlo && i == 0) fA = lds.fxe[bf][4 + c][w];
        if (xhi) {
          if (i + 1 == g.nx) fB = lds.fxe[bf][1 + c][w];
          if (i + 2 == g.nx) fC = lds.fxe[bf][1 + c][w];
        }
        ya[c] = make_double2(0., 0.);
        mom_row_comp(r, make_double2(uw, ucc[c].x), ucc[c], make_double2(ucc[c].y, ue), make_double2(fA, fB), make_double2(fB, fC), ya[c]);
        if (AB) aa[c] = mom_row_abs(r, tx[3], tx[4], tx[5], tx[6], make_double2(fA, fB), make_double2(fB, fC), c == 0);
        if (DG) {
          da[c] = r.ac;
          if (c == 0) da[0] = fma2(make_double2(fB, fC), tx[5], fma2(make_double2(fA, fB), tx[4], da[0]));
        }
        __builtin_amdgcn_sched_barrier(0);
      }
      if (xlo || xhi) {
#pragma unroll
        for (int a = 0; a < 2; ++a) {
          const int cell = i + a;
          if (cell == 0 || cell == g.nx - 1) {
            const int side = cell == 0 ? 0 : 1;
            double    um[3], uc[3], up[3], uf[3], wl[3], wh[3], y_[3] = {0., 0., 0.}, d_[3] = {0., 0., 0.}, a_[3] = {0., 0., 0.};
#pragma unroll
            for (int c = 0; c < 3; ++c) {
              um[c] = a ? ucc[c].x : lds.u[bf][c][w + 1][2 * lane + 1];
              uc[c] = a ? ucc[c].y : ucc[c].x;
              up[c] = a ? lds.u[bf][c][w + 1][2 * lane + 4] : ucc[c].y;
              {  // the cell's two face values again: inner faces from v0, the block-end face stored
                const int    kr = c == 0 ? 0 : 2, col = 2 * lane + a;
                const double vm = a ? vcc[c].x : lds.v[bf][c][w + 1][2 * lane + 1], vc = a ? vcc[c].y : vcc[c].x, vp = a ? lds.v[bf][c][w + 1][2 * lane + 4] : vcc[c].y;
                wl[c] = fma(lds.tabw[kr + 1][col], vc, lds.tabw[kr][col] * vm);
                wh[c] = fma(lds.tabw[kr + 1][col + 1], vp, lds.tabw[kr][col + 1] * vc);
                if (cell == 0) wl[c] = lds.fxe[bf][4 + c][w];
                if (cell == g.nx - 1) wh[c] = lds.fxe[bf][1 + c][w];
              }
              uf[c] = lds.u[bf][c][w + 1][fxi[a]];
              if (xfar_g) uf[c] = (x + (int64_t)c * csl + rb)[max(g.nx - 3, -1)];  // single-column tile (rare; a synchronous load)
            }
            const double vl = a ? fxl.y : fxl.x, vh = a ? vhx : fxl.y;
            auto         T = [&](int q) { return lds.tabg[side][q]; };
            mom_row_wall<0, DG, AB>(T, um, uc, up, uf, vl, vh, wl, wh, y_, d_, a_);
#pragma unroll
            for (int c = 0; c < 3; ++c) {
              if (a) {
                ya[c].y = y_[c];
                if (DG) da[c].y = d_[c];
                if (AB) aa[c].y = a_[c];
              } else {
                ya[c].x = y_[c];
                if (DG) da[c].x = d_[c];
                if (AB) aa[c].x = a_[c];
              }
            }
          }
        }
      }
    }
    __builtin_amdgcn_sched_barrier(0);
    // ---- G: V0 on the x-faces of plane k + 1
    fxl = LD2s(F + rb1, lo, inner);
    __builtin_amdgcn_sched_barrier(0);

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
    {
      // B rows of the row's two y-faces: [0] normal rule (component 1), [1] tangential rule
      int fql = fyl_, fqh = fyh_;
      asm volatile("" : "+s"(fql), "+s"(fqh));  // re-read per plane (scalar cache) instead of 16 more SGPRs carried around the loop
      const cdouble4 *bl = as_const(m.bw[1]) + 4 * fql, *bh = as_const(m.bw[1]) + 4 * fqh;
      const double    wl0[2] = {bl[0], bl[2]}, wl1[2] = {bl[1], bl[3]}, wh0[2] = {bh[0], bh[2]}, wh1[2] = {bh[1], bh[3]};
      const double2 vhy = *reinterpret_cast<const double2 *>(&lds.fy[bf][w + 1][2 * lane]);
      if (ywall) {
        double2 us[3], un[3], wlp[3], whp[3];
#pragma unroll
        for (int c = 0; c < 3; ++c) {
          const int     kr = c == 1 ? 0 : 1;
          const double2 vs = *reinterpret_cast<const double2 *>(&lds.v[bf][c][w][2 * lane + 2]), vn = *reinterpret_cast<const double2 *>(&lds.v[bf][c][w + 2][2 * lane + 2]);
          us[c]  = *reinterpret_cast<const double2 *>(&lds.u[bf][c][w][2 * lane + 2]);
          un[c]  = *reinterpret_cast<const double2 *>(&lds.u[bf][c][w + 2][2 * lane + 2]);
          wlp[c] = fma2(vcc[c], wl1[kr], mul2(vs, wl0[kr]));
          whp[c] = fma2(vn, wh1[kr], mul2(vcc[c], wh0[kr]));
          const double2 st = *reinterpret_cast<const double2 *>(&lds.fyw[bf][c][2 * lane]);
          if (ylo) wlp[c] = st;
          if (yhi) whp[c] = st;
        }
#pragma unroll
        for (int a = 0; a < 2; ++a) {
          double um[3], uc[3], up[3], uf[3], wl[3], wh[3];
#pragma unroll
          for (int c = 0; c < 3; ++c) {
            um[c] = a ? us[c].y : us[c].x;
            uc[c] = a ? ucc[c].y : ucc[c].x;
            up[c] = a ? un[c].y : un[c].x;
            wl[c] = a ? wlp[c].y : wlp[c].x;
            wh[c] = a ? whp[c].y : whp[c].x;
            uf[c] = lds.u[bf][c][fyr][2 * lane + 2 + a];
            if (yfar_g) uf[c] = (x + (int64_t)c * csl + g.off0 + (int64_t)k * sxy + (int64_t)max(g.ny - 3, -1) * sx)[il + a];
          }
          const double vl = a ? fyl.y : fyl.x, vh = a ? vhy.y : vhy.x;
          auto         T = [&](int q) { return (double)taby[8 + q]; };
          MOM_WALL_CELL(1, T, um, uc, up, uf, vl, vh, wl, wh)
        }
      } else {
        const double2 us1 = *reinterpret_cast<const double2 *>(&lds.u[bf][1][w][2 * lane + 2]);
        const double2 un1 = *reinterpret_cast<const double2 *>(&lds.u[bf][1][w + 2][2 * lane + 2]);
        const MomRow  r   = mom_row_coef(tyi, fyl, vhy, us1, ucc[1], un1);
#pragma unroll
        for (int c = 0; c < 3; ++c) {
          const int     kr = c == 1 ? 0 : 1;
          const double2 us = c == 1 ? us1 : *reinterpret_cast<const double2 *>(&lds.u[bf][c][w][2 * lane + 2]);
          const double2 un = c == 1 ? un1 : *reinterpret_cast<const double2 *>(&lds.u[bf][c][w + 2][2 * lane + 2]);
          const double2 vs = *reinterpret_cast<const double2 *>(&lds.v[bf][c][w][2 * lane + 2]), vn = *reinterpret_cast<const double2 *>(&lds.v[bf][c][w + 2][2 * lane + 2]);
          const double2 wl = fma2(vcc[c], wl1[kr], mul2(vs, wl0[kr])), wh = fma2(vn, wh1[kr], mul2(vcc[c], wh0[kr]));
          mom_row_comp(r, us, ucc[c], un, wl, wh, ya[c]);
          if (AB) aa[c] = add2(aa[c], mom_row_abs(r, tyi[3], tyi[4], tyi[5], tyi[6], wl, wh, c == 1));
          if (DG) {
            da[c].x += r.ac.x;
            da[c].y += r.ac.y;
            if (c == 1) da[1] = fma2(wh, tyi[5], fma2(wl, tyi[4], da[1]));
          }
          __builtin_amdgcn_sched_barrier(0);
        }
      }
    }
    __builtin_amdgcn_sched_barrier(0);
    // ---- E: V0 on the y-faces of plane k + 1
    fyl = LD2s(F + csl + rb1, lo, inner);
    __builtin_amdgcn_sched_barrier(0);
    // ---- H: the z axis (the shadow vector of the inner product is fetched here: it is consumed right after this phase)
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      oc[c] = (((DOT & 1) || CH) && o) ? LD2(o + (int64_t)c * csl + rb, lo) : make_double2(0., 0.);
      if (CH) xo[c] = LD2(y + (int64_t)c * csl + rb, lo);
    }
    {
      const cdouble4 *tabz = as_const(m.stab[2]) + (int64_t)k * MOM_STAB;
      const int       fl = min(k, g.fz - 1), fh = min(k + 1, g.fz - 1);
      const cdouble4 *bl = as_const(m.bw[2]) + 4 * fl, *bh = as_const(m.bw[2]) + 4 * fh;
      const double    zl0[2] = {bl[0], bl[2]}, zl1[2] = {bl[1], bl[3]}, zh0[2] = {bh[0], bh[2]}, zh1[2] = {bh[1], bh[3]};
      if (zlo || zhi) {  // first / last plane of the axis: far plane and block-end face values straight from memory (synchronous loads)
        double2 wlp[3], whp[3];
#pragma unroll
        for (int c = 0; c < 3; ++c) {
          const int kr = c == 2 ? 0 : 1;
          wlp[c] = fma2(vcc[c], zl1[kr], mul2(vzm[c], zl0[kr]));
          whp[c] = fma2(vzp[c], zh1[kr], mul2(vcc[c], zh0[kr]));
          const double *Fs = F + (int64_t)(3 * (c + 1) + 2) * csl;
          if (zlo) wlp[c] = LD2(Fs + rb, lo);
          if (zhi) whp[c] = LD2(Fs + rb1, lo);
        }
#pragma unroll
        for (int a = 0; a < 2; ++a) {
          double um[3], uc[3], up[3], uf[3], wl[3], wh[3];
#pragma unroll
          for (int c = 0; c < 3; ++c) {
            um[c] = a ? uzm[c].y : uzm[c].x;
            uc[c] = a ? ucc[c].y : ucc[c].x;
            up[c] = a ? uzp[c].y : uzp[c].x;
            wl[c] = a ? wlp[c].y : wlp[c].x;
            wh[c] = a ? whp[c].y : whp[c].x;
            uf[c] = (x + (int64_t)c * csl + rb + (k == 0 ? 2 : -2) * sxy)[il + a];
          }
          const double vl = a ? fzl.y : fzl.x, vh = a ? fzh.y : fzh.x;
          auto         T = [&](int q) { return (double)tabz[8 + q]; };
          MOM_WALL_CELL(2, T, um, uc, up, uf, vl, vh, wl, wh)
        }
      } else {
        double tz[7];
#pragma unroll
        for (int q = 0; q < 7; ++q) tz[q] = tabz[q];
        const MomRow r = mom_row_coef(tz, fzl, fzh, uzm[2], ucc[2], uzp[2]);
#pragma unroll
        for (int c = 0; c < 3; ++c) {
          const int     kr = c == 2 ? 0 : 1;
          const double2 wl = fma2(vcc[c], zl1[kr], mul2(vzm[c], zl0[kr])), wh = fma2(vzp[c], zh1[kr], mul2(vcc[c], zh0[kr]));
          mom_row_comp(r, uzm[c], ucc[c], uzp[c], wl, wh, ya[c]);
          if (AB) aa[c] = add2(aa[c], mom_row_abs(r, tz[3], tz[4], tz[5], tz[6], wl, wh, c == 2));
          if (DG) {
            da[c].x += r.ac.x;
            da[c].y += r.ac.y;
            if (c == 2) da[2] = fma2(wh, tz[5], fma2(wl, tz[4], da[2]));
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
      double2 rr = make_double2(0., 0.);
      if (CH) {
        rr = make_double2(oc[c].x - yv.x, oc[c].y - yv.y);  // r = b - A x_k; the Jacobi scaling below turns yv into z
        yv = rr;
      }
      if (DG) {
        const double d0 = cI + da[c].x, d1 = cI + da[c].y;
        if (OUT == 2) {
          yv.x = d0;
          yv.y = d1;
        } else if (AB) {
          yv.x = aa[c].x / fabs(d0);
          yv.y = aa[c].y / fabs(d1);
        } else {
          yv.x = yv.x * recip(d0);
          yv.y = yv.y * recip(d1);
        }
      }
      double2 zz = yv;
      if (CH) {
        yv.x = fma(ch_c, zz.x, fma(ch_rho, ucc[c].x - xo[c].x, ucc[c].x));
        yv.y = fma(ch_c, zz.y, fma(ch_rho, ucc[c].y - xo[c].y, ucc[c].y));
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
      if (CH) {
        const bool   o0 = rown && own0, o1 = rown && own1;
        const double z0 = o0 ? zz.x : 0., z1 = o1 ? zz.y : 0., r0 = o0 ? rr.x : 0., r1 = o1 ? rr.y : 0.;
        acc[0] += z0 + z1;
        acc[1] += z0 * z0 + z1 * z1;
        acc[2] += r0 * r0 + r1 * r1;
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
    // ---- J: plane k + 2 into the registers that became free
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      uzm[c] = LD2s(x + (int64_t)c * csl + rb2, lo, inner);
      vzm[c] = LD2s(v0 + (int64_t)c * csl + rb2, lo, inner);
    }
    fzn = LD2s(F + 2 * csl + rb2, lo, inner);
    __builtin_amdgcn_sched_barrier(0);  // ... and nowhere later: under register pressure the scheduler would sink them to their first use in the next plane
  };
  for (int kk = k0; kk < k1; kk += 3) {
    plane(kk, uA, uB, uC, vA, vB, vC, fzA, fzB, fzC);
    if (kk + 1 >= k1) break;
    plane(kk + 1, uB, uC, uA, vB, vC, vA, fzB, fzC, fzA);
    if (kk + 2 >= k1) break;
    plane(kk + 2, uC, uA, uB, vC, vA, vB, fzC, fzA, fzB);
  }
  if (DOT || CH) {
    __syncthreads();  // the sums reuse tabx
    double *red = &lds.tabx[0][0];
#pragma unroll
    for (int a = 0; a < 4; ++a) {
      const double v = wave_sum(acc[a]);
      if (lane == 0) red[a * NW + w] = v;
    }
    __syncthreads();
    if (tid < 4) {
      double v = 0.;
#pragma unroll
      for (int q = 0; q < NW; ++q) v += red[tid * NW + q];
      partial[(int64_t)tid * pstride + blockIdx.x] = v;
    }
  }
}

}  // namespace fl
