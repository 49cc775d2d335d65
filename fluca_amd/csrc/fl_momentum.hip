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

#ifndef FL_MOM_WPE
#define FL_MOM_WPE 2  // waves per SIMD the register allocator must leave room for (see tools/experiments/mom_wpe.sh)
#endif

namespace fl {

struct MomP {
  const double *tab[3];   // MOM_NTAB x len, slot-major, LOCAL block of each axis
  const double *stab[3];  // the same numbers times (cC, cL), cell-major (k_mom_scale_tab): what k_mom2 reads
  const double *bw[3];    // rows of B (build_axis_faceinterp) per LOCAL face, face-major: w0, w1 of the normal rule, w0, w1 of the tangential rule (k_mom3)
  int           len[3];
  double        cI, cC, cL;
};

// cell-to-face interpolation rows (build_axis_faceinterp): per LOCAL face of each axis, V_f = w0 v[c0] + w1 v[c0 + 1]
// (c0 local, -1 = low ghost).  kind 0 = T, 1 = B normal component, 2 = B tangential component.
struct FaceT {
  const double *w0[3][3], *w1[3][3];  // [kind][axis]
  const int    *c0[3][3];
};

__device__ __forceinline__ double uniform_d(double v)
{
  // v is wave-uniform: keep it in scalar registers
  const int lo = __builtin_amdgcn_readfirstlane(__double2loint(v)), hi = __builtin_amdgcn_readfirstlane(__double2hiint(v));
  return __hiloint2double(hi, lo);
}

// 1/d: hardware estimate + two Newton steps (correctly rounded for all but a vanishing fraction of inputs)
__device__ __forceinline__ double recip(double d)
{
  double r = __builtin_amdgcn_rcp(d);
  r        = fma(r, fma(-d, r, 1.), r);
  r        = fma(r, fma(-d, r, 1.), r);
  return r;
}

// uniform base pointer + per-lane 32-bit byte offset: lets the backend use the scalar-base addressing mode of global_load
__device__ __forceinline__ double LD(const double *base, unsigned byteoff) { return *reinterpret_cast<const double *>(reinterpret_cast<const char *>(base) + byteoff); }
__device__ __forceinline__ void   ST(double *base, unsigned byteoff, double v) { *reinterpret_cast<double *>(reinterpret_cast<char *>(base) + byteoff) = v; }

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

// One axis' share of row (cell, component C) of A.  T(slot) yields the 1-D table number of this cell along axis D
// (build_axis_momentum).  um/uc/up: component C at the cell and its two neighbours along D; nm/nc/np: the face-normal
// component D at the same places (the same numbers when C == D).  WALL == false is the fast path for cells that are
// not next to a wall of this axis: "normal" and "tangential" rules coincide, there is no far column, and the low / high
// face rows are plain two-point interpolations -- 7 table numbers instead of 20.
template <int D, int C, bool WALL, bool DG, class TF>
__device__ __forceinline__ void mom_axis(TF T, double um, double uc, double up, double ufar, double nm, double nc, double np, double vl, double vh, double wl, double wh, double cC, double cL,
                                         double &yacc, double &dacc)
{
  constexpr int r = C == D ? 1 : 0;
  double        L0, L1, L2, L3 = 0., Tl0 = 0., Tl1 = 0., Tl2 = 0., Th0 = 0., Th1 = 0., Th2 = 0., Nl0, Nl1, Nl2 = 0., Nh0 = 0., Nh1, Nh2;
  if (WALL) {
    L0 = T(r * 4 + 0); L1 = T(r * 4 + 1); L2 = T(r * 4 + 2); L3 = T(r * 4 + 3);
    Nl0 = T(11); Nl1 = T(12); Nl2 = T(13);
    Nh0 = T(17); Nh1 = T(18); Nh2 = T(19);
    if (!r) {
      Tl0 = T(8); Tl1 = T(9); Tl2 = T(10);
      Th0 = T(14); Th1 = T(15); Th2 = T(16);
    }
  } else {
    L0 = T(0); L1 = T(1); L2 = T(2);
    Nl0 = T(8); Nl1 = T(9);
    Nh1 = T(15); Nh2 = T(16);
    Tl0 = Nl0; Tl1 = Nl1; Th1 = Nh1; Th2 = Nh2;
  }
  // face values of the face-normal component ("normal" rule): second term v0interp_C v_D, and the first term when C == D
  const double Glo = Nl0 * nm + Nl1 * nc + Nl2 * np, Ghi = Nh0 * nm + Nh1 * nc + Nh2 * np;
  const double Ilo = r ? Glo : Tl0 * um + Tl1 * uc + Tl2 * up, Ihi = r ? Ghi : Th0 * um + Th1 * uc + Th2 * up;
  const double conv = vl * Ilo + vh * Ihi + wl * Glo + wh * Ghi;
  double       lap = L0 * um + L1 * uc + L2 * up;
  if (WALL) lap += L3 * ufar;
  yacc += cC * conv + cL * lap;
  if (DG) {
    double dc = r ? (vl + wl) * Nl1 + (vh + wh) * Nh1 : vl * Tl1 + vh * Th1;
    dacc += cC * dc + cL * L1;
  }
}

constexpr int MOM_RY = 8;            // grid rows per block = waves per block
[[maybe_unused]] constexpr int MOM_NT = 64 * MOM_RY;  // (the kbench build's k_mom_apply)

// LDS image of one plane of a 64 x MOM_RY tile: what a cell needs from its x/y neighbours.
struct MomLds {
  double u[3][MOM_RY + 2][66];   // velocity components incl. a one-cell ring (rows -1..RY, columns -1..64)
  double fx[4][MOM_RY][66];      // V0x, v0interp_{0,1,2} on x-faces: low face of column 0..64 (64 = high face of the last cell)
  double fy[4][MOM_RY + 1][64];  // the same on y-faces: low face of row 0..RY
};

// y = [1/diag] A x   (x padded with valid ghosts; y padded, or unpadded component-major when OUT == 1).
// OUT == 2 writes diag(A) instead (padded).  DOT: partial slots 0 sum y, 1 y.o (o padded, may be NULL), 2 x.y, 3 y.y.
//
// Block = 64 x MOM_RY tile marching through a z chunk, one thread per cell column, all three components.
//  * Every global address is (wave-uniform base) + (per-lane 32-bit byte offset).
//  * Every value is fetched from global memory exactly once per tile: the centre column of the 3 velocity components
//    and the low faces of the 12 face fields (15 streams), plus the one-cell ring of the tile.  x/y neighbours and high
//    faces are exchanged through LDS (double-buffered, one barrier per plane); z neighbours ride in registers.
//  * Software pipeline: while plane k is computed, the loads of plane k+1 (faces, ring) and k+2 (velocity, z-faces) are
//    in flight; they are consumed by the register rotation at the end of the iteration.
#ifdef FL_KBENCH_VARIANTS  // round 1 / 2's momentum product (A/B runs)
template <bool DOT, bool JAC, int OUT>
__global__ void __launch_bounds__(MOM_NT, FL_MOM_WPE) k_mom_apply(GridP g, MomP m, const double *__restrict__ x, double *__restrict__ y, const double *__restrict__ F, int64_t cs, const double *__restrict__ o,
                                                                const KspScal *__restrict__ s, double *__restrict__ partial, int pstride, int tiles_x, int nchunk, int zc, int order)
{
  __shared__ MomLds lds[2];
  __shared__ double ltabx[MOM_NTAB][64];  // the x-axis table numbers of this tile's 64 columns (general rows of wall tiles)
  __shared__ double red[4 * MOM_RY];
  if (s && s->reason != 0) return;
  constexpr bool DG = JAC || OUT == 2;
  // order 1: chunk-major and XCD-contiguous like k_cg_A (blocks are dealt round-robin over the 8 XCDs: give each XCD a contiguous range of
  // neighbouring tiles of one z chunk, so that the ring a tile re-reads was just fetched into the same L2 by its neighbour)
  int b = blockIdx.x, chunk, tile;
  if (order == 1) {
    const int nb = gridDim.x, tiles = nb / nchunk;
    if ((nb & 7) == 0) b = (b & 7) * (nb >> 3) + (b >> 3);
    chunk = b / tiles;
    tile  = b % tiles;
  } else {
    chunk = b % nchunk;
    tile  = b / nchunk;
  }
  const int      lane = threadIdx.x & 63, w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int      i0 = (tile % tiles_x) * 64, j0 = (tile / tiles_x) * MOM_RY, j = j0 + w;
  const int      i = i0 + lane;
  const bool     own = i < g.nx && j < g.ny;
  // loads are clamped to the ghost column / row (valid memory holding the right neighbour of the last cell)
  const int      il = min(i, g.nx), jl = min(j, g.ny), it = min(i, g.nx - 1), jt = min(j, g.ny - 1);
  const int      k0 = chunk * zc, k1 = min(k0 + zc, g.nz);
  const unsigned lo0 = (unsigned)(il - i0) * 8u;
  // ring: lane 0 fetches column -1 of its row, lane 63 column 64 (clamped), the others nothing
  const bool     ringlane = lane == 0 || lane == 63;
  const unsigned loring0 = lane == 0 ? 0u : (unsigned)(min(i0 + 64, g.nx) - i0 + 1) * 8u;  // relative to (row base - 1)
  const bool     toprow = w == MOM_RY - 1, botrow = w == 0;                                   // wave-uniform
  const int      jr = botrow ? max(j0 - 1, -1) : min(j0 + MOM_RY, g.ny);                      // the ring row this wave fetches
  const unsigned far0 = it == 0 ? 32u : 0u;          // relative to (base - 2): column i+2 at the low wall, i-2 elsewhere
  const bool     xwall = i0 == 0 || i0 + 64 >= g.nx;  // block-uniform: this tile touches an x wall
  const bool     ywall = jt == 0 || jt == g.ny - 1;   // wave-uniform
  const int      lx = m.len[0], ly = m.len[1], lz = m.len[2];
  const double  *tabx = m.tab[0] + it, *taby = m.tab[1] + jt;
  double         txi[7], tyi[7];  // fast-path numbers of the x and y axes: fixed for the whole chunk
  {
    const int sl[7] = {0, 1, 2, 8, 9, 15, 16};
#pragma unroll
    for (int a = 0; a < 7; ++a) {
      txi[a] = tabx[(int64_t)sl[a] * lx];
      tyi[a] = taby[(int64_t)sl[a] * ly];
    }
  }
  if (xwall) {
    // wall tiles read the general x rows from LDS: a global load in the compute phase would have to wait for every
    // prefetch issued before it (vmcnt retires in order)
    for (int q = w; q < MOM_NTAB; q += MOM_RY) ltabx[q][lane] = tabx[(int64_t)q * lx];
    __syncthreads();
  }
  auto          slot7 = [](int q) { return q < 3 ? q : (q < 10 ? q - 5 : q - 10); };  // 0,1,2,8,9,15,16 -> 0..6
  const int64_t sx = g.sx, sxy = g.sxy;
  const int64_t foy = (jt == 0 ? 2 : -2) * sx;
  const int64_t ncell = (int64_t)g.nx * g.ny * g.nz;
  const int64_t rb0 = g.off0 + (int64_t)jl * sx + i0;  // wave-uniform offset of this row in plane 0
  const int64_t rr0 = g.off0 + (int64_t)jr * sx + i0;  // ... of the ring row
  const double  cC = m.cC, cL = m.cL;
  double        acc[4] = {0., 0., 0., 0.};

  // registers of the pipeline.  Face fields are indexed f = 0: V0, 1..3: v0interp_{0,1,2}.
  double uzm[3], uc[3], uzp[3], vzl[4], vzh[4], fxl[4], fyl[4], oc[3] = {0., 0., 0.};
  auto   fld = [&](int f, int d) { return F + (f == 0 ? d : 3 + (f - 1) * 3 + d) * cs; };
  {
    const int64_t rb = rb0 + (int64_t)k0 * sxy, rr = rr0 + (int64_t)k0 * sxy;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      const double *X = x + c * cs;
      uzm[c] = LD(X + rb - sxy, lo0);
      uc[c]  = LD(X + rb, lo0);
      uzp[c] = LD(X + rb + sxy, lo0);
      if (ringlane) lds[k0 & 1].u[c][w + 1][lane == 0 ? 0 : 65] = LD(X + rb - 1, loring0);
      if (botrow) lds[k0 & 1].u[c][0][lane + 1] = LD(X + rr, lo0);
      if (toprow) lds[k0 & 1].u[c][MOM_RY + 1][lane + 1] = LD(X + rr, lo0);
      if (DOT && o) oc[c] = LD(o + c * cs + rb, lo0);
    }
#pragma unroll
    for (int f = 0; f < 4; ++f) {
      vzl[f] = LD(fld(f, 2) + rb, lo0);
      vzh[f] = LD(fld(f, 2) + rb + sxy, lo0);
      fxl[f] = LD(fld(f, 0) + rb, lo0);
      fyl[f] = LD(fld(f, 1) + rb, lo0);
      if (lane == 63) lds[k0 & 1].fx[f][w][64] = LD(fld(f, 0) + rb - 1, loring0);
      if (toprow) lds[k0 & 1].fy[f][MOM_RY][lane] = LD(fld(f, 1) + rr, lo0);
    }
  }
  for (int kl = k0; kl < k1; ++kl) {
    // Opaque copies of the plane index and the lane offsets: without them the loop optimiser turns every load stream
    // into its own 64-bit per-lane pointer carried around the loop (dozens of VGPRs, no scalar-base addressing).
    int      k = kl;
    unsigned lo = lo0, loring = loring0;
    asm volatile("" : "+s"(k), "+v"(lo), "+v"(loring));
    MomLds       &L = lds[k & 1];
    const int64_t rb = rb0 + (int64_t)k * sxy;
    const bool    zwall = k == 0 || k == g.nz - 1;
    // ---- 1: publish this thread's part of plane k (its ring went into this buffer at the end of the previous trip)
#pragma unroll
    for (int c = 0; c < 3; ++c) L.u[c][w + 1][lane + 1] = uc[c];
#pragma unroll
    for (int f = 0; f < 4; ++f) {
      L.fx[f][w][lane] = fxl[f];
      L.fy[f][w][lane] = fyl[f];
    }
    // far column of the one-sided wall rows (rare: uniform branches; straight from global memory, issued before the prefetches so that waiting for them does not drain those)
    double ufx[3] = {0., 0., 0.}, ufy[3] = {0., 0., 0.}, ufz[3] = {0., 0., 0.};
    if (xwall) {
#pragma unroll
      for (int c = 0; c < 3; ++c) ufx[c] = LD(x + c * cs + rb - 2, lo + far0);
    }
    if (ywall) {
#pragma unroll
      for (int c = 0; c < 3; ++c) ufy[c] = LD(x + c * cs + rb + foy, lo);
    }
    if (zwall) {
#pragma unroll
      for (int c = 0; c < 3; ++c) ufz[c] = LD(x + c * cs + rb + (k == 0 ? 2 : -2) * sxy, lo);
    }
    // ---- 2: loads of the next plane(s) (plane k+2 is clamped to the high ghost plane: never read past the array)
    const int64_t rb1 = rb + sxy, rb2 = rb0 + (int64_t)min(k + 2, g.nz) * sxy, rr1 = rr0 + (int64_t)(k + 1) * sxy;
    double        n_u[3], n_vz[4], n_fxl[4], n_fyl[4], n_oc[3] = {0., 0., 0.}, n_rcol_u[3], n_rcol_fx[4], n_rrow_u[3], n_rrow_fy[4];  // n_r*: ring of plane k+1
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      const double *X = x + c * cs;
      n_u[c] = LD(X + rb2, lo);
      if (DOT && o) n_oc[c] = LD(o + c * cs + rb1, lo);
    }
#pragma unroll
    for (int f = 0; f < 4; ++f) {
      n_vz[f]  = LD(fld(f, 2) + rb2, lo);
      n_fxl[f] = LD(fld(f, 0) + rb1, lo);
      n_fyl[f] = LD(fld(f, 1) + rb1, lo);
    }
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      const double *X = x + c * cs;
      n_rcol_u[c] = ringlane ? LD(X + rb1 - 1, loring) : 0.;
      n_rrow_u[c] = (botrow || toprow) ? LD(X + rr1, lo) : 0.;
    }
#pragma unroll
    for (int f = 0; f < 4; ++f) {
      n_rcol_fx[f] = lane == 63 ? LD(fld(f, 0) + rb1 - 1, loring) : 0.;
      n_rrow_fy[f] = toprow ? LD(fld(f, 1) + rr1, lo) : 0.;
    }
    __syncthreads();
    // ---- 3: the three axes
    double yacc[3] = {0., 0., 0.}, dacc[3] = {0., 0., 0.};
    {
      double um[3], up[3], fh[4];
#pragma unroll
      for (int c = 0; c < 3; ++c) {
        um[c] = L.u[c][w + 1][lane];
        up[c] = L.u[c][w + 1][lane + 2];
      }
#pragma unroll
      for (int f = 0; f < 4; ++f) fh[f] = L.fx[f][w][lane + 1];
      if (xwall) {
        auto T = [&](int q) { return ltabx[q][lane]; };
        mom_axis<0, 0, true, DG>(T, um[0], uc[0], up[0], ufx[0], um[0], uc[0], up[0], fxl[0], fh[0], fxl[1], fh[1], cC, cL, yacc[0], dacc[0]);
        mom_axis<0, 1, true, DG>(T, um[1], uc[1], up[1], ufx[1], um[0], uc[0], up[0], fxl[0], fh[0], fxl[2], fh[2], cC, cL, yacc[1], dacc[1]);
        mom_axis<0, 2, true, DG>(T, um[2], uc[2], up[2], ufx[2], um[0], uc[0], up[0], fxl[0], fh[0], fxl[3], fh[3], cC, cL, yacc[2], dacc[2]);
      } else {
        auto T = [&](int q) { return txi[slot7(q)]; };
        mom_axis<0, 0, false, DG>(T, um[0], uc[0], up[0], 0., um[0], uc[0], up[0], fxl[0], fh[0], fxl[1], fh[1], cC, cL, yacc[0], dacc[0]);
        mom_axis<0, 1, false, DG>(T, um[1], uc[1], up[1], 0., um[0], uc[0], up[0], fxl[0], fh[0], fxl[2], fh[2], cC, cL, yacc[1], dacc[1]);
        mom_axis<0, 2, false, DG>(T, um[2], uc[2], up[2], 0., um[0], uc[0], up[0], fxl[0], fh[0], fxl[3], fh[3], cC, cL, yacc[2], dacc[2]);
      }
    }
    {
      double um[3], up[3], fh[4];
#pragma unroll
      for (int c = 0; c < 3; ++c) {
        um[c] = L.u[c][w][lane + 1];
        up[c] = L.u[c][w + 2][lane + 1];
      }
#pragma unroll
      for (int f = 0; f < 4; ++f) fh[f] = L.fy[f][w + 1][lane];
      if (ywall) {
        auto T = [&](int q) { return taby[(int64_t)q * ly]; };
        mom_axis<1, 0, true, DG>(T, um[0], uc[0], up[0], ufy[0], um[1], uc[1], up[1], fyl[0], fh[0], fyl[1], fh[1], cC, cL, yacc[0], dacc[0]);
        mom_axis<1, 1, true, DG>(T, um[1], uc[1], up[1], ufy[1], um[1], uc[1], up[1], fyl[0], fh[0], fyl[2], fh[2], cC, cL, yacc[1], dacc[1]);
        mom_axis<1, 2, true, DG>(T, um[2], uc[2], up[2], ufy[2], um[1], uc[1], up[1], fyl[0], fh[0], fyl[3], fh[3], cC, cL, yacc[2], dacc[2]);
      } else {
        auto T = [&](int q) { return tyi[slot7(q)]; };
        mom_axis<1, 0, false, DG>(T, um[0], uc[0], up[0], 0., um[1], uc[1], up[1], fyl[0], fh[0], fyl[1], fh[1], cC, cL, yacc[0], dacc[0]);
        mom_axis<1, 1, false, DG>(T, um[1], uc[1], up[1], 0., um[1], uc[1], up[1], fyl[0], fh[0], fyl[2], fh[2], cC, cL, yacc[1], dacc[1]);
        mom_axis<1, 2, false, DG>(T, um[2], uc[2], up[2], 0., um[1], uc[1], up[1], fyl[0], fh[0], fyl[3], fh[3], cC, cL, yacc[2], dacc[2]);
      }
    }
    {
      const double *tabz = m.tab[2] + k;
      auto          T = [&](int q) { return tabz[(int64_t)q * lz]; };
      if (zwall) {
        mom_axis<2, 0, true, DG>(T, uzm[0], uc[0], uzp[0], ufz[0], uzm[2], uc[2], uzp[2], vzl[0], vzh[0], vzl[1], vzh[1], cC, cL, yacc[0], dacc[0]);
        mom_axis<2, 1, true, DG>(T, uzm[1], uc[1], uzp[1], ufz[1], uzm[2], uc[2], uzp[2], vzl[0], vzh[0], vzl[2], vzh[2], cC, cL, yacc[1], dacc[1]);
        mom_axis<2, 2, true, DG>(T, uzm[2], uc[2], uzp[2], ufz[2], uzm[2], uc[2], uzp[2], vzl[0], vzh[0], vzl[3], vzh[3], cC, cL, yacc[2], dacc[2]);
      } else {
        mom_axis<2, 0, false, DG>(T, uzm[0], uc[0], uzp[0], 0., uzm[2], uc[2], uzp[2], vzl[0], vzh[0], vzl[1], vzh[1], cC, cL, yacc[0], dacc[0]);
        mom_axis<2, 1, false, DG>(T, uzm[1], uc[1], uzp[1], 0., uzm[2], uc[2], uzp[2], vzl[0], vzh[0], vzl[2], vzh[2], cC, cL, yacc[1], dacc[1]);
        mom_axis<2, 2, false, DG>(T, uzm[2], uc[2], uzp[2], 0., uzm[2], uc[2], uzp[2], vzl[0], vzh[0], vzl[3], vzh[3], cC, cL, yacc[2], dacc[2]);
      }
    }
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      double yv = m.cI * uc[c] + yacc[c];
      if (DG) {
        const double dg = m.cI + dacc[c];
        if (OUT == 2) yv = dg;
        else yv = yv * recip(dg);  // PCJacobi: VecReciprocal(diag) once, VecPointwiseMult per apply
      }
      if (own) {
        if (OUT == 1) y[c * ncell + ((int64_t)k * g.ny + j) * g.nx + i] = yv;
        else ST(y + c * cs + rb, lo, yv);
        if (DOT) {
          acc[0] += yv;
          acc[1] += yv * oc[c];
          acc[2] += uc[c] * yv;
          acc[3] += yv * yv;
        }
      }
    }
    // ---- 4: rotate (this is where the loads of step 2 are waited for)
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      uzm[c] = uc[c];
      uc[c]  = uzp[c];
      uzp[c] = n_u[c];
      oc[c]  = n_oc[c];
    }
    // the ring of plane k+1 goes straight into the other LDS buffer: nobody reads that one before the next barrier
    MomLds &Ln = lds[(k + 1) & 1];
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      if (ringlane) Ln.u[c][w + 1][lane == 0 ? 0 : 65] = n_rcol_u[c];
      if (botrow) Ln.u[c][0][lane + 1] = n_rrow_u[c];
      if (toprow) Ln.u[c][MOM_RY + 1][lane + 1] = n_rrow_u[c];
    }
#pragma unroll
    for (int f = 0; f < 4; ++f) {
      vzl[f] = vzh[f];
      vzh[f] = n_vz[f];
      fxl[f] = n_fxl[f];
      fyl[f] = n_fyl[f];
      if (lane == 63) Ln.fx[f][w][64] = n_rcol_fx[f];
      if (toprow) Ln.fy[f][MOM_RY][lane] = n_rrow_fy[f];
    }
  }
  if (DOT) {
#pragma unroll
    for (int a = 0; a < 4; ++a) {
      const double v = wave_sum(acc[a]);
      if (lane == 0) red[a * MOM_RY + w] = v;
    }
    __syncthreads();
    if (threadIdx.x < 4) {
      double v = 0.;
#pragma unroll
      for (int q = 0; q < MOM_RY; ++q) v += red[threadIdx.x * MOM_RY + q];
      partial[(int64_t)threadIdx.x * pstride + blockIdx.x] = v;
    }
  }
}
#endif  // FL_KBENCH_VARIANTS

}  // namespace fl
#include "fl_stencil.h"
#include "fl_mom_tile.h"
#include "fl_mom_tile3.h"
namespace fl {

// BiCGStab vector updates on three-component padded vectors (interior cells only).
// OP 0: P = R - (omega_old beta) V + beta P
// OP 1: S = R - alpha V
// OP 2: X += alpha P + omega S ; R = S - omega T          slots: 0 R.R  1 R.RP  2 sum R
// OP 3: R = RP = b / diag (b unpadded, component-major)    slots: 0 sum R  1 R.R        (dg NULL: no preconditioner)
#ifdef FL_KBENCH_VARIANTS  // round 1's vector updates (A/B runs)
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
#endif  // FL_KBENCH_VARIANTS

// The same four updates on 128-cell row segments, two x-adjacent cells per lane (16-byte accesses, non-temporal where a value is not read
// again before it would be evicted anyway): one wave per segment, grid-stride over the segments of the block.  Padded rows start on
// a 128-byte boundary (PADX), so every pair is 16-byte aligned; the unpadded b of OP 3 is read in pairs when nx is even (pairs != 0).
#ifdef FL_KBENCH_VARIANTS  // round 2's vector updates (A/B runs)
template <int OP>
__global__ void __launch_bounds__(256) k_mom_pw2(GridP g, int64_t cs, const double *__restrict__ a0, const double *__restrict__ a1, const double *__restrict__ a2, const double *__restrict__ a3, double *__restrict__ w0,
                                                 double *__restrict__ w1, const KspScal *__restrict__ s, double *__restrict__ partial, int pstride, int pairs)
{
  __shared__ double red[3 * 4];
  if (OP != 3 && s->reason != 0) return;
  const double  alpha = s->alpha, omega = s->omega, beta = s->beta, ob = s->omega_old * s->beta;
  const int     lane = threadIdx.x & 63, nxs = (g.nx + 127) / 128;
  const int64_t nseg = (int64_t)nxs * g.ny * g.nz, ncell = (int64_t)g.nx * g.ny * g.nz;
  double        acc[3] = {0., 0., 0.};
  for (int64_t seg = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6); seg < nseg; seg += (int64_t)gridDim.x * 4) {
    const int     xs = (int)(seg % nxs);
    const int64_t R  = seg / nxs;
    const int     j = (int)(R % g.ny), k = (int)(R / g.ny), i = xs * 128 + 2 * lane;
    if (i >= g.nx) continue;
    const bool    two = i + 1 < g.nx;
    const int64_t idx = g.off0 + (int64_t)k * g.sxy + (int64_t)j * g.sx + i, ub = ((int64_t)k * g.ny + j) * g.nx + i;
    auto          ldp = [&](const double *p, int64_t q) { return two ? ld2<1>(p + q) : make_double2(p[q], 0.); };
    auto          ldk = [&](const double *p, int64_t q) { return two ? ld2<0>(p + q) : make_double2(p[q], 0.); };  // read again soon: keep it cached
    auto          stp = [&](double *p, int64_t q, double2 v) {
      if (two) st2<0>(p + q, v);
      else p[q] = v.x;
    };
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      const int64_t q = c * cs + idx;
      if (OP == 0) {
        const double2 r = ldk(a0, q), v = ldp(a1, q), p = ldp(w0, q);
        stp(w0, q, make_double2(r.x - ob * v.x + beta * p.x, r.y - ob * v.y + beta * p.y));
      } else if (OP == 1) {
        const double2 r = ldp(a0, q), v = ldp(a1, q);
        stp(w0, q, make_double2(r.x - alpha * v.x, r.y - alpha * v.y));
      } else if (OP == 2) {
        const double2 P = ldp(a0, q), S = ldp(a1, q), T = ldp(a2, q), RP = ldk(a3, q), X = ldp(w0, q);
        const double2 rn = make_double2(S.x - omega * T.x, S.y - omega * T.y);
        double2       xn = X;
        xn.x += alpha * P.x + omega * S.x;
        xn.y += alpha * P.y + omega * S.y;
        if (two) st2<1>(w0 + q, xn);
        else w0[q] = xn.x;
        stp(w1, q, rn);
        acc[0] += rn.x * rn.x + (two ? rn.y * rn.y : 0.);
        acc[1] += rn.x * RP.x + (two ? rn.y * RP.y : 0.);
        acc[2] += rn.x + (two ? rn.y : 0.);
      } else {
        const int64_t u = c * ncell + ub;
        double2       b;
        if (two && pairs) b = ld2<1>(a0 + u);
        else b = make_double2(a0[u], two ? a0[u + 1] : 0.);
        double2 r = b;
        if (a1) {
          const double2 d = ldp(a1, q);
          r.x = b.x / d.x;
          if (two) r.y = b.y / d.y;
        }
        stp(w0, q, r);
        stp(w1, q, r);
        acc[0] += r.x + (two ? r.y : 0.);
        acc[1] += r.x * r.x + (two ? r.y * r.y : 0.);
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
#endif  // FL_KBENCH_VARIANTS

// The tile walk of k_mom2 for the vector updates (experiment FLUCA_MOM_PW=3): a block marches a 128 x 8 tile through a z chunk, blocks in
// the XCD-contiguous order -- the access pattern at which the stencil kernels move 6 TB/s where the grid-stride form above moves 5.3.
template <int OP>
__global__ void __launch_bounds__(512) k_mom_pw3(GridP g, int64_t cs, const double *__restrict__ a0, const double *__restrict__ a1, const double *__restrict__ a2, const double *__restrict__ a3, double *__restrict__ w0,
                                                 double *__restrict__ w1, const KspScal *__restrict__ s, double *__restrict__ partial, int pstride, int pairs, int tiles_x, int nchunk, int zc)
{
  __shared__ double red[3 * 8];
  if (OP != 3 && s->reason != 0) return;
  const double alpha = s->alpha, omega = s->omega, beta = s->beta, ob = s->omega_old * s->beta;
  const int    nb = gridDim.x, tiles = nb / nchunk, b = xcd_remap(blockIdx.x, nb), chunk = b / tiles, tile = b % tiles;
  const int    lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int    i = (tile % tiles_x) * 128 + 2 * lane, j = (tile / tiles_x) * 8 + w;
  const int    k0 = chunk * zc, k1 = min(k0 + zc, g.nz);
  const int64_t ncell = (int64_t)g.nx * g.ny * g.nz;
  double       acc[3] = {0., 0., 0.};
  if (i < g.nx && j < g.ny) {
    const bool two = i + 1 < g.nx;
    auto       ldp = [&](const double *p, int64_t q) { return two ? ld2<1>(p + q) : make_double2(p[q], 0.); };
    auto       ldk = [&](const double *p, int64_t q) { return two ? ld2<0>(p + q) : make_double2(p[q], 0.); };
    auto       stp = [&](double *p, int64_t q, double2 v) {
      if (two) st2<0>(p + q, v);
      else p[q] = v.x;
    };
    for (int k = k0; k < k1; ++k) {
      const int64_t idx = g.off0 + (int64_t)k * g.sxy + (int64_t)j * g.sx + i, ub = ((int64_t)k * g.ny + j) * g.nx + i;
#pragma unroll
      for (int c = 0; c < 3; ++c) {
        const int64_t q = c * cs + idx;
        if (OP == 0) {
          const double2 r = ldk(a0, q), v = ldp(a1, q), p = ldp(w0, q);
          stp(w0, q, make_double2(r.x - ob * v.x + beta * p.x, r.y - ob * v.y + beta * p.y));
        } else if (OP == 4) {  // the first iteration's OP 0: P and V are zero by definition, so P = R (nothing is zeroed or read for it)
          stp(w0, q, ldk(a0, q));
        } else if (OP == 1) {
          const double2 r = ldp(a0, q), v = ldp(a1, q);
          stp(w0, q, make_double2(r.x - alpha * v.x, r.y - alpha * v.y));
        } else if (OP == 6) {  // one Chebyshev step behind an un-fused product (see k_mom3, OUT == 4): a0 = x_k, a1 = A x_k, a2 = b, a3 = diag(A) or NULL, w0 = x_{k-1} -> x_{k+1}
          const double2 xk = ldp(a0, q), ax = ldp(a1, q), bb = ldp(a2, q), xo = ldp(w0, q);
          const double2 rr = make_double2(bb.x - ax.x, bb.y - ax.y);
          double2       zz = rr;
          if (a3) {
            const double2 d = ldk(a3, q);
            zz.x = rr.x * recip(d.x);
            zz.y = two ? rr.y * recip(d.y) : 0.;
          }
          stp(w0, q, make_double2(fma(s->cheb_c, zz.x, fma(s->cheb_rho, xk.x - xo.x, xk.x)), fma(s->cheb_c, zz.y, fma(s->cheb_rho, xk.y - xo.y, xk.y))));
          acc[0] += zz.x + (two ? zz.y : 0.);
          acc[1] += zz.x * zz.x + (two ? zz.y * zz.y : 0.);
          acc[2] += rr.x * rr.x + (two ? rr.y * rr.y : 0.);
        } else if (OP == 2 || OP == 5) {  // OP 5: the first iteration's OP 2 -- X is zero by definition and not read
          const double2 P = ldp(a0, q), S = ldp(a1, q), T = ldp(a2, q), RP = ldk(a3, q), X = OP == 5 ? make_double2(0., 0.) : ldp(w0, q);
          const double2 rn = make_double2(S.x - omega * T.x, S.y - omega * T.y);
          double2       xn = X;
          xn.x += alpha * P.x + omega * S.x;
          xn.y += alpha * P.y + omega * S.y;
          if (two) st2<1>(w0 + q, xn);
          else w0[q] = xn.x;
          stp(w1, q, rn);
          acc[0] += rn.x * rn.x + (two ? rn.y * rn.y : 0.);
          acc[1] += rn.x * RP.x + (two ? rn.y * RP.y : 0.);
          acc[2] += rn.x + (two ? rn.y : 0.);
        } else {
          const int64_t u = c * ncell + ub;
          double2       bb;
          if (two && pairs) bb = ld2<1>(a0 + u);
          else bb = make_double2(a0[u], two ? a0[u + 1] : 0.);
          double2 r = bb;
          if (a1) {
            const double2 d = ldp(a1, q);
            r.x = bb.x / d.x;
            if (two) r.y = bb.y / d.y;
          }
          stp(w0, q, r);
          stp(w1, q, r);
          acc[0] += r.x + (two ? r.y : 0.);
          acc[1] += r.x * r.x + (two ? r.y * r.y : 0.);
        }
      }
    }
  }
  if (OP == 2 || OP == 3 || OP == 5 || OP == 6) {
#pragma unroll
    for (int a = 0; a < 3; ++a) {
      const double v = wave_sum(acc[a]);
      if (lane == 0) red[a * 8 + w] = v;
    }
    __syncthreads();
    if (threadIdx.x < 3) {
      double v = 0.;
#pragma unroll
      for (int q = 0; q < 8; ++q) v += red[threadIdx.x * 8 + q];
      partial[(int64_t)threadIdx.x * pstride + blockIdx.x] = v;
    }
  }
}

// y = a x + b z (z may be NULL; y may alias x or z)
__global__ void __launch_bounds__(256) k_fill(int64_t n, double v, double *__restrict__ out)
{
  for (int64_t q = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; q < n; q += (int64_t)gridDim.x * blockDim.x) out[q] = v;
}
// out = 1 / in   (VecReciprocal)
__global__ void __launch_bounds__(256) k_recip(int64_t n, const double *__restrict__ in, double *__restrict__ out)
{
  for (int64_t q = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; q < n; q += (int64_t)gridDim.x * blockDim.x) out[q] = 1. / in[q];
}
// y = (w - shift) .* x  [+ y]   (VecPointwiseMult with the scaling 1/a or 1/a - 1)
__global__ void __launch_bounds__(256) k_scale_by(int64_t n, const double *__restrict__ w, double shift, const double *x, double *y, int add)
{
  for (int64_t q = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; q < n; q += (int64_t)gridDim.x * blockDim.x) {
    const double t = (w[q] - shift) * x[q];
    y[q]           = add ? y[q] + t : t;
  }
}
// VecMDot / VecMAXPY on up to 8 vectors per launch: x is read once for all of them
// max of a and sum of b over the owned cells of three padded components each (partial[2 * block], partial[2 * block + 1]; either array may be NULL):
// four rows per pass, 16-byte loads -- eight loads in flight per thread (round 4; one row and 8-byte loads per pass ran at 3.2 TB/s)
__global__ void __launch_bounds__(256) k_max_sum_owned(GridP g, int64_t cs, const double *__restrict__ a, const double *__restrict__ b, double *__restrict__ partial)
{
  __shared__ double red[2][4];
  const int64_t plane = (int64_t)g.nz * g.ny, rows = 3 * plane;
  double        mx = 0., sm = 0.;
  for (int64_t r0 = (int64_t)blockIdx.x * 4; r0 < rows; r0 += (int64_t)gridDim.x * 4) {
    int64_t base[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int64_t r = min(r0 + u, rows - 1);  // a repeated row changes neither the maximum nor (skipped below) the sum
      const int     c = (int)(r / plane), kj = (int)(r % plane), k = kj / g.ny, j = kj % g.ny;
      base[u] = (int64_t)c * cs + g.off0 + (int64_t)k * g.sxy + (int64_t)j * g.sx;
    }
    for (int i = 2 * threadIdx.x; i < g.nx; i += 512) {
      const bool two = i + 1 < g.nx;
      double2    va[4], vb[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        va[u] = a ? (two ? *reinterpret_cast<const double2 *>(a + base[u] + i) : make_double2(a[base[u] + i], 0.)) : make_double2(0., 0.);
        vb[u] = b ? (two ? *reinterpret_cast<const double2 *>(b + base[u] + i) : make_double2(b[base[u] + i], 0.)) : make_double2(0., 0.);
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        mx = fmax(mx, fmax(va[u].x, va[u].y));
        if (r0 + u < rows) sm += vb[u].x + vb[u].y;
      }
    }
  }
  for (int off = 32; off > 0; off >>= 1) {
    mx = fmax(mx, __shfl_down(mx, off, 64));
    sm += __shfl_down(sm, off, 64);
  }
  if ((threadIdx.x & 63) == 0) {
    red[0][threadIdx.x >> 6] = mx;
    red[1][threadIdx.x >> 6] = sm;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    partial[2 * blockIdx.x]     = fmax(fmax(red[0][0], red[0][1]), fmax(red[0][2], red[0][3]));
    partial[2 * blockIdx.x + 1] = (red[1][0] + red[1][1]) + (red[1][2] + red[1][3]);
  }
}

struct Vec8 {
  const double *p[8];
};
struct Coef8 {
  double a[8];
};
// partial[i * stride + block] = sum over the block's elements of x * y_i
__global__ void __launch_bounds__(256) k_mdot8(int64_t n, const double *__restrict__ x, Vec8 Y, int k, double *__restrict__ partial, int stride)
{
  __shared__ double red[8 * 4];
  double            acc[8] = {0., 0., 0., 0., 0., 0., 0., 0.};
  for (int64_t q = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; q < n; q += (int64_t)gridDim.x * blockDim.x) {
    const double xv = x[q];
#pragma unroll
    for (int i = 0; i < 8; ++i)
      if (i < k) acc[i] += xv * Y.p[i][q];
  }
  block_sum<8>(acc, red);
  if (threadIdx.x == 0)
#pragma unroll
    for (int i = 0; i < 8; ++i) partial[(int64_t)i * stride + blockIdx.x] = acc[i];
}
// x += sum_i a_i y_i
__global__ void __launch_bounds__(256) k_maxpy8(int64_t n, double *__restrict__ x, Coef8 A, Vec8 Y, int k)
{
  for (int64_t q = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; q < n; q += (int64_t)gridDim.x * blockDim.x) {
    double t = x[q];
#pragma unroll
    for (int i = 0; i < 8; ++i)
      if (i < k) t += A.a[i] * Y.p[i][q];
    x[q] = t;
  }
}
__global__ void __launch_bounds__(256) k_lincomb(int64_t n, double a, const double *x, double b, const double *z, double *y)
{
  // a zero coefficient means "not part of the sum": the vector is not read (VecSet(y, 0) as 0 x + 0 z must not keep a NaN of x alive, nor pay for reading it)
  const bool ux = a != 0., uz = z && b != 0.;
  for (int64_t q = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; q < n; q += (int64_t)gridDim.x * blockDim.x) y[q] = (ux ? a * x[q] : 0.) + (uz ? b * z[q] : 0.);
}

// partial[block] = sum over this block's grid-stride share of x[q] y[q]
__global__ void __launch_bounds__(256) k_dot(int64_t n, const double *__restrict__ x, const double *__restrict__ y, double *__restrict__ partial)
{
  __shared__ double red[4];
  double            v[1] = {0.};
  for (int64_t q = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; q < n; q += (int64_t)gridDim.x * blockDim.x) v[0] += x[q] * y[q];
  block_sum<1>(v, red);
  if (threadIdx.x == 0) partial[blockIdx.x] = v[0];
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

// The same for the faces at the two ends of axis d only (index 0 and, where the array holds it, index n_d): what k_mom3 reads of a stored
// v0interp field.  One thread per face of the two planes.
__global__ void __launch_bounds__(256) k_pad_copy_ends(GridP g, const double *__restrict__ src, double *__restrict__ dst, int ex, int ey, int ez, int d)
{
  const int     nd = d == 0 ? g.nx : (d == 1 ? g.ny : g.nz), ed = d == 0 ? ex : (d == 1 ? ey : ez);
  const int     na = d == 0 ? ey : ex, nb = d == 2 ? ey : ez;  // the two transverse extents, a fastest
  const int     nend = ed > nd ? 2 : 1;
  const int64_t n = (int64_t)na * nb * nend;
  for (int64_t q = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; q < n; q += (int64_t)gridDim.x * blockDim.x) {
    const int a = (int)(q % na), b = (int)((q / na) % nb), f = (q / ((int64_t)na * nb)) ? nd : 0;
    const int i = d == 0 ? f : a, j = d == 1 ? f : (d == 0 ? a : b), k = d == 2 ? f : b;
    dst[g.off0 + (int64_t)k * g.sxy + (int64_t)j * g.sx + i] = src[((int64_t)k * ey + j) * ex + i];
  }
}

// V_d = rhs_d + alpha (T v)_d on the owned d-faces (unpadded face array, rhs may alias V); v: padded component d with valid ghosts
__global__ void __launch_bounds__(256) k_face_interp(GridP g, FaceT t, int kind, int d, double alpha, const double *__restrict__ vpad, const double *rhs, double *V)
{
  // a wave per 64-face row segment: row and plane numbers are wave-uniform (scalar index arithmetic and, for d != 0, scalar table reads)
  const int     ex = d == 0 ? g.fx : g.nx, ey = d == 1 ? g.fy : g.ny, ez = d == 2 ? g.fz : g.nz;
  const int64_t str = d == 0 ? 1 : (d == 1 ? (int64_t)g.sx : g.sxy);
  const int     lane = threadIdx.x & 63, w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), nw = blockDim.x >> 6;
  const int     nseg = (ex + 63) / 64;
  const int64_t nitem = (int64_t)nseg * ey * ez;
  for (int64_t it = (int64_t)blockIdx.x * nw + w; it < nitem; it += (int64_t)gridDim.x * nw) {
    const int seg = (int)(it % nseg), row = (int)(it / nseg);
    const int j = row % ey, k = row / ey, i = seg * 64 + lane;
    if (i >= ex) continue;
    const int     f = d == 0 ? i : (d == 1 ? j : k);
    const int     c0 = t.c0[kind][d][f];
    const int64_t q = ((int64_t)k * ey + j) * ex + i;
    // cell (i,j,k) with the d-th index replaced by c0
    const int64_t base = g.off0 + (int64_t)k * g.sxy + (int64_t)j * g.sx + i + (int64_t)(c0 - f) * str;
    const double  w0 = alpha * t.w0[kind][d][f], w1 = alpha * t.w1[kind][d][f];
    double        s = rhs ? rhs[q] : 0.;
    if (w0 != 0.) s += w0 * vpad[base];
    if (w1 != 0.) s += w1 * vpad[base + str];
    V[q] = s;
  }
}

// The same row on the faces at the two ends of axis d only (face 0 and, where this rank owns it, face n_d): all fl_momentum_set_state_v0 reads of a
// stored v0interp field.  One thread per face of the two planes.
__global__ void __launch_bounds__(256) k_face_interp_ends(GridP g, FaceT t, int kind, int d, const double *__restrict__ vpad, const double *rhs, double *V)
{
  const int     ex = d == 0 ? g.fx : g.nx, ey = d == 1 ? g.fy : g.ny, ez = d == 2 ? g.fz : g.nz;
  const int     nd = d == 0 ? g.nx : (d == 1 ? g.ny : g.nz), ed = d == 0 ? ex : (d == 1 ? ey : ez);
  const int     na = d == 0 ? ey : ex, nb = d == 2 ? ey : ez;
  const int64_t n = (int64_t)na * nb * (ed > nd ? 2 : 1);
  const int64_t str = d == 0 ? 1 : (d == 1 ? (int64_t)g.sx : g.sxy);
  for (int64_t q = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; q < n; q += (int64_t)gridDim.x * blockDim.x) {
    const int     a = (int)(q % na), b = (int)((q / na) % nb), f = (q / ((int64_t)na * nb)) ? nd : 0;
    const int     i = d == 0 ? f : a, j = d == 1 ? f : (d == 0 ? a : b), k = d == 2 ? f : b;
    const int64_t u = ((int64_t)k * ey + j) * ex + i;
    const int     c0 = t.c0[kind][d][f];
    const int64_t base = g.off0 + (int64_t)k * g.sxy + (int64_t)j * g.sx + i + (int64_t)(c0 - f) * str;
    const double  w0 = t.w0[kind][d][f], w1 = t.w1[kind][d][f];
    double        s = rhs ? rhs[u] : 0.;
    if (w0 != 0.) s += w0 * vpad[base];
    if (w1 != 0.) s += w1 * vpad[base + str];
    V[u] = s;
  }
}

}  // namespace fl

using namespace fl;

// ------------------------------------------------------------------------------------------------ handle

struct fl_momentum {
  fl_poisson *p = nullptr;
  MomP        mp;
  FaceT       ft;
  void       *tabs[3] = {nullptr, nullptr, nullptr};
  double     *bws[3] = {nullptr, nullptr, nullptr};
  double     *stabs[2][3] = {{nullptr, nullptr, nullptr}, {nullptr, nullptr, nullptr}};  // scaled tables: [0] the handle's coefficients, [1] fl_momentum_rhs
  std::vector<void *> ttabs;
  double     *srhs = nullptr;  // Schur right-hand side of fl_abf_apply
  double     *gr = nullptr, *gd = nullptr;  // -ksp_initial_guess_nonzero: b - A x0 and the correction (3*cells each, unpadded)
  double     *tmpv = nullptr;  // 3*cells scratch of fl_abf_jacobian_mult
  double     *F = nullptr;   // 12 padded face fields: V0[0..2], v0interp[c*3+d] at 3 + c*3 + d
  double     *dg = nullptr;  // diag(A), 3 padded components (valid after set_state)
  double     *v0p = nullptr; // v0 (3 padded components with valid ghosts) when the state came through fl_momentum_set_state_v0
  bool        fly = false;   // k_mom3: v0interp on inner faces is interpolated from v0p inside the kernel
  double     *vec[8] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
  // PCABF with schurainv / upperainv != ID (abfpc.c:81-94, 151-171): 1 / diag(A) or 1 / rowsum(A), scratch, FGMRES basis
  int                   schur_ainv = FL_ABF_AINV_ID, upper_ainv = FL_ABF_AINV_ID;
  double               *ainv[2] = {nullptr, nullptr}, *zV[3] = {nullptr, nullptr, nullptr}, *gv = nullptr;
  std::vector<double *> gm;  // cell vectors of the Schur solve with a variable-coefficient S
  std::vector<double *> kb;  // KSPGMRES on A: Krylov basis (3*cells each, allocated as the iteration needs them), w, x, r
  bool        have_state = false;
  double      dmean = 1.;   // mean_i a_ii, formed with gersh
  double      gersh = -1.;  // cached Gershgorin radius of the Jacobi-scaled operator (fl_momentum_gershgorin); < 0: not computed for this state
  int         tiles_x = 1, tiles_y = 1, nchunk = 1, zc = 1, nblocks = 1;  // 64 x 4 x zc tiles of the vector-update kernels
  int         anchunk = 1, azc = 1, ablocks = 1;                        // 64 x MOM_RY x azc tiles of k_mom_apply
  int         t2x = 1, t2chunk = 1, t2zc = 1, t2blocks = 1;             // 128 x 8 x t2zc tiles of k_mom2
  int         pw2blocks = 1;                                            // k_mom_pw2: grid-stride over 128-cell row segments
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
  if (!h->multi) {  // periodic images inside the one block: the three components in one launch per axis
    for (int d = 0; d < 3; ++d)
      if (h->wrap_local[d]) launch_wrap(h->stream, h->g, v3, d, 3, (int64_t)h->padlen);
    return 0;
  }
  for (int c = 0; c < 3; ++c) FL_CHK(fl_fill_ghosts(h, v3 + (size_t)c * h->padlen));
  return 0;
}

int mom_order()
{
  // 1 (shipped): chunk-major, XCD-contiguous; 0: chunk fastest (round 1's order); three alternating rounds at 512^3: apply 5.83 against 6.03 ms
  // (profiles/r02c_mom_order.txt)
  return FL_VARIANT(mom_order, 1);
}

int mom_kernel()
{
  // 3 (shipped): k_mom3 (v0interp formed in the kernel) when the state came with v0 (fl_momentum_set_state_v0), else k_mom2;
  // 2: always k_mom2, two cells per lane on 128 x 8 tiles, all twelve face fields read; 1: round 1/2's k_mom_apply (kbench build)
  return FL_VARIANT(mom_kernel, 3);
}
int mom_nt()
{
  return FL_VARIANT(mom_nt, 1);  // non-temporal stores of k_mom2
}

// DOT: 0 no inner products, 1 sum y and y.o (slots 0, 1), 2 x.y and y.y (slots 2, 3), 3 all four (k_mom_apply always forms all four)
template <int DOT, bool JAC, int OUT>
void mom_apply_t(fl_momentum *m, const double *x, double *y, const double *o, const KspScal *s, const MomP *coeffs = nullptr)
{
  fl_poisson *h = m->p;
  if (mom_kernel() >= 2) {
    int flags = mom_order() & 1;
    if (OUT == 1 && (h->g.nx & 1) == 0 && (reinterpret_cast<uintptr_t>(y) & 15) == 0) flags |= 2;
    if (mom_kernel() >= 3 && m->fly && h->g.ny > 8) {  // ny > 8: no 128 x 8 tile holds both ends of the y axis (k_mom3 stages one block-end row)
      hipLaunchKernelGGL((k_mom3<8, DOT, JAC, OUT, 1>), dim3(m->t2blocks), dim3(512), 0, h->stream, h->g, coeffs ? *coeffs : m->mp, x, y, m->F, (const double *)m->v0p, (int64_t)h->padlen, o, s, h->partial,
                         h->partial_stride, m->t2x, m->t2chunk, m->t2zc, flags);
      return;
    }
    if (mom_nt()) hipLaunchKernelGGL((k_mom2<8, DOT, JAC, OUT, 1>), dim3(m->t2blocks), dim3(512), 0, h->stream, h->g, coeffs ? *coeffs : m->mp, x, y, m->F, (int64_t)h->padlen, o, s, h->partial, h->partial_stride, m->t2x, m->t2chunk, m->t2zc, flags);
    else hipLaunchKernelGGL((k_mom2<8, DOT, JAC, OUT, 0>), dim3(m->t2blocks), dim3(512), 0, h->stream, h->g, coeffs ? *coeffs : m->mp, x, y, m->F, (int64_t)h->padlen, o, s, h->partial, h->partial_stride, m->t2x, m->t2chunk, m->t2zc, flags);
    return;
  }
#ifdef FL_KBENCH_VARIANTS
  hipLaunchKernelGGL((k_mom_apply<(DOT != 0), JAC, OUT>), dim3(m->ablocks), dim3(MOM_NT), 0, h->stream, h->g, coeffs ? *coeffs : m->mp, x, y, m->F, (int64_t)h->padlen, o, s, h->partial, h->partial_stride, m->tiles_x, m->anchunk, m->azc, mom_order());
#endif
}
// the scaled, cell-major tables of k_mom2 for the coefficients in p (stream-ordered; slot 0: the handle's own, slot 1: a one-off operator)
int mom_scale_tables(fl_momentum *m, int slot, MomP &p)
{
  fl_poisson *h = m->p;
  for (int d = 0; d < 3; ++d) {
    const int len = m->mp.len[d], n = len * MOM_STAB;
    hipLaunchKernelGGL(k_mom_scale_tab, dim3((n + 255) / 256), dim3(256), 0, h->stream, (const double *)m->tabs[d], len, p.cC, p.cL, m->stabs[slot][d]);
    p.stab[d] = m->stabs[slot][d];
  }
  FL_HIP(hipGetLastError());
  return 0;
}

// blocks whose partial sums a DOT launch of the momentum operator leaves behind
int mom_apply_blocks(const fl_momentum *m) { return mom_kernel() >= 2 ? m->t2blocks : m->ablocks; }

int mom_pw_kernel()
{
  return FL_VARIANT(mom_pw, 3);  // 3 (shipped): k_mom_pw3, the tile walk; 2: k_mom_pw2, 16-byte row segments; 1: round 1's k_mom_pw (kbench build)
}
// blocks of the vector-update kernels = entries of their partial sums
int mom_pw_blocks(const fl_momentum *m) { return mom_pw_kernel() >= 3 ? m->t2blocks : (mom_pw_kernel() == 2 ? m->pw2blocks : m->nblocks); }

template <int OP>
void mom_pw(fl_momentum *m, const double *a0, const double *a1, const double *a2, const double *a3, double *w0, double *w1)
{
  fl_poisson *h = m->p;
  if (mom_pw_kernel() >= 3) {
    const int pairs = (h->g.nx & 1) == 0 && (reinterpret_cast<uintptr_t>(a0) & 15) == 0;
    hipLaunchKernelGGL((k_mom_pw3<OP>), dim3(m->t2blocks), dim3(512), 0, h->stream, h->g, (int64_t)h->padlen, a0, a1, a2, a3, w0, w1, h->scal, h->partial, h->partial_stride, pairs, m->t2x, m->t2chunk, m->t2zc);
    return;
  }
#ifdef FL_KBENCH_VARIANTS
  if (mom_pw_kernel() >= 2) {
    const int pairs = (h->g.nx & 1) == 0 && (reinterpret_cast<uintptr_t>(a0) & 15) == 0;
    hipLaunchKernelGGL((k_mom_pw2<OP>), dim3(m->pw2blocks), dim3(256), 0, h->stream, h->g, (int64_t)h->padlen, a0, a1, a2, a3, w0, w1, h->scal, h->partial, h->partial_stride, pairs);
    return;
  }
  hipLaunchKernelGGL((k_mom_pw<OP>), dim3(m->nblocks), dim3(256), 0, h->stream, h->g, (int64_t)h->padlen, a0, a1, a2, a3, w0, w1, h->scal, h->partial, h->partial_stride, m->tiles_x, m->nchunk, m->zc);
#endif
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
    for (int slot = 0; slot < 2; ++slot) FL_HIP(hipMalloc((void **)&m->stabs[slot][d], sizeof(double) * (size_t)MOM_STAB * len[d]));
    m->mp.stab[d] = m->stabs[0][d];
    // interpolation rows of the owned faces
    const int nfl = d == 0 ? g.fx : (d == 1 ? g.fy : g.fz);
    std::vector<double> bw((size_t)4 * nfl, 0.);
    for (int kind = 0; kind < 3; ++kind) {
      std::vector<double> w0, w1;
      std::vector<int>    c0;
      FL_CHK(build_axis_faceinterp(h->ax[d], kind, w0, w1, c0));
      std::vector<double> l0(nfl), l1(nfl);
      std::vector<int>    lc(nfl);
      for (int f = 0; f < nfl; ++f) {
        l0[f] = w0[(size_t)(lo[d] + f)];
        l1[f] = w1[(size_t)(lo[d] + f)];
        lc[f] = (int)(c0[(size_t)(lo[d] + f)] - lo[d]);
        if (kind >= 1) {
          bw[(size_t)4 * f + 2 * (kind - 1)]     = l0[f];
          bw[(size_t)4 * f + 2 * (kind - 1) + 1] = l1[f];
        }
      }
      const void  *src[3] = {l0.data(), l1.data(), lc.data()};
      const size_t by[3] = {sizeof(double) * nfl, sizeof(double) * nfl, sizeof(int) * nfl};
      void        *dv[3];
      for (int a = 0; a < 3; ++a) {
        FL_HIP(hipMalloc(&dv[a], std::max<size_t>(by[a], 8)));
        m->ttabs.push_back(dv[a]);
        FL_HIP(hipMemcpy(dv[a], src[a], by[a], hipMemcpyHostToDevice));
      }
      m->ft.w0[kind][d] = (const double *)dv[0];
      m->ft.w1[kind][d] = (const double *)dv[1];
      m->ft.c0[kind][d] = (const int *)dv[2];
    }
    FL_HIP(hipMalloc((void **)&m->bws[d], sizeof(double) * std::max<size_t>(bw.size(), 4)));
    FL_HIP(hipMemcpy(m->bws[d], bw.data(), sizeof(double) * bw.size(), hipMemcpyHostToDevice));
    m->mp.bw[d] = m->bws[d];
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
  {
    const int atiles = m->tiles_x * ((g.ny + MOM_RY - 1) / MOM_RY);
    int       nc = std::max(1, (2048 + atiles / 2) / atiles);
    nc         = std::max(1, std::min(std::min(nc, std::max(1, g.nz / 8)), g.nz));
    if (FL_VARIANT(mom_chunks, 0) > 0) nc = std::min(FL_VARIANT(mom_chunks, 0), g.nz);  // experiments: z chunks of k_mom_apply
    if (atiles * nc > MAX_PARTIAL_BLOCKS) nc = std::max(1, MAX_PARTIAL_BLOCKS / atiles);
    m->azc     = (g.nz + nc - 1) / nc;
    m->anchunk = (g.nz + m->azc - 1) / m->azc;
    m->ablocks = atiles * m->anchunk;
    if (m->ablocks > MAX_PARTIAL_BLOCKS) return FL_ERR_SUP;
  }
  {
    // k_mom2: 128 x 8 tiles; about four blocks per CU in all (one is resident at a time: 158 KB of LDS) -- 512^3: 3.35 ms with 4 z chunks,
    // 3.59 with 2, 3.50 with 8 (profiles/r03_mom_plan.txt)
    m->t2x          = (g.nx + 127) / 128;
    const int tiles = m->t2x * ((g.ny + 7) / 8);
    int       nc    = std::max(1, (1024 + tiles / 2) / tiles);
    nc              = std::max(1, std::min(std::min(nc, std::max(1, g.nz / 8)), g.nz));
    if (FL_VARIANT(mom_chunks, 0) > 0) nc = std::min(FL_VARIANT(mom_chunks, 0), g.nz);
    if (tiles * nc > MAX_PARTIAL_BLOCKS) nc = std::max(1, MAX_PARTIAL_BLOCKS / tiles);
    m->t2zc     = (g.nz + nc - 1) / nc;
    m->t2chunk  = (g.nz + m->t2zc - 1) / m->t2zc;
    m->t2blocks = tiles * m->t2chunk;
    if (m->t2blocks > MAX_PARTIAL_BLOCKS) return FL_ERR_SUP;
  }
  {
    const int64_t nseg = (int64_t)((g.nx + 127) / 128) * g.ny * g.nz;
    int           nb   = 2048;  // 8 blocks of 4 waves per CU
    if (FL_VARIANT(mom_pw_blocks, 0) > 0) nb = FL_VARIANT(mom_pw_blocks, 0);
    m->pw2blocks = (int)std::max<int64_t>(1, std::min<int64_t>(std::min<int64_t>((nseg + 3) / 4, nb), MAX_PARTIAL_BLOCKS));
  }
  FL_CHK(fl_ensure_partials(h, std::max(std::max(m->nblocks, m->ablocks), m->t2blocks)));
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
  for (void *t : m->ttabs)
    if (t) (void)hipFree(t);
  for (double *t : m->bws)
    if (t) (void)hipFree(t);
  for (auto &sl : m->stabs)
    for (double *t : sl)
      if (t) (void)hipFree(t);
  if (m->srhs) (void)hipFree(m->srhs);
  if (m->gr) (void)hipFree(m->gr);
  if (m->gd) (void)hipFree(m->gd);
  if (m->tmpv) (void)hipFree(m->tmpv);
  if (m->F) (void)hipFree(m->F);
  if (m->dg) (void)hipFree(m->dg);
  if (m->v0p) (void)hipFree(m->v0p);
  for (double *v : m->vec)
    if (v) (void)hipFree(v);
  for (double *v : {m->ainv[0], m->ainv[1], m->zV[0], m->zV[1], m->zV[2], m->gv})
    if (v) (void)hipFree(v);
  for (double *v : m->gm)
    if (v) (void)hipFree(v);
  for (double *v : m->kb)
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
  m->gersh = -1.;
  fl_poisson *h = m->p;
  FL_HIP(hipSetDevice(h->device));
  FL_CHK(mom_scale_tables(m, 0, m->mp));
  mom_apply_t<0, false, 2>(m, m->F, m->dg, nullptr, nullptr);  // x is not used for the diagonal: any valid padded array
  FL_HIP(hipGetLastError());
  return FL_SUCCESS;
}

namespace {
int mom_set_state(fl_momentum *m, double dt, double rho, double mu, const double *const V0_dev[3], const double *const v0interp_dev[9], const double *v0_dev)
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
  // with v0 handed over and k_mom3 certain to run (see mom_apply_t), the inner faces of the nine stored v0interp fields are never read: only their
  // block-end faces are copied (FLUCA_MOM_KERNEL=2 and grids with ny <= 8 read the whole fields through k_mom2 and get whole copies)
  const bool ends_only = v0_dev && mom_kernel() >= 3 && g.ny > 8;
  for (int f = 0; f < 12; ++f) {
    const int     d = f < 3 ? f : (f - 3) % 3;
    const double *src = f < 3 ? V0_dev[f] : v0interp_dev[f - 3];
    double       *dst = m->F + (size_t)f * h->padlen;
    const int64_t n = (int64_t)ext[d][0] * ext[d][1] * ext[d][2];
    const int     nb = (int)std::min<int64_t>((n + 255) / 256, 8192);
    if (f >= 3 && ends_only) {  // k_mom3 reads a stored v0interp field on the block-end faces only
      const int64_t ne = 2 * n / std::max(ext[d][d], 1);
      hipLaunchKernelGGL(k_pad_copy_ends, dim3((unsigned)std::max<int64_t>(1, std::min<int64_t>((ne + 255) / 256, 8192))), dim3(256), 0, h->stream, g, src, dst, ext[d][0], ext[d][1], ext[d][2], d);
    } else hipLaunchKernelGGL(k_pad_copy_ext, dim3(std::max(nb, 1)), dim3(256), 0, h->stream, g, src, dst, ext[d][0], ext[d][1], ext[d][2]);
    // high face of the last owned cell: periodic image or the neighbour's first face (the physical last face came with the copy)
    if (fl_any_ghost_exchange(h)) FL_CHK(fl_fill_ghosts(h, dst));
  }
  m->fly = false;
  if (v0_dev) {
    if (!m->v0p) FL_CHK(fl_dev_alloc(h, (void **)&m->v0p, sizeof(double) * 3 * h->padlen, true));
    for (int c = 0; c < 3; ++c) launch_pad_copy(h->stream, g, v0_dev + (size_t)c * h->ncell, m->v0p + (size_t)c * h->padlen);
    FL_CHK(mom_ghosts(m, m->v0p));
    m->fly = true;
  }
  m->have_state = true;
  return fl_momentum_set_coefficients(m, 1., dt, -0.5 * mu * dt / rho);  // MatScale(A, dt); MatAXPY(A, -mu dt / 2 rho, L); MatShift(A, 1)
}
}  // namespace

extern "C" int fl_momentum_set_state(fl_momentum *m, double dt, double rho, double mu, const double *const V0_dev[3], const double *const v0interp_dev[9])
{
  return mom_set_state(m, dt, rho, mu, V0_dev, v0interp_dev, nullptr);
}

// The same state, with the cell-centred v0 that v0interp was interpolated from (v0interp = B v0 + vbc, cnlinearcart3d.c:2826-2829; vbc
// lives on boundary faces only): the operator then reads v0interp only on the faces at the ends of this rank's block and forms the
// inner ones from v0 (k_mom3: 96 B/cell per product instead of 144).
extern "C" int fl_momentum_set_state_v0(fl_momentum *m, double dt, double rho, double mu, const double *const V0_dev[3], const double *const v0interp_dev[9], const double *v0_dev)
{
  if (!v0_dev) return FL_ERR_ARG_NULL;
  return mom_set_state(m, dt, rho, mu, V0_dev, v0interp_dev, v0_dev);
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
  mom_apply_t<0, false, 1>(m, m->vec[7], y_dev, nullptr, nullptr);
  FL_HIP(hipGetLastError());
  return FL_SUCCESS;
}

extern "C" int fl_momentum_diagonal(fl_momentum *m, double *d_dev)
{
  if (!m || !d_dev) return FL_ERR_ARG_NULL;
  if (!m->have_state && m->mp.cC != 0.) return FL_ERR_ARG_WRONGSTATE;
  fl_poisson *h = m->p;
  FL_HIP(hipSetDevice(h->device));
  mom_apply_t<0, false, 2>(m, m->F, m->dg, nullptr, nullptr);
  for (int c = 0; c < 3; ++c) launch_unpad_copy(h->stream, h->g, m->dg + (size_t)c * h->padlen, d_dev + (size_t)c * h->ncell, nullptr);
  FL_HIP(hipGetLastError());
  return FL_SUCCESS;
}

// max of a and sum of b over the owned cells (all ranks) of three padded components each: set-up quantities, one launch and one host wait for both
static int max_sum_owned(fl_momentum *m, const double *a3, const double *b3, double *mx_out, double *sum_out)
{
  fl_poisson   *h = m->p;
  const int64_t rows = (int64_t)3 * h->g.nz * h->g.ny;
  const int     nb = (int)std::max<int64_t>(1, std::min<int64_t>((rows + 3) / 4, 2048));
  FL_CHK(fl_ensure_partials(h, 2 * nb));
  hipLaunchKernelGGL(k_max_sum_owned, dim3(nb), dim3(256), 0, h->stream, h->g, (int64_t)h->padlen, a3, b3, h->partial);
  FL_HIP(hipGetLastError());
  std::vector<double> part((size_t)(2 * nb));
  FL_HIP(hipMemcpyAsync(part.data(), h->partial, sizeof(double) * (size_t)(2 * nb), hipMemcpyDeviceToHost, h->stream));
  FL_HIP(hipStreamSynchronize(h->stream));
  double mx = 0., sm = 0.;
  for (int q = 0; q < nb; ++q) {  // fixed order: the same numbers run to run
    mx = std::max(mx, part[(size_t)(2 * q)]);
    sm += part[(size_t)(2 * q + 1)];
  }
  FL_CHK(fl_allreduce_max(h, &mx));  // several ranks: the bound of the whole operator
  FL_CHK(fl_allreduce_sum(h, &sm));
  if (mx_out) *mx_out = mx;
  if (sum_out) *sum_out = sm;
  return 0;
}

// Gershgorin radius of the Jacobi-scaled operator: max_i (sum_{j != i} |a_ij|) / |a_ii|, the entries taken from the same row
// coefficients the product multiplies with (k_mom2 / k_mom3, OUT == 3).  Every eigenvalue of D^-1 A lies in the disc |lambda - 1| <= radius.
// One product-sized launch and one host wait per state; cached until the state or the coefficients change.
extern "C" int fl_momentum_gershgorin(fl_momentum *m, double *radius)
{
  if (!m || !radius) return FL_ERR_ARG_NULL;
  if (!m->have_state && m->mp.cC != 0.) return FL_ERR_ARG_WRONGSTATE;
  if (mom_kernel() < 2) return FL_ERR_SUP;
  fl_poisson *h = m->p;
  FL_HIP(hipSetDevice(h->device));
  if (m->gersh < 0.) {
    FL_CHK(mom_vec(m, 7));
    mom_apply_t<0, false, 3>(m, m->F, m->vec[7], nullptr, nullptr);  // x is not used for the row sums: any valid padded array
    double dsum = 0.;
    FL_CHK(max_sum_owned(m, m->vec[7], m->dg, &m->gersh, &dsum));
    m->dmean = dsum / (3. * (double)h->ax[0].n * (double)h->ax[1].n * (double)h->ax[2].n);
  }
  *radius = m->gersh;
  return FL_SUCCESS;
}

// The interval KSPCHEBYSHEV on kspA uses when none is given (PCJACOBI): emax = 1 + g from the Gershgorin disc of D^-1 A (a bound), emin = the
// larger of 1 - g (the disc again, while the operator is diagonally dominant) and 0.9 / mean_i a_ii.  Why the mean: with E = dt C - (mu dt /
// 2 rho) L of non-negative symmetric part (a skew convection operator, a symmetric negative Laplacian) the Rayleigh quotient of
// D^-1/2 A D^-1/2 is (x.x + x.E x) / (x.D x) >= x.x / x.D x, and the vectors that make x.E x small are the smooth ones, for which x.D x / x.x
// is the MEAN diagonal entry -- 1 / max a_ii, the rigorous bound, is decided by the wall rows (twice the interior diagonal) and costs half
// again as many steps: a 32^3 channel at nu dt / h^2 = 2.56 has lambda_min = 0.1188, 1 / mean = 0.112, 1 / max = 0.061; 24 steps with
// the true interval, 27 with this rule, 37 with 1 / max (profiles/r04_mom_cheb.txt).  An emin above the true one slows the lowest modes
// down, it does not break the iteration.
extern "C" int fl_momentum_chebyshev_interval(fl_momentum *m, double *emin, double *emax)
{
  if (!m || !emin || !emax) return FL_ERR_ARG_NULL;
  double g = 0.;
  FL_CHK(fl_momentum_gershgorin(m, &g));
  *emax = 1. + g;
  *emin = std::max(1. - g, 0.9 / m->dmean);
  return FL_SUCCESS;
}

// KSPCHEBYSHEV on the momentum block (-ns_abf_momentum_ksp_type chebyshev, a PETSc option the reference's kspA accepts like any other):
// PETSc's three-term recurrence with PCJACOBI / PCNONE, zero initial guess.  Interval: opts->emin / emax (-ksp_chebyshev_eigenvalues), or, with
// PCJACOBI, fl_momentum_chebyshev_interval: [max(1 - g, 0.9 / mean a_ii), 1 + g], g the Gershgorin radius of D^-1 A -- a heuristic for emin, a bound for
// emax, and a REAL interval for a spectrum that is not (the convective part of g is imaginary): PETSc would have estimated the spectrum with GMRES.
// So a solve on the default interval is never left unwatched: with -ksp_norm_type none it still forms the preconditioned norm and stops with
// KSP_DIVERGED_DTOL / NANORINF like any KSP (dtol 1e5) instead of returning garbage after maxit steps.
// One fused launch per step where the state came with v0 (k_mom3, OUT == 4: 144 B/cell), the product and a vector update otherwise.
// from_guess: x_dev holds x_0 (the recurrence starts from any x_0: rho_0 = 0, the first step is x_1 = x_0 + c M (b - A x_0)); the first norm is then
// the guess's residual norm, so the caller passes the tolerances it means in absolute terms (momentum_solve_from_guess)
static int momentum_cheb(fl_momentum *m, const double *b_dev, double *x_dev, const fl_ksp_opts *opts, fl_ksp_stats *stats, bool from_guess = false)
{
  if (opts->pc != FL_PC_JACOBI && opts->pc != FL_PC_NONE) return FL_ERR_SUP;
  if (opts->norm_type == FL_NORM_NATURAL) return FL_ERR_SUP;
  if (opts->maxit < 0) return FL_ERR_ARG_OUTOFRANGE;
  if (mom_kernel() < 2 || mom_pw_kernel() < 3) return FL_ERR_SUP;
  fl_poisson *h = m->p;
  FL_HIP(hipSetDevice(h->device));
  std::memset(stats, 0, sizeof(*stats));
  const bool jac = opts->pc == FL_PC_JACOBI;
  double     emin = opts->emin, emax = opts->emax;
  bool       default_interval = false;
  if (emin == 0. && emax == 0.) {
    if (!jac) return FL_ERR_SUP;  // no bound of the unscaled operator is formed: give -ksp_chebyshev_eigenvalues
    FL_CHK(fl_momentum_chebyshev_interval(m, &emin, &emax));
    default_interval = true;
  }
  if (!(emax > emin) || !(emin > 0.)) return FL_ERR_ARG_OUTOFRANGE;
  const bool fused = mom_kernel() >= 3 && m->fly && h->g.ny > 8;
  for (int a : {0, 2, 4}) FL_CHK(mom_vec(m, a));
  if (!fused) FL_CHK(mom_vec(m, 3));
  double *B = m->vec[0], *X0 = m->vec[4], *X1 = m->vec[2], *AX = m->vec[3];
  const int nhist = opts->maxit + 2;
  FL_CHK(fl_ensure_hist(h, nhist));
  FL_CHK(fl_ensure_partials(h, std::max(std::max(m->nblocks, m->ablocks), m->t2blocks)));
  fl_ksp_opts o = *opts;
  o.remove_nullspace = 0;  // A = I + ... is non-singular
  const bool watched = default_interval && o.norm_type == FL_NORM_NONE;
  if (watched) {  // the heuristic interval with no norm asked for: watch the preconditioned norm anyway, stop on divergence only
    o.norm_type = FL_NORM_PRECONDITIONED;
    o.rtol      = 0.;
    o.atol      = 0.;
  }
  FL_CHK(fl_cheb_begin(h, &o, emin, emax));
  const size_t bytes = sizeof(double) * 3 * h->padlen;
  // x_0 = 0 (or the caller's guess) and "x_-1" (multiplied by rho_0 = 0) must be finite numbers
  for (double *v : {X0, X1}) FL_HIP(hipMemsetAsync(v, 0, bytes, h->stream));
  for (int c = 0; c < 3; ++c) launch_pad_copy(h->stream, h->g, b_dev + (size_t)c * h->ncell, B + (size_t)c * h->padlen);
  if (from_guess)
    for (int c = 0; c < 3; ++c) launch_pad_copy(h->stream, h->g, x_dev + (size_t)c * h->ncell, X0 + (size_t)c * h->padlen);
  const int every = o.check_every > 0 ? o.check_every : 8;
  const int total = o.norm_type == FL_NORM_NONE ? o.maxit : o.maxit + 1;  // with a norm, launch maxit is only the final test
  int       j = 0, hostcur = 0;
  bool      done = total <= 0;
  int       flags = mom_order() & 1;
  while (!done) {
    const int stop = std::min(total, j + every);
    for (; j < stop; ++j) {
      double *xin = hostcur ? X1 : X0, *xout = hostcur ? X0 : X1;
      if (j > 0 || from_guess) FL_CHK(mom_ghosts(m, xin));
      if (fused) {
        if (jac) hipLaunchKernelGGL((k_mom3<8, 0, true, 4, 1>), dim3(m->t2blocks), dim3(512), 0, h->stream, h->g, m->mp, (const double *)xin, xout, m->F, (const double *)m->v0p, (int64_t)h->padlen, (const double *)B,
                                    (const KspScal *)h->scal, h->partial, h->partial_stride, m->t2x, m->t2chunk, m->t2zc, flags);
        else hipLaunchKernelGGL((k_mom3<8, 0, false, 4, 1>), dim3(m->t2blocks), dim3(512), 0, h->stream, h->g, m->mp, (const double *)xin, xout, m->F, (const double *)m->v0p, (int64_t)h->padlen, (const double *)B,
                                (const KspScal *)h->scal, h->partial, h->partial_stride, m->t2x, m->t2chunk, m->t2zc, flags);
        FL_CHK(fl_cheb_fin_step(h, m->t2blocks, nhist));
      } else {
        mom_apply_t<0, false, 0>(m, xin, AX, nullptr, h->scal);
        mom_pw<6>(m, xin, AX, B, jac ? m->dg : nullptr, xout, nullptr);
        FL_CHK(fl_cheb_fin_step(h, mom_pw_blocks(m), nhist));
      }
      hostcur ^= 1;
    }
    FL_HIP(hipGetLastError());
    FL_CHK(fl_poll_scal(h));
    if (h->scal_host->reason != 0 || j >= total) done = true;
  }
  FL_CHK(fl_poll_scal(h));
  const double *ans = h->scal_host->cur ? X1 : X0;
  for (int c = 0; c < 3; ++c) launch_unpad_copy(h->stream, h->g, ans + (size_t)c * h->padlen, x_dev + (size_t)c * h->ncell, nullptr);
  FL_CHK(fl_ksp_finish(h, &o, stats));
  if (watched && stats->reason == FL_DIVERGED_ITS) stats->reason = FL_CONVERGED_ITS;  // what KSP_NORM_NONE reports after maxit steps
  return 0;
}

static int momentum_gmres(fl_momentum *m, const double *b_dev, double *x_dev, const fl_ksp_opts *opts, fl_ksp_stats *stats);

// KSPSolve(kspA): left-preconditioned BiCGStab (KSPBCGS) or restarted GMRES (KSPGMRES, the reference's default type for kspA,
// abfpc.c:72), zero initial guess, PCJACOBI or PCNONE
namespace {
void lincomb(fl_poisson *h, int64_t n, double a, const double *x, double b, const double *z, double *y);
}
// -ksp_initial_guess_nonzero on kspA: x_dev holds x0.  The Krylov methods here start from zero, and a Krylov method started from x0 is the same method
// started from zero on the shifted system  A d = b - A x0,  x = x0 + d  (same residuals, same iterates).  What changes is the convergence test:
// KSPConvergedDefault compares with the norm of the right-hand side b in the KSP's norm when the guess is non-zero (not with the initial residual),
// so the shifted solve runs to the ABSOLUTE tolerance max(rtol ||M b||, atol).  Costs one product, a scaled norm of b and two vector updates in
// front of the solve; saves every iteration the guess is worth -- in a time step, where x0 = the previous velocity, b - A x0 = O(dt) b.
static int momentum_solve_from_guess(fl_momentum *m, const double *b_dev, double *x_dev, const fl_ksp_opts *opts, fl_ksp_stats *stats)
{
  fl_poisson *h = m->p;
  FL_HIP(hipSetDevice(h->device));
  const int64_t n3 = 3 * h->ncell;
  // || M b || (M = 1 / diag with PCJACOBI and a preconditioned norm, else the identity): the update kernel that starts BiCGStab forms b / diag and
  // its square sum in one pass (slot 1); its two outputs are scratch here
  const bool scaled = opts->pc == FL_PC_JACOBI && opts->norm_type != FL_NORM_UNPRECONDITIONED;
  for (int a = 0; a < 2; ++a) FL_CHK(mom_vec(m, a));
  FL_CHK(fl_ensure_partials(h, std::max(std::max(m->nblocks, m->ablocks), m->t2blocks)));
  mom_pw<3>(m, b_dev, scaled ? m->dg : nullptr, nullptr, nullptr, m->vec[0], m->vec[1]);
  launch_reduce(h->stream, h->partial, mom_pw_blocks(m), h->partial_stride, 3, h->sums);
  if (h->multi) FL_CHK(h->comm.allreduce(h->stream, h->sums, NSLOT));
  double sums[NSLOT];
  FL_HIP(hipMemcpyAsync(sums, h->sums, sizeof(sums), hipMemcpyDeviceToHost, h->stream));
  FL_HIP(hipStreamSynchronize(h->stream));
  const double bnorm = std::sqrt(sums[1] > 0. ? sums[1] : 0.);
  if (opts->type == FL_KSP_CHEBYSHEV) {  // the recurrence takes the guess as it is: no shifted system, no product and no vector update in front of it
    fl_ksp_opts o = *opts;
    o.initial_guess_nonzero = 0;
    o.rtol = 0.;
    o.atol = std::max(opts->rtol * bnorm, opts->atol);
    FL_CHK(momentum_cheb(m, b_dev, x_dev, &o, stats, true));
    if (stats->reason == FL_CONVERGED_ATOL && !(stats->rnorm < opts->atol)) stats->reason = FL_CONVERGED_RTOL;
    stats->rnorm0 = bnorm;
    return FL_SUCCESS;
  }
  if (!m->gr) FL_CHK(fl_dev_alloc(h, (void **)&m->gr, sizeof(double) * (size_t)n3, false));
  if (!m->gd) FL_CHK(fl_dev_alloc(h, (void **)&m->gd, sizeof(double) * (size_t)n3, false));
  // r = b - A x0
  FL_CHK(fl_momentum_apply(m, x_dev, m->gr));
  lincomb(h, n3, 1., b_dev, -1., m->gr, m->gr);
  fl_ksp_opts o = *opts;
  o.initial_guess_nonzero = 0;
  o.rtol = 0.;
  o.atol = std::max(opts->rtol * bnorm, opts->atol);
  FL_CHK(fl_momentum_solve(m, m->gr, m->gd, &o, stats));
  lincomb(h, n3, 1., x_dev, 1., m->gd, x_dev);  // x = x0 + d
  FL_HIP(hipGetLastError());
  if (stats->reason == FL_CONVERGED_ATOL && !(stats->rnorm < opts->atol)) stats->reason = FL_CONVERGED_RTOL;  // it was the relative test that was met
  stats->rnorm0 = bnorm;  // what the relative tolerance refers to
  return FL_SUCCESS;
}

extern "C" int fl_momentum_solve(fl_momentum *m, const double *b_dev, double *x_dev, const fl_ksp_opts *opts, fl_ksp_stats *stats)
{
  if (!m || !b_dev || !x_dev || !opts || !stats) return FL_ERR_ARG_NULL;
  if (!m->have_state && m->mp.cC != 0.) return FL_ERR_ARG_WRONGSTATE;
  if (opts->initial_guess_nonzero) return momentum_solve_from_guess(m, b_dev, x_dev, opts, stats);
  if (opts->type == FL_KSP_GMRES) return momentum_gmres(m, b_dev, x_dev, opts, stats);
  if (opts->type == FL_KSP_CHEBYSHEV) return momentum_cheb(m, b_dev, x_dev, opts, stats);
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
  FL_CHK(fl_ensure_partials(h, std::max(std::max(m->nblocks, m->ablocks), m->t2blocks)));
  fl_ksp_opts o = *opts;
  o.remove_nullspace = 0;  // A = I + ... is non-singular
  FL_CHK(fl_ksp_begin(h, &o));
  // P, V and X start as zero vectors: with the tile-walk updates the first iteration says so in its kernels (P = R; X is written, not read) and
  // nothing is zeroed -- three 3N memsets and the reads of them less per solve; the older update kernels (FLUCA_MOM_PW < 3) keep the memsets
  const bool   lazy0 = mom_pw_kernel() >= 3;
  const size_t bytes = sizeof(double) * 3 * h->padlen;
  if (!lazy0)
    for (double *v : {P, V, X}) FL_HIP(hipMemsetAsync(v, 0, bytes, h->stream));
  mom_pw<3>(m, b_dev, jac ? m->dg : nullptr, nullptr, nullptr, R, RP);
  FL_CHK(fl_bcgs_fin_step(h, 0, mom_pw_blocks(m), 3, nhist));
  const int every = o.check_every > 0 ? o.check_every : 4;
  int       it = 0;
  bool      done = false;
  while (!done) {
    const int stop = std::min(o.maxit, it + every);
    for (; it < stop; ++it) {
      if (lazy0 && it == 0) mom_pw<4>(m, R, nullptr, nullptr, nullptr, P, nullptr);
      else mom_pw<0>(m, R, V, nullptr, nullptr, P, nullptr);
      FL_CHK(mom_ghosts(m, P));
      if (jac) mom_apply_t<1, true, 0>(m, P, V, RP, h->scal);
      else mom_apply_t<1, false, 0>(m, P, V, RP, h->scal);
      FL_CHK(fl_bcgs_fin_step(h, 1, mom_apply_blocks(m), 4, nhist));
      mom_pw<1>(m, R, V, nullptr, nullptr, S, nullptr);
      FL_CHK(mom_ghosts(m, S));
      if (jac) mom_apply_t<2, true, 0>(m, S, T, nullptr, h->scal);
      else mom_apply_t<2, false, 0>(m, S, T, nullptr, h->scal);
      FL_CHK(fl_bcgs_fin_step(h, 3, mom_apply_blocks(m), 4, nhist));
      if (lazy0 && it == 0) mom_pw<5>(m, P, S, T, RP, X, R);
      else mom_pw<2>(m, P, S, T, RP, X, R);
      FL_CHK(fl_bcgs_fin_step(h, 4, mom_pw_blocks(m), 3, nhist));
    }
    FL_CHK(fl_poll_scal(h));
    if (h->scal_host->reason != 0 || it >= o.maxit) done = true;
  }
  if (lazy0 && h->scal_host->it == 0) FL_HIP(hipMemsetAsync(X, 0, bytes, h->stream));  // stopped before the first update wrote X: the answer is the zero guess
  for (int c = 0; c < 3; ++c) launch_unpad_copy(h->stream, h->g, X + (size_t)c * h->padlen, x_dev + (size_t)c * h->ncell, nullptr);
  return fl_ksp_finish(h, &o, stats);
}

namespace {
int face_interp(fl_momentum *m, double alpha, const double *v_dev, const double *const rhs_dev[3], double *const V_dev[3])
{
  fl_poisson *h = m->p;
  FL_CHK(mom_vec(m, 7));
  for (int c = 0; c < 3; ++c) launch_pad_copy(h->stream, h->g, v_dev + (size_t)c * h->ncell, m->vec[7] + (size_t)c * h->padlen);
  FL_CHK(mom_ghosts(m, m->vec[7]));
  for (int d = 0; d < 3; ++d) {
    if (!V_dev[d]) return FL_ERR_ARG_NULL;
    const int64_t n = h->nface[d];
    const int     nb = (int)std::min<int64_t>((n + 255) / 256, 8192);
    hipLaunchKernelGGL(k_face_interp, dim3(std::max(nb, 1)), dim3(256), 0, h->stream, h->g, m->ft, 0, d, alpha, m->vec[7] + (size_t)d * h->padlen, rhs_dev ? rhs_dev[d] : nullptr, V_dev[d]);
  }
  FL_HIP(hipGetLastError());
  return 0;
}
void lincomb(fl_poisson *h, int64_t n, double a, const double *x, double b, const double *z, double *y)
{
  const int nb = (int)std::min<int64_t>((n + 255) / 256, 8192);
  hipLaunchKernelGGL(k_lincomb, dim3(std::max(nb, 1)), dim3(256), 0, h->stream, n, a, x, b, z, y);
}
}  // namespace

// V* = interprhs - (-T) v*   (MatMult(negT) + VecAYPX, abfpc.c:73-74)
extern "C" int fl_momentum_face_interp(fl_momentum *m, const double *v_dev, const double *const rhs_dev[3], double *const V_dev[3])
{
  if (!m || !v_dev || !V_dev) return FL_ERR_ARG_NULL;
  FL_HIP(hipSetDevice(m->p->device));
  return face_interp(m, 1., v_dev, rhs_dev, V_dev);
}

extern "C" int fl_momentum_face_interp_scaled(fl_momentum *m, double alpha, const double *v_dev, const double *const rhs_dev[3], double *const V_dev[3])
{
  if (!m || !v_dev || !V_dev) return FL_ERR_ARG_NULL;
  FL_HIP(hipSetDevice(m->p->device));
  return face_interp(m, alpha, v_dev, rhs_dev, V_dev);
}

// The part of momrhs (NSFormFunction_CNLinear_Cart3d_Internal, cnlinearcart3d.c:2976-2998) that lives on every cell:
//   momrhs = v0 + (mu dt / 2 rho) L v0 - kappa G p  (+ vbc: the boundary-condition vectors, combined by the caller)
extern "C" int fl_momentum_rhs(fl_momentum *m, double dt, double rho, double mu, const double *v0_dev, const double *p_dev, const double *vbc_dev, double *momrhs_dev)
{
  if (!m || !v0_dev || !momrhs_dev) return FL_ERR_ARG_NULL;
  if (!(rho > 0.)) return FL_ERR_ARG_OUTOFRANGE;
  fl_poisson *h = m->p;
  FL_HIP(hipSetDevice(h->device));
  const size_t N = (size_t)h->ncell;
  FL_CHK(mom_vec(m, 7));
  for (int c = 0; c < 3; ++c) launch_pad_copy(h->stream, h->g, v0_dev + c * N, m->vec[7] + (size_t)c * h->padlen);
  FL_CHK(mom_ghosts(m, m->vec[7]));
  MomP co = m->mp;
  co.cI   = 1.;
  co.cC   = 0.;
  co.cL   = 0.5 * mu * dt / rho;  // VecAXPBYPCZ(momrhs, 1, mu dt / 2 rho, 0, v0, Lv), :2988
  FL_CHK(mom_scale_tables(m, 1, co));
  mom_apply_t<0, false, 1>(m, m->vec[7], momrhs_dev, nullptr, nullptr, &co);
  if (p_dev) FL_CHK(fl_poisson_project(h, p_dev, momrhs_dev, momrhs_dev + N, momrhs_dev + 2 * N, nullptr, nullptr, nullptr));  // VecAXPY(momrhs, -1, Gp), :2993
  if (vbc_dev) lincomb(h, (int64_t)(3 * N), 1., momrhs_dev, 1., vbc_dev, momrhs_dev);
  FL_HIP(hipGetLastError());
  return FL_SUCCESS;
}

// v0interp = B v0 (+ vbc): MatMult(cnl->B, v0, cnl->v0interp); VecAXPY(v0interp, 1, vbc), cnlinearcart3d.c:2826-2829
extern "C" int fl_momentum_interp_faces(fl_momentum *m, const double *v_dev, const double *const vbc_dev[9], double *const out_dev[9])
{
  if (!m || !v_dev || !out_dev) return FL_ERR_ARG_NULL;
  fl_poisson *h = m->p;
  FL_HIP(hipSetDevice(h->device));
  FL_CHK(mom_vec(m, 7));
  for (int c = 0; c < 3; ++c) launch_pad_copy(h->stream, h->g, v_dev + (size_t)c * h->ncell, m->vec[7] + (size_t)c * h->padlen);
  FL_CHK(mom_ghosts(m, m->vec[7]));
  for (int c = 0; c < 3; ++c)
    for (int d = 0; d < 3; ++d) {
      if (!out_dev[c * 3 + d]) return FL_ERR_ARG_NULL;
      const int64_t n = h->nface[d];
      const int     nb = (int)std::min<int64_t>((n + 255) / 256, 8192);
      hipLaunchKernelGGL(k_face_interp, dim3(std::max(nb, 1)), dim3(256), 0, h->stream, h->g, m->ft, c == d ? 1 : 2, d, 1., m->vec[7] + (size_t)c * h->padlen, vbc_dev ? vbc_dev[c * 3 + d] : nullptr,
                         out_dev[c * 3 + d]);
    }
  FL_HIP(hipGetLastError());
  return FL_SUCCESS;
}

// The block-end faces of the same nine fields only (face 0 and face n of each axis, where this rank owns them): what fl_momentum_set_state_v0 reads of
// v0interp when the operator forms the inner faces from v0 itself.  The inner entries of out_dev are NOT written.
extern "C" int fl_momentum_interp_faces_ends(fl_momentum *m, const double *v_dev, const double *const vbc_dev[9], double *const out_dev[9])
{
  if (!m || !v_dev || !out_dev) return FL_ERR_ARG_NULL;
  fl_poisson *h = m->p;
  const GridP &g = h->g;
  // where the operator will read whole fields (k_mom2: FLUCA_MOM_KERNEL=2, or a block with ny <= 8) the whole fields are what is needed
  if (!(mom_kernel() >= 3 && g.ny > 8)) return fl_momentum_interp_faces(m, v_dev, vbc_dev, out_dev);
  FL_HIP(hipSetDevice(h->device));
  FL_CHK(mom_vec(m, 7));
  for (int c = 0; c < 3; ++c) launch_pad_copy(h->stream, h->g, v_dev + (size_t)c * h->ncell, m->vec[7] + (size_t)c * h->padlen);
  FL_CHK(mom_ghosts(m, m->vec[7]));
  for (int c = 0; c < 3; ++c)
    for (int d = 0; d < 3; ++d) {
      if (!out_dev[c * 3 + d]) return FL_ERR_ARG_NULL;
      const int64_t n = 2 * h->nface[d] / std::max(d == 0 ? g.fx : (d == 1 ? g.fy : g.fz), 1);
      const int     nb = (int)std::max<int64_t>(1, std::min<int64_t>((n + 255) / 256, 8192));
      hipLaunchKernelGGL(k_face_interp_ends, dim3(nb), dim3(256), 0, h->stream, g, m->ft, c == d ? 1 : 2, d, m->vec[7] + (size_t)c * h->padlen, vbc_dev ? vbc_dev[c * 3 + d] : nullptr, out_dev[c * 3 + d]);
    }
  FL_HIP(hipGetLastError());
  return FL_SUCCESS;
}

// MatMult(J) of the 3 x 3 MatNest the preconditioner is built for (NSFormJacobian_CNLinear_Cart3d_Internal,
// cnlinearcart3d.c:2885-2941):  rows v: [A, 0, kappa G]   V: [-T, I, -R]   p: [0, D, 0],   -R = (-T)(kappa G) + kappa Gst
extern "C" int fl_abf_jacobian_mult(fl_momentum *m, const double *v_dev, const double *const V_dev[3], const double *p_dev, double *fv_dev, double *const fV_dev[3], double *fp_dev)
{
  if (!m || !v_dev || !V_dev || !p_dev || !fv_dev || !fV_dev || !fp_dev) return FL_ERR_ARG_NULL;
  fl_poisson *h = m->p;
  FL_HIP(hipSetDevice(h->device));
  const int64_t N = h->ncell;
  if (!m->tmpv) FL_CHK(fl_dev_alloc(h, (void **)&m->tmpv, sizeof(double) * 3 * (size_t)N, true));
  if (!m->srhs) FL_CHK(fl_dev_alloc(h, (void **)&m->srhs, sizeof(double) * (size_t)N, true));
  for (int d = 0; d < 3; ++d)
    if (!V_dev[d] || !fV_dev[d]) return FL_ERR_ARG_NULL;
  FL_CHK(fl_momentum_apply(m, v_dev, fv_dev));                                                        // fv = A v
  lincomb(h, N, -1., p_dev, 0., nullptr, m->srhs);                                                    // -p
  lincomb(h, 3 * N, 1., v_dev, 0., nullptr, m->tmpv);                                                 // w = v
  for (int d = 0; d < 3; ++d) lincomb(h, h->nface[d], 1., V_dev[d], 0., nullptr, fV_dev[d]);          // fV = V
  FL_CHK(fl_poisson_project(h, m->srhs, m->tmpv, m->tmpv + N, m->tmpv + 2 * N, fV_dev[0], fV_dev[1], fV_dev[2]));  // w = v + kappa G p ; fV = V + kappa Gst p
  lincomb(h, 3 * N, 1., fv_dev, 1., m->tmpv, fv_dev);                                                 // fv += w
  lincomb(h, 3 * N, 1., fv_dev, -1., v_dev, fv_dev);                                                  // fv -= v      -> A v + kappa G p
  FL_CHK(face_interp(m, -1., m->tmpv, fV_dev, fV_dev));                                               // fV -= T w    -> V - T v - T kappa G p + kappa Gst p
  FL_CHK(fl_poisson_rhs(h, V_dev[0], V_dev[1], V_dev[2], nullptr, fp_dev));                           // -D V
  lincomb(h, N, -1., fp_dev, 0., nullptr, fp_dev);                                                    // fp = D V
  FL_HIP(hipGetLastError());
  return FL_SUCCESS;
}

// ------------------------------------------------------------------------------------------------ schurainv / upperainv
// PC_ABF_AINV_DIAG / ROWSUM (abfpc.c:81-94, 151-171).  With a = diag(A) or A 1 and -R = (-T)(kappa G) + kappa Gst:
//   S     = D ((-T) a^-1 kappa G - (-R)) = -D [ T ((a^-1 - 1) .* kappa G p) + kappa Gst p ]          (schurainv)
//   stage 2: v = v* - a^-1 .* kappa G p ;  V = V* - T ((a^-1 - 1) .* kappa G p) - kappa Gst p          (upperainv)
// S is then a variable-coefficient 13-point operator that changes every step.  It is applied as the composition of the
// kernels that exist (cell gradient + staggered gradient, scaling, face interpolation, divergence) and solved by flexible
// GMRES preconditioned with the constant-coefficient Schur solve (the ID operator: a = 1 + O(dt) keeps S close to it).  This is
// the non-default option of the reference; the fused matrix-free path is the ID one.
namespace {

int nblk_flat(int64_t n) { return (int)std::max<int64_t>(1, std::min<int64_t>((n + 255) / 256, 8192)); }

// out (3 * cells, unpadded) = 1 / diag(A)  or  1 / (A 1)
int compute_ainv(fl_momentum *m, int type, double **out)
{
  fl_poisson   *h = m->p;
  const int64_t n3 = 3 * h->ncell;
  if (!*out) FL_CHK(fl_dev_alloc(h, (void **)out, sizeof(double) * (size_t)n3, true));
  if (type == FL_ABF_AINV_DIAG) FL_CHK(fl_momentum_diagonal(m, *out));  // MatGetDiagonal(A)
  else FL_CHK(fl_momentum_rowsum(m, *out));                             // MatGetRowSum(A)
  hipLaunchKernelGGL(k_recip, dim3(nblk_flat(n3)), dim3(256), 0, h->stream, n3, (const double *)*out, *out);  // VecReciprocal
  return 0;
}

int ensure_ainv_scratch(fl_momentum *m)
{
  fl_poisson *h = m->p;
  if (!m->gv) FL_CHK(fl_dev_alloc(h, (void **)&m->gv, sizeof(double) * 3 * (size_t)h->ncell, true));
  for (int d = 0; d < 3; ++d)
    if (!m->zV[d]) FL_CHK(fl_dev_alloc(h, (void **)&m->zV[d], sizeof(double) * (size_t)std::max<int64_t>(h->nface[d], 1), true));
  return 0;
}

// y = S p with S = D ((-T) a^-1 kappa G - (-R)), a^-1 in m->ainv[0]
int schur_apply_var(fl_momentum *m, const double *p, double *y)
{
  fl_poisson   *h = m->p;
  const int64_t N = h->ncell, n3 = 3 * N;
  if (!h->multi && knob(K_schur_var_fused) != 0) {  // one pass over p and a^-1 (fl_schur_var.hip); several ranks keep the composition below
    FL_CHK(fl_ensure_vec(h, &h->w0));
    launch_pad_copy(h->stream, h->g, p, h->w0);
    FL_CHK(fl_fill_ghosts(h, h->w0));
    SchurVarT t;
    for (int d = 0; d < 3; ++d) {
      t.w0[d] = m->ft.w0[0][d];
      t.w1[d] = m->ft.w1[0][d];
      t.c0[d] = m->ft.c0[0][d];
    }
    return fl_schur_var_apply_fused(h, t, m->ainv[0], h->w0, y);
  }
  FL_CHK(ensure_ainv_scratch(m));
  FL_HIP(hipMemsetAsync(m->gv, 0, sizeof(double) * (size_t)n3, h->stream));
  for (int d = 0; d < 3; ++d) FL_HIP(hipMemsetAsync(m->zV[d], 0, sizeof(double) * (size_t)h->nface[d], h->stream));
  FL_CHK(fl_poisson_project(h, p, m->gv, m->gv + N, m->gv + 2 * N, m->zV[0], m->zV[1], m->zV[2]));      // gv = -kappa G p ; zV = -kappa Gst p
  hipLaunchKernelGGL(k_scale_by, dim3(nblk_flat(n3)), dim3(256), 0, h->stream, n3, (const double *)m->ainv[0], 1., (const double *)m->gv, m->gv, 0);  // gv *= a^-1 - 1
  FL_CHK(face_interp(m, 1., m->gv, m->zV, m->zV));                                                        // zV += T gv
  FL_CHK(fl_poisson_rhs(h, m->zV[0], m->zV[1], m->zV[2], nullptr, y));                                    // y = -D zV
  lincomb(h, N, -1., y, 0., nullptr, y);                                                                  // y = D zV = S p
  return 0;
}

int dot_cells(fl_poisson *h, const double *x, const double *y, double *out)
{
  const int64_t n = h->ncell;
  FL_CHK(fl_ensure_partials(h, 1024));
  const int nb = (int)std::max<int64_t>(1, std::min<int64_t>((n + 255) / 256, 1024));
  hipLaunchKernelGGL(k_dot, dim3(nb), dim3(256), 0, h->stream, n, x, y, h->partial);
  launch_reduce(h->stream, h->partial, nb, h->partial_stride, 1, h->sums);
  if (h->multi) FL_CHK(h->comm.allreduce(h->stream, h->sums, NSLOT));
  FL_HIP(hipMemcpyAsync(out, h->sums, sizeof(double), hipMemcpyDeviceToHost, h->stream));
  FL_HIP(hipStreamSynchronize(h->stream));
  return 0;
}

// KSPSolve(kspS) for the variable-coefficient S: flexible GMRES(20), right preconditioner = the constant-coefficient Schur solve
// with the caller's options (its tolerance loosened to 1e-2: the outer iteration is flexible).  Unpreconditioned residual
// norm against opts->rtol / atol, zero initial guess.
int schur_solve_var(fl_momentum *m, const double *b, double *x, const fl_ksp_opts *o, fl_ksp_stats *st)
{
  fl_poisson   *h = m->p;
  const int64_t N = h->ncell;
  constexpr int M = 20;
  // vectors: V_0..V_M, Z_0..Z_{M-1}, w
  while ((int)m->gm.size() < 2 * M + 2) {
    double *v = nullptr;
    FL_CHK(fl_dev_alloc(h, (void **)&v, sizeof(double) * (size_t)N, true));
    m->gm.push_back(v);
  }
  double **V = m->gm.data(), **Z = m->gm.data() + M + 1, *w = m->gm[2 * M + 1];
  fl_ksp_opts inner = *o;
  inner.rtol        = std::max(o->rtol, 1e-2);
  inner.history     = nullptr;
  inner.nhistory    = 0;
  fl_ksp_stats ist;
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
  FL_HIP(hipEventRecord(e0, h->stream));
  double H[(M + 1) * M], cs[M], sn[M], g[M + 1], y[M];
  double bnorm = 0., beta = 0.;
  FL_CHK(dot_cells(h, b, b, &bnorm));
  bnorm = std::sqrt(bnorm);
  const double ttol = std::max(o->rtol * bnorm, o->atol);
  FL_HIP(hipMemsetAsync(x, 0, sizeof(double) * (size_t)N, h->stream));
  int    its = 0, reason = 0, nh = 0;
  double rnorm = bnorm;
  if (o->history && o->nhistory > 0) o->history[nh++] = bnorm;
  bool first = true;
  while (!reason) {
    if (first) lincomb(h, N, 1., b, 0., nullptr, w);
    else {
      FL_CHK(schur_apply_var(m, x, w));
      lincomb(h, N, -1., w, 1., b, w);  // r = b - S x
    }
    first = false;
    FL_CHK(dot_cells(h, w, w, &beta));
    beta  = std::sqrt(beta);
    rnorm = beta;
    if (std::isnan(beta)) { reason = FL_DIVERGED_NANORINF; break; }
    if (beta <= ttol) { reason = beta < o->atol ? FL_CONVERGED_ATOL : FL_CONVERGED_RTOL; break; }
    if (its >= o->maxit) { reason = FL_DIVERGED_ITS; break; }
    lincomb(h, N, 1. / beta, w, 0., nullptr, V[0]);
    std::fill(g, g + M + 1, 0.);
    g[0] = beta;
    int j = 0;
    for (; j < M && its < o->maxit; ++j) {
      FL_CHK(fl_poisson_solve(h, V[j], Z[j], &inner, &ist));  // z_j = S_ID^-1 v_j
      if (ist.reason < 0 && ist.reason != FL_DIVERGED_ITS) { reason = ist.reason; break; }
      FL_CHK(schur_apply_var(m, Z[j], w));                    // w = S z_j
      for (int i = 0; i <= j; ++i) {                          // modified Gram-Schmidt
        double hij;
        FL_CHK(dot_cells(h, w, V[i], &hij));
        H[i * M + j] = hij;
        lincomb(h, N, 1., w, -hij, V[i], w);
      }
      double hn;
      FL_CHK(dot_cells(h, w, w, &hn));
      hn = std::sqrt(hn);
      H[(j + 1) * M + j] = hn;
      for (int i = 0; i < j; ++i) {
        const double a = H[i * M + j], c = H[(i + 1) * M + j];
        H[i * M + j]       = cs[i] * a + sn[i] * c;
        H[(i + 1) * M + j] = -sn[i] * a + cs[i] * c;
      }
      {
        const double a = H[j * M + j], c = H[(j + 1) * M + j], r = std::hypot(a, c);
        cs[j] = r > 0. ? a / r : 1.;
        sn[j] = r > 0. ? c / r : 0.;
        H[j * M + j]       = r;
        H[(j + 1) * M + j] = 0.;
        g[j + 1]           = -sn[j] * g[j];
        g[j]               = cs[j] * g[j];
      }
      ++its;
      rnorm = std::fabs(g[j + 1]);
      if (o->history && nh < o->nhistory) o->history[nh++] = rnorm;
      if (rnorm <= ttol || hn == 0.) {
        ++j;
        break;
      }
      lincomb(h, N, 1. / hn, w, 0., nullptr, V[j + 1]);
    }
    for (int i = j - 1; i >= 0; --i) {  // y = H^-1 g ; x += Z y
      double acc = g[i];
      for (int k = i + 1; k < j; ++k) acc -= H[i * M + k] * y[k];
      y[i] = acc / H[i * M + i];
    }
    for (int i = 0; i < j; ++i) lincomb(h, N, 1., x, y[i], Z[i], x);
    if (reason) break;
  }
  FL_HIP(hipEventRecord(e1, h->stream));
  FL_HIP(hipStreamSynchronize(h->stream));
  float ms = 0.f;
  FL_HIP(hipEventElapsedTime(&ms, e0, e1));
  st->iters   = its;
  st->reason  = reason;
  st->rnorm0  = bnorm;
  st->rnorm   = rnorm;
  st->seconds = ms * 1e-3;
  return 0;
}

}  // namespace

// MatGetRowSum(A): A applied to the vector of ones (couplings between the components included)
extern "C" int fl_momentum_rowsum(fl_momentum *m, double *out_dev)
{
  if (!m || !out_dev) return FL_ERR_ARG_NULL;
  if (!m->have_state && m->mp.cC != 0.) return FL_ERR_ARG_WRONGSTATE;
  fl_poisson   *h = m->p;
  const int64_t n3 = 3 * h->ncell;
  FL_HIP(hipSetDevice(h->device));
  if (!m->tmpv) FL_CHK(fl_dev_alloc(h, (void **)&m->tmpv, sizeof(double) * (size_t)n3, true));
  hipLaunchKernelGGL(k_fill, dim3(nblk_flat(n3)), dim3(256), 0, h->stream, n3, 1., m->tmpv);
  return fl_momentum_apply(m, m->tmpv, out_dev);
}

extern "C" int fl_abf_set_ainv_types(fl_momentum *m, int schur_type, int upper_type)
{
  if (!m) return FL_ERR_ARG_NULL;
  for (int t : {schur_type, upper_type})
    if (t != FL_ABF_AINV_ID && t != FL_ABF_AINV_DIAG && t != FL_ABF_AINV_ROWSUM) return FL_ERR_SUP;  // "Unsupported Ainv type"
  m->schur_ainv = schur_type;
  m->upper_ainv = upper_type;
  return FL_SUCCESS;
}

extern "C" int fl_abf_schur_apply(fl_momentum *m, const double *p_dev, double *y_dev)
{
  if (!m || !p_dev || !y_dev) return FL_ERR_ARG_NULL;
  fl_poisson *h = m->p;
  FL_HIP(hipSetDevice(h->device));
  if (m->schur_ainv == FL_ABF_AINV_ID) return fl_poisson_apply(h, p_dev, y_dev);
  if (!m->have_state && m->mp.cC != 0.) return FL_ERR_ARG_WRONGSTATE;
  FL_CHK(compute_ainv(m, m->schur_ainv, &m->ainv[0]));
  FL_CHK(schur_apply_var(m, p_dev, y_dev));
  FL_HIP(hipGetLastError());
  return FL_SUCCESS;
}

// PCApply_ABF (abfpc.c:48-111), start to finish on the device; schurainv / upperainv as set by fl_abf_set_ainv_types (default ID)
extern "C" int fl_abf_apply(fl_momentum *m, const fl_ksp_opts *momentum_opts, const fl_ksp_opts *schur_opts, const double *momrhs_dev, const double *const interprhs_dev[3], const double *contrhs_dev,
                            double *v_dev, double *const V_dev[3], double *p_dev, fl_ksp_stats stats[2])
{
  if (!m || !momentum_opts || !schur_opts || !momrhs_dev || !v_dev || !V_dev || !p_dev || !stats) return FL_ERR_ARG_NULL;
  fl_poisson *h = m->p;
  FL_HIP(hipSetDevice(h->device));
  if (!m->srhs) FL_CHK(fl_dev_alloc(h, (void **)&m->srhs, sizeof(double) * (size_t)h->ncell, true));
  /* stage 1: the lower-triangular factor */
  FL_CHK(fl_momentum_solve(m, momrhs_dev, v_dev, momentum_opts, &stats[0]));                        /* :72     v* = A^-1 momrhs */
  FL_CHK(fl_momentum_face_interp(m, v_dev, interprhs_dev, V_dev));                                  /* :73-74  V* = interprhs + T v* */
  FL_CHK(fl_poisson_rhs(h, V_dev[0], V_dev[1], V_dev[2], contrhs_dev, m->srhs));                    /* :75-76  Srhs = contrhs - D V* */
  const size_t N = (size_t)h->ncell;
  if (m->schur_ainv == FL_ABF_AINV_ID) FL_CHK(fl_poisson_solve(h, m->srhs, p_dev, schur_opts, &stats[1]));   /* :77     p = S^-1 Srhs */
  else {
    FL_CHK(compute_ainv(m, m->schur_ainv, &m->ainv[0]));                                            /* :155-165 (PCSetUp: S follows A) */
    FL_CHK(schur_solve_var(m, m->srhs, p_dev, schur_opts, &stats[1]));
  }
  /* stage 2: the upper-triangular factor; with upperainv = ID, (-T)(Gp) - (-R)p == -kappa Gst p, see DESIGN.md section 1 */
  if (m->upper_ainv == FL_ABF_AINV_ID) return fl_poisson_project(h, p_dev, v_dev, v_dev + N, v_dev + 2 * N, V_dev[0], V_dev[1], V_dev[2]);   /* :80-101 */
  FL_CHK(compute_ainv(m, m->upper_ainv, &m->ainv[1]));                                              /* :84-90 */
  FL_CHK(ensure_ainv_scratch(m));
  const int64_t n3 = 3 * (int64_t)N;
  FL_HIP(hipMemsetAsync(m->gv, 0, sizeof(double) * (size_t)n3, h->stream));
  FL_CHK(fl_poisson_project(h, p_dev, m->gv, m->gv + N, m->gv + 2 * N, V_dev[0], V_dev[1], V_dev[2]));     /* gv = -kappa G p ; V = V* - kappa Gst p */
  hipLaunchKernelGGL(k_scale_by, dim3(nblk_flat(n3)), dim3(256), 0, h->stream, n3, (const double *)m->ainv[1], 0., (const double *)m->gv, v_dev, 1);   /* v = v* - a^-1 kappa G p  (:91,95) */
  hipLaunchKernelGGL(k_scale_by, dim3(nblk_flat(n3)), dim3(256), 0, h->stream, n3, (const double *)m->ainv[1], 1., (const double *)m->gv, m->gv, 0);  /* gv = -(a^-1 - 1) kappa G p */
  FL_CHK(face_interp(m, 1., m->gv, V_dev, V_dev));                                                  /* V -= T ((a^-1 - 1) kappa G p)  (:96-101) */
  FL_HIP(hipGetLastError());
  return FL_SUCCESS;
}

// ------------------------------------------------------------------------------------------------ device BLAS-1 for hosts
// For hosts that drive an outer iteration over device vectors without a vector library of their own (the C host mirror).

extern "C" int fl_vec_lincomb(fl_poisson *h, int64_t n, double a, const double *x_dev, double b, const double *z_dev, double *y_dev)
{
  if (!h || !x_dev || !y_dev) return FL_ERR_ARG_NULL;
  if (n < 0) return FL_ERR_ARG_OUTOFRANGE;
  FL_HIP(hipSetDevice(h->device));
  if (n > 0) lincomb(h, n, a, x_dev, b, z_dev, y_dev);
  FL_HIP(hipGetLastError());
  return FL_SUCCESS;
}

extern "C" int fl_vec_dot(fl_poisson *h, int64_t n, const double *x_dev, const double *y_dev, double *result)
{
  if (!h || !x_dev || !y_dev || !result) return FL_ERR_ARG_NULL;
  if (n < 0) return FL_ERR_ARG_OUTOFRANGE;
  FL_HIP(hipSetDevice(h->device));
  FL_CHK(fl_ensure_partials(h, 1024));
  const int nb = (int)std::max<int64_t>(1, std::min<int64_t>((n + 255) / 256, 1024));
  hipLaunchKernelGGL(k_dot, dim3(nb), dim3(256), 0, h->stream, n, x_dev, y_dev, h->partial);
  launch_reduce(h->stream, h->partial, nb, h->partial_stride, 1, h->sums);
  if (h->multi) FL_CHK(h->comm.allreduce(h->stream, h->sums, NSLOT));   // every rank holds its OWNED entries only
  FL_HIP(hipMemcpyAsync(result, h->sums, sizeof(double), hipMemcpyDeviceToHost, h->stream));
  FL_HIP(hipStreamSynchronize(h->stream));
  return FL_SUCCESS;
}

// VecMDot: out[i] = x . y_i for i < k, summed over all ranks; one host wait for all k
extern "C" int fl_vec_mdot(fl_poisson *h, int64_t n, const double *x_dev, const double *const *ys_dev, int k, double *out)
{
  if (!h || !x_dev || !ys_dev || !out) return FL_ERR_ARG_NULL;
  if (n < 0 || k < 0) return FL_ERR_ARG_OUTOFRANGE;
  FL_HIP(hipSetDevice(h->device));
  FL_CHK(fl_ensure_partials(h, 1024));
  const int nb = (int)std::max<int64_t>(1, std::min<int64_t>((n + 255) / 256, 1024));
  for (int i0 = 0; i0 < k; i0 += 8) {
    const int kk = std::min(8, k - i0);
    Vec8      Y;
    for (int i = 0; i < 8; ++i) {
      Y.p[i] = i < kk ? ys_dev[i0 + i] : x_dev;
      if (!Y.p[i]) return FL_ERR_ARG_NULL;
    }
    hipLaunchKernelGGL(k_mdot8, dim3(nb), dim3(256), 0, h->stream, n, x_dev, Y, kk, h->partial, h->partial_stride);
    launch_reduce(h->stream, h->partial, nb, h->partial_stride, 8, h->sums);
    if (h->multi) FL_CHK(h->comm.allreduce(h->stream, h->sums, NSLOT));
    FL_HIP(hipMemcpyAsync(out + i0, h->sums, sizeof(double) * kk, hipMemcpyDeviceToHost, h->stream));
  }
  FL_HIP(hipStreamSynchronize(h->stream));
  return FL_SUCCESS;
}

// VecMAXPY: x += sum_i alpha_i y_i
extern "C" int fl_vec_maxpy(fl_poisson *h, int64_t n, double *x_dev, const double *alphas, const double *const *ys_dev, int k)
{
  if (!h || !x_dev || !ys_dev || !alphas) return FL_ERR_ARG_NULL;
  if (n < 0 || k < 0) return FL_ERR_ARG_OUTOFRANGE;
  FL_HIP(hipSetDevice(h->device));
  const int nb = (int)std::max<int64_t>(1, std::min<int64_t>((n + 255) / 256, 8192));
  for (int i0 = 0; i0 < k && n > 0; i0 += 8) {
    const int kk = std::min(8, k - i0);
    Vec8      Y;
    Coef8     A;
    for (int i = 0; i < 8; ++i) {
      Y.p[i] = i < kk ? ys_dev[i0 + i] : x_dev;
      A.a[i] = i < kk ? alphas[i0 + i] : 0.;
      if (!Y.p[i]) return FL_ERR_ARG_NULL;
    }
    hipLaunchKernelGGL(k_maxpy8, dim3(nb), dim3(256), 0, h->stream, n, x_dev, A, Y, kk);
  }
  FL_HIP(hipGetLastError());
  return FL_SUCCESS;
}

// ------------------------------------------------------------------------------------------------ diagnostics
// Streaming ceiling of the access mix of k_mom_apply: 15 read streams + 3 write streams, flat, 16 B per lane.
#ifdef FL_KBENCH_VARIANTS  // flat streaming probe with the product's read / write mix (tools/mom_bench.py)
namespace fl {
__global__ void __launch_bounds__(256) k_mom_stream(const double *__restrict__ F, const double *__restrict__ x, double *__restrict__ y, int64_t cs, int64_t n2)
{
  for (int64_t q = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; q < n2; q += (int64_t)gridDim.x * blockDim.x) {
    double2 a[3] = {{0., 0.}, {0., 0.}, {0., 0.}};
#pragma unroll
    for (int f = 0; f < 12; ++f) {
      const double2 v = reinterpret_cast<const double2 *>(F + f * cs)[q];
      a[f % 3].x += v.x;
      a[f % 3].y += v.y;
    }
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      const double2 v = reinterpret_cast<const double2 *>(x + c * cs)[q];
      a[c].x += v.x;
      a[c].y += v.y;
      reinterpret_cast<double2 *>(y + c * cs)[q] = a[c];
    }
  }
}
}  // namespace fl

extern "C" int fldbg_mom_stream(fl_momentum *m, int reps, int blocks, double *ms_out)
{
  fl_poisson *h = m->p;
  FL_HIP(hipSetDevice(h->device));
  FL_CHK(mom_vec(m, 6));
  FL_CHK(mom_vec(m, 7));
  const int64_t n2 = (int64_t)(h->padlen / 2);
  auto go = [&]() { hipLaunchKernelGGL(k_mom_stream, dim3(blocks), dim3(256), 0, h->stream, m->F, m->vec[7], m->vec[6], (int64_t)h->padlen, n2); };
  go();
  FL_HIP(hipEventRecord(h->ev0, h->stream));
  for (int r = 0; r < reps; ++r) go();
  FL_HIP(hipEventRecord(h->ev1, h->stream));
  FL_HIP(hipStreamSynchronize(h->stream));
  float ms = 0.f;
  FL_HIP(hipEventElapsedTime(&ms, h->ev0, h->ev1));
  *ms_out = ms / reps;
  return 0;
}
#endif  // FL_KBENCH_VARIANTS

// the operator kernel alone on padded work vectors (no pad / unpad copies): mode 0 plain, 1 Jacobi, 2 Jacobi + y.o (the first product of a
// BiCGStab iteration), 3 Jacobi + x.y, y.y (the second)
extern "C" int fldbg_mom_apply(fl_momentum *m, int mode, int reps, double *ms_out)
{
  fl_poisson *h = m->p;
  FL_HIP(hipSetDevice(h->device));
  FL_CHK(mom_vec(m, 5));
  FL_CHK(mom_vec(m, 6));
  FL_CHK(mom_vec(m, 7));
  auto go = [&]() {
    switch (mode & 3) {
      case 0: mom_apply_t<0, false, 0>(m, m->vec[7], m->vec[6], nullptr, nullptr); break;
      case 1: mom_apply_t<0, true, 0>(m, m->vec[7], m->vec[6], nullptr, nullptr); break;
      case 2: mom_apply_t<1, true, 0>(m, m->vec[7], m->vec[6], m->vec[5], nullptr); break;
      default: mom_apply_t<2, true, 0>(m, m->vec[7], m->vec[6], nullptr, nullptr); break;
    }
  };
  go();
  FL_HIP(hipEventRecord(h->ev0, h->stream));
  for (int r = 0; r < reps; ++r) go();
  FL_HIP(hipEventRecord(h->ev1, h->stream));
  FL_HIP(hipStreamSynchronize(h->stream));
  FL_HIP(hipGetLastError());
  float ms = 0.f;
  FL_HIP(hipEventElapsedTime(&ms, h->ev0, h->ev1));
  *ms_out = ms / reps;
  return 0;
}

// KSPGMRES as PETSc runs it by default: restart 30 (-ksp_gmres_restart), classical Gram-Schmidt without refinement, left
// preconditioning, the preconditioned residual norm from the Givens recurrence as the monitored norm, KSPConvergedDefault, the
// residual re-formed at every restart.  Host control flow over device vectors: per iteration one operator application, one
// VecMDot (one host wait), one VecMAXPY and one norm (a second wait) -- PETSc's own sequence.  The basis lives in unpadded
// component-major vectors; they are allocated as the iteration reaches them (PETSc's GMRES_DELTA_DIRECTIONS idea): a 512^3
// basis vector is 3.2 GB.
__global__ void __launch_bounds__(256) k_div(int64_t n, const double *__restrict__ a, const double *__restrict__ d, double *__restrict__ out)
{
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) out[i] = a[i] / d[i];
}

static int momentum_gmres(fl_momentum *m, const double *b_dev, double *x_dev, const fl_ksp_opts *opts, fl_ksp_stats *stats)
{
  if (opts->pc != FL_PC_JACOBI && opts->pc != FL_PC_NONE) return FL_ERR_SUP;
  if (opts->norm_type != FL_NORM_PRECONDITIONED) return FL_ERR_SUP;
  if (opts->maxit < 0) return FL_ERR_ARG_OUTOFRANGE;
  const int restart = opts->gmres_restart > 0 ? opts->gmres_restart : 30;
  if (restart > 1000) return FL_ERR_ARG_OUTOFRANGE;
  fl_poisson *h = m->p;
  FL_HIP(hipSetDevice(h->device));
  std::memset(stats, 0, sizeof(*stats));
  const bool    jac = opts->pc == FL_PC_JACOBI;
  const int64_t n = 3 * (int64_t)h->ncell;
  hipStream_t   s = h->stream;
  FL_CHK(mom_vec(m, 7));
  // kb[0] = w, kb[1] = x, kb[2] = r, kb[3] = M b, kb[4 + k] = v_k
  auto kvec = [&](size_t i) -> int {
    if (m->kb.size() <= i) m->kb.resize(i + 1, nullptr);
    if (!m->kb[i]) FL_CHK(fl_dev_alloc(h, (void **)&m->kb[i], sizeof(double) * (size_t)n, true));
    return 0;
  };
  for (size_t i = 0; i < 4; ++i) FL_CHK(kvec(i));
  double *W = m->kb[0], *X = m->kb[1], *R = m->kb[2];
  // y = M A x for unpadded x, y
  auto apply = [&](const double *xin, double *yout) -> int {
    for (int c = 0; c < 3; ++c) launch_pad_copy(s, h->g, xin + (size_t)c * h->ncell, m->vec[7] + (size_t)c * h->padlen);
    FL_CHK(mom_ghosts(m, m->vec[7]));
    if (jac) mom_apply_t<0, true, 1>(m, m->vec[7], yout, nullptr, nullptr);
    else mom_apply_t<0, false, 1>(m, m->vec[7], yout, nullptr, nullptr);
    return 0;
  };
  // M b, once: b ./ diag(A) (PCJACOBI) or b
  double *MB = m->kb[3];
  if (jac) {
    FL_CHK(fl_momentum_diagonal(m, W));  // unpadded diag(A); W is free at this point
    hipLaunchKernelGGL(k_div, dim3(nblk_flat(n)), dim3(256), 0, s, n, b_dev, (const double *)W, MB);
  } else lincomb(h, n, 1., b_dev, 0., nullptr, MB);
  hipEvent_t e0 = h->ev0, e1 = h->ev1;
  FL_HIP(hipEventRecord(e0, s));
  std::vector<double> hist, H((size_t)(restart + 1) * restart, 0.), cs(restart, 0.), sn(restart, 0.), g(restart + 1, 0.), hcol(restart + 1, 0.), y(restart, 0.);
  std::vector<const double *> basis(restart + 1, nullptr);
  FL_HIP(hipMemsetAsync(X, 0, sizeof(double) * (size_t)n, s));
  lincomb(h, n, 1., MB, 0., nullptr, R);  // x = 0: r = M b
  double beta = 0.;
  FL_CHK(fl_vec_dot(h, n, R, R, &beta));
  beta = std::sqrt(beta);
  const double rnorm0 = beta, ttol = std::max(opts->rtol * rnorm0, opts->atol);
  auto converged = [&](double v) {
    if (std::isnan(v) || std::isinf(v)) return (int)FL_DIVERGED_NANORINF;
    if (v <= ttol) return v < opts->atol ? (int)FL_CONVERGED_ATOL : (int)FL_CONVERGED_RTOL;
    if (v >= opts->dtol * rnorm0) return (int)FL_DIVERGED_DTOL;
    return 0;
  };
  hist.push_back(beta);
  int    it = 0, reason = converged(beta);
  double res = beta;
  if (!reason && opts->maxit == 0) reason = FL_DIVERGED_ITS;
  while (!reason) {
    // one restart cycle
    FL_CHK(kvec(4));
    lincomb(h, n, 1. / beta, R, 0., nullptr, m->kb[4]);
    std::fill(g.begin(), g.end(), 0.);
    g[0] = beta;
    int k = 0;
    for (; k < restart && !reason; ++k) {
      FL_CHK(kvec(4 + (size_t)k + 1));
      for (int i = 0; i <= k; ++i) basis[i] = m->kb[4 + (size_t)i];
      FL_CHK(apply(m->kb[4 + (size_t)k], W));
      FL_CHK(fl_vec_mdot(h, n, W, basis.data(), k + 1, hcol.data()));            // h_i = w . v_i   (classical Gram-Schmidt)
      for (int i = 0; i <= k; ++i) y[i] = -hcol[i];
      FL_CHK(fl_vec_maxpy(h, n, W, y.data(), basis.data(), k + 1));              // w -= sum h_i v_i
      double hk1 = 0.;
      FL_CHK(fl_vec_dot(h, n, W, W, &hk1));
      hk1 = std::sqrt(hk1);
      hcol[k + 1] = hk1;
      const bool happy = !(hk1 > 1e-30 * rnorm0);                                 // KSPGMRES happy breakdown: the Krylov space is invariant
      if (!happy) lincomb(h, n, 1. / hk1, W, 0., nullptr, m->kb[4 + (size_t)k + 1]);
      // previous rotations, then the new one
      for (int i = 0; i < k; ++i) {
        const double t = cs[i] * hcol[i] + sn[i] * hcol[i + 1];
        hcol[i + 1]    = -sn[i] * hcol[i] + cs[i] * hcol[i + 1];
        hcol[i]        = t;
      }
      const double d = std::hypot(hcol[k], hcol[k + 1]);
      cs[k] = d > 0. ? hcol[k] / d : 1.;
      sn[k] = d > 0. ? hcol[k + 1] / d : 0.;
      hcol[k]     = d;
      hcol[k + 1] = 0.;
      g[k + 1]    = -sn[k] * g[k];
      g[k]        = cs[k] * g[k];
      for (int i = 0; i <= k; ++i) H[(size_t)i * restart + k] = hcol[i];
      res = std::fabs(g[k + 1]);
      ++it;
      hist.push_back(res);
      reason = converged(res);
      if (!reason && happy) reason = FL_CONVERGED_RTOL;  // exact solution in the current space (PETSc: CONVERGED_HAPPY_BREAKDOWN maps to converged)
      if (!reason && it >= opts->maxit) reason = FL_DIVERGED_ITS;
    }
    // x += V y,  H y = g  (upper triangular, k columns)
    for (int i = k - 1; i >= 0; --i) {
      double t = g[i];
      for (int j = i + 1; j < k; ++j) t -= H[(size_t)i * restart + j] * y[j];
      y[i] = t / H[(size_t)i * restart + i];
    }
    for (int i = 0; i < k; ++i) basis[i] = m->kb[4 + (size_t)i];
    FL_CHK(fl_vec_maxpy(h, n, X, y.data(), basis.data(), k));
    if (reason) break;
    // restart: r = M (b - A x)
    FL_CHK(apply(X, W));                    // W = M A x
    lincomb(h, n, 1., MB, -1., W, R);
    FL_CHK(fl_vec_dot(h, n, R, R, &beta));
    beta = std::sqrt(beta);
    if (!(beta > 0.)) {
      reason = std::isnan(beta) ? FL_DIVERGED_NANORINF : FL_CONVERGED_ATOL;
      break;
    }
  }
  lincomb(h, n, 1., X, 0., nullptr, x_dev);
  FL_HIP(hipEventRecord(e1, s));
  FL_HIP(hipStreamSynchronize(s));
  FL_HIP(hipGetLastError());
  float ms = 0.f;
  FL_HIP(hipEventElapsedTime(&ms, e0, e1));
  stats->iters   = it;
  stats->reason  = reason;
  stats->rnorm0  = rnorm0;
  stats->rnorm   = res;
  stats->seconds = ms * 1e-3;
  if (opts->history && opts->nhistory > 0) std::memcpy(opts->history, hist.data(), sizeof(double) * std::min<size_t>((size_t)opts->nhistory, hist.size()));
  return FL_SUCCESS;
}

